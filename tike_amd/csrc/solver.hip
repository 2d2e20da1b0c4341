// Small fused kernels of the lstsq_grad host loop for gfx950: everything that
// is psi-, probe- or (positions,)-sized between the heavy kernels of a
// minibatch.  Each replaces a chain of 5-30 element-wise / reduction launches
// (about 120 per minibatch before) with one; sums that span all ranks are
// left in small device buffers so that the caller can all-reduce them between
// two phases.
//
// Reference: src/tike/ptycho/solvers/lstsq.py
//   :605-616  _precondition_object_update            -> tike_object_update_precond
//   :641-718  2x2 step-size systems, batch means      -> tike_lstsq_step_sums / _solve
//   :175-201  probe update                            -> tike_probe_update
//   :721-738, ptycho/probe.py:362-476 eigen-probe weights / normalisation
//                                                     -> tike_eigen_*
#include "internal.h"
#include "tike_amd.h"

// ------------------------------------------------ preconditioned object update
// g = acc (planar re / im float planes, the scatter accumulator)
// out = g / sqrt(((1 - alpha) P)^2 + (alpha max P)^2),  P = Re precond
// optional: upd_sum = g as interleaved complex; combined (planar) += g.
__global__ __launch_bounds__(256) void object_update_precond_kernel(
    const float* __restrict__ acc, const cf* __restrict__ precond,
    const float* __restrict__ pmax, float alpha, cf* __restrict__ upd_sum,
    cf* __restrict__ upd_precond, float* __restrict__ combined, long npix) {
  const float am = alpha * pmax[0];
  for (long i = blockIdx.x * 256L + threadIdx.x; i < npix; i += gridDim.x * 256L) {
    const float re = acc[i], im = acc[npix + i];
    const float p = (1.0f - alpha) * precond[i].x;
    const float inv = 1.0f / sqrtf(p * p + am * am);
    if (upd_sum) upd_sum[i] = mk(re, im);
    if (upd_precond) upd_precond[i] = mk(re * inv, im * inv);
    if (combined) {
      combined[i] += re;
      combined[npix + i] += im;
    }
  }
}

extern "C" int tike_object_update_precond(const float* acc, const void* precond,
                                          const float* pmax, float alpha, void* upd_sum,
                                          void* upd_precond, float* combined, long npix,
                                          void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(npix >= 0);
  if (npix == 0) return TK_OK;
  TK_CHECK_ARG(acc && precond && pmax);
  hipLaunchKernelGGL(object_update_precond_kernel, dim3(tk_grid((npix + 255) / 256, 8)),
                     dim3(256), 0, (hipStream_t)stream, acc, (const cf*)precond, pmax, alpha,
                     (cf*)upd_sum, (cf*)upd_precond, combined, npix);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------------------- step-size systems
// stats (B, 8) from tike_lstsq_step_stats.  Phase 1: sums = { sum(A1 + eps),
// sum(A4 + eps), sum(costs) } over this rank's B positions (all-reduced by the
// caller when there are several ranks).  Phase 2, with the GLOBAL sums and
// count: A1 += sums0 / (2 count), A4 += sums1 / (2 count) (lstsq.py:666-674),
// the 2x2 solves (:676-700), out = { sum 0.9 max(0, Re x1), sum 0.9 max(0, Re
// x2) } (all-reduced by the caller), and -- valid for a single rank -- out[2..4]
// = { beta_object, beta_probe, mean cost }.  One workgroup: B is a minibatch.
__global__ __launch_bounds__(256) void step_sums_kernel(const float* __restrict__ stats,
                                                        const float* __restrict__ costs, int B,
                                                        float eps, float* __restrict__ sums) {
  __shared__ float red[4];
  float a1 = 0.f, a4 = 0.f, c = 0.f;
  for (int n = threadIdx.x; n < B; n += 256) {
    a1 += stats[8 * n] + eps;
    a4 += stats[8 * n + 1] + eps;
    if (costs) c += costs[n];
  }
  a1 = tk_block_sum256(a1, red);
  a4 = tk_block_sum256(a4, red);
  c = tk_block_sum256(c, red);
  if (threadIdx.x == 0) {
    sums[0] = a1;
    sums[1] = a4;
    sums[2] = c;
  }
}

__global__ __launch_bounds__(256) void step_solve_kernel(const float* __restrict__ stats, int B,
                                                         float eps,
                                                         const float* __restrict__ sums,
                                                         float inv_count, int recover_psi,
                                                         int recover_probe,
                                                         float* __restrict__ out) {
  __shared__ float red[4];
  const float r1 = 0.5f * sums[0] * inv_count, r4 = 0.5f * sums[1] * inv_count;
  float so = 0.f, sp = 0.f;
  for (int n = threadIdx.x; n < B; n += 256) {
    const float* s = stats + 8 * n;
    const float A1 = s[0] + eps + r1, A4 = s[1] + eps + r4;
    const float b1 = s[4], b2 = s[5];
    float x1 = 0.f, x2 = 0.f;
    if (recover_psi && recover_probe) {
      // A2 = s2 + i s3, A3 = conj(A2): determinant and both numerators are
      // complex with (numerically) zero imaginary part in the reference; the
      // step uses the real parts only
      const float det = A1 * A4 - (s[2] * s[2] + s[3] * s[3]);
      x1 = -(s[2] * b2 - A4 * b1) / det;
      x2 = (A1 * b2 - s[2] * b1) / det;
    } else if (recover_psi) {
      x1 = b1 / A1;
    } else if (recover_probe) {
      x2 = b2 / A4;
    }
    so += 0.9f * fmaxf(x1, 0.f);
    sp += 0.9f * fmaxf(x2, 0.f);
  }
  so = tk_block_sum256(so, red);
  sp = tk_block_sum256(sp, red);
  if (threadIdx.x == 0) {
    out[0] = so;
    out[1] = sp;
    out[2] = so * inv_count;
    out[3] = sp * inv_count;
    out[4] = sums[2] * inv_count;
  }
}

extern "C" int tike_lstsq_step_sums(const float* stats, const float* costs, int B, float eps,
                                    float* sums, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(B >= 0 && sums && (B == 0 || stats));
  hipLaunchKernelGGL(step_sums_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, costs,
                     B, eps, sums);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_lstsq_step_solve(const float* stats, int B, float eps, const float* sums,
                                     double count, int recover_psi, int recover_probe,
                                     float* out, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(B >= 0 && sums && out && count > 0 && (B == 0 || stats));
  hipLaunchKernelGGL(step_solve_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, B,
                     eps, sums, (float)(1.0 / count), recover_psi, recover_probe, out);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------------------------- probe update
// dprobe = beta * mpu;  combined += dprobe * inv_num_batch;  probe += dprobe
// (lstsq.py:177-181; beta is a device scalar, so no host round trip)
__global__ __launch_bounds__(256) void probe_update_kernel(cf* __restrict__ probe,
                                                           cf* __restrict__ combined,
                                                           const cf* __restrict__ mpu,
                                                           const float* __restrict__ beta,
                                                           float inv_num_batch, long n) {
  const float b = beta[0];
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const cf d = mpu[i] * b;
    probe[i] = probe[i] + d;
    if (combined) combined[i] = combined[i] + d * inv_num_batch;
  }
}

extern "C" int tike_probe_update(void* probe, void* combined, const void* mpu,
                                 const float* beta, float inv_num_batch, long n, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(n >= 0);
  if (n == 0) return TK_OK;
  TK_CHECK_ARG(probe && mpu && beta);
  hipLaunchKernelGGL(probe_update_kernel, dim3(tk_grid((n + 255) / 256, 8)), dim3(256), 0,
                     (hipStream_t)stream, (cf*)probe, (cf*)combined, (const cf*)mpu, beta,
                     inv_num_batch, n);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------------------ eigen-probe weights
// weights (B, C+1, S) rows of this minibatch, mode m.
// weights[n][0][m] += 0.1 stats[n][6] / stats[n][7]            (lstsq.py:721-738)
// norms[c-1] = sum_n weights[n][c][m]^2, c = 1..C      (probe.py:417-424; all-reduced
//                                                        by the caller across ranks)
__global__ __launch_bounds__(256) void eigen_weights0_kernel(float* __restrict__ weights,
                                                             const float* __restrict__ stats,
                                                             int B, int C, int S, int m,
                                                             float* __restrict__ norms) {
  __shared__ float red[4];
  const long row = (long)(C + 1) * S;
  for (int n = threadIdx.x; n < B; n += 256)
    weights[n * row + m] += 0.1f * stats[8 * n + 6] / stats[8 * n + 7];
  for (int c = 1; c <= C; ++c) {
    float a = 0.f;
    for (int n = threadIdx.x; n < B; n += 256) {
      const float w = weights[n * row + (long)c * S + m];
      a += w * w;
    }
    a = tk_block_sum256(a, red);
    if (threadIdx.x == 0) norms[c - 1] = a;
  }
}

extern "C" int tike_eigen_weights0(float* weights, const float* stats, int B, int C, int S,
                                   int m, float* norms, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(B >= 0 && C >= 0 && S >= 1 && m >= 0 && m < S);
  if (B == 0) return TK_OK;
  TK_CHECK_ARG(weights && stats && (C == 0 || norms));
  hipLaunchKernelGGL(eigen_weights0_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, weights,
                     stats, B, C, S, m, norms);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// pm[n] = (first[n] / P + weights[n][c][m]) / norm      (probe.py:429-433: the mean
// over pixels of the projection), the per-position factor of the pixel update
__global__ __launch_bounds__(256) void eigen_proj_mean_kernel(const float* __restrict__ first,
                                                              int first_stride,
                                                              const float* __restrict__ weights,
                                                              long row, const float* norm,
                                                              float inv_P, int B,
                                                              float* __restrict__ pm) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < B) pm[n] = (first[(long)n * first_stride] * inv_P + weights[n * row]) / norm[0];
}

extern "C" int tike_eigen_proj_mean(const float* first, int first_stride, const float* weights_c,
                                    long weights_row, const float* norm, long P, int B, float* pm,
                                    void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(B >= 0 && P > 0 && first_stride >= 1);
  if (B == 0) return TK_OK;
  TK_CHECK_ARG(first && weights_c && norm && pm);
  hipLaunchKernelGGL(eigen_proj_mean_kernel, dim3((B + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, first, first_stride, weights_c, weights_row, norm,
                     1.0f / (float)P, B, pm);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// E <- normalise(E + beta * u / mnorm(u)),  u = update / count  (probe.py:440-448;
// mnorm = sqrt(mean |.|^2)).  esum[0] = sum |E_new|^2.  Two launches over many
// workgroups instead of three dependent sweeps of one: the first forms
// sum |update|^2, sum |E|^2 and sum Re(conj(E) update), from which both norms
// follow (|E + k u|^2 = |E|^2 + 2 k Re(conj(E) u) + k^2 |u|^2); the second
// applies the update and the normalisation and accumulates esum.
// acc: 4 floats { sum|update|^2, sum|E|^2, sum Re(conj(E) update), esum }, zeroed.
__global__ __launch_bounds__(256) void eigen_normalise_sums_kernel(const cf* __restrict__ E,
                                                                   const cf* __restrict__ update,
                                                                   int npix,
                                                                   float* __restrict__ acc) {
  __shared__ float red[4];
  float uu = 0.f, ee = 0.f, eu = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    const cf e = E[i], u = update[i];
    uu += norm2(u);
    ee += norm2(e);
    eu += e.x * u.x + e.y * u.y;
  }
  uu = tk_block_sum256(uu, red);
  ee = tk_block_sum256(ee, red);
  eu = tk_block_sum256(eu, red);
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(&acc[0], uu);
    unsafeAtomicAdd(&acc[1], ee);
    unsafeAtomicAdd(&acc[2], eu);
  }
}

__global__ __launch_bounds__(256) void eigen_normalise_apply_kernel(cf* __restrict__ E,
                                                                    const cf* __restrict__ update,
                                                                    float inv_count, float beta,
                                                                    int npix,
                                                                    float* __restrict__ acc) {
  __shared__ float red[4];
  const float uu = acc[0], ee = acc[1], eu = acc[2];
  const float mu = sqrtf(uu * inv_count * inv_count / (float)npix);
  const float k = beta / mu * inv_count;
  const float inv = 1.0f / sqrtf((ee + 2.0f * k * eu + k * k * uu) / (float)npix);
  float c = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    const cf e = (E[i] + update[i] * k) * inv;
    E[i] = e;
    c += norm2(e);
  }
  c = tk_block_sum256(c, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(&acc[3], c);
}

__global__ void eigen_normalise_finish_kernel(const float* __restrict__ acc,
                                              float* __restrict__ esum) {
  esum[0] = acc[3];
}

extern "C" int tike_eigen_normalise(void* eigen, const void* update, double count, float beta,
                                    int npix, float* esum, float* work, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(npix >= 1 && eigen && update && work && count > 0);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(work, 0, 4 * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  const int grid = tk_deterministic() ? 1 : (npix >= 256 * 64 ? 64 : (npix + 255) / 256);
  hipLaunchKernelGGL(eigen_normalise_sums_kernel, dim3(grid), dim3(256), 0, st, (const cf*)eigen,
                     (const cf*)update, npix, work);
  hipLaunchKernelGGL(eigen_normalise_apply_kernel, dim3(grid), dim3(256), 0, st, (cf*)eigen,
                     (const cf*)update, (float)(1.0 / count), beta, npix, work);
  if (esum) hipLaunchKernelGGL(eigen_normalise_finish_kernel, dim3(1), dim3(1), 0, st, work, esum);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// sums (B, 5) from tike_eigen_position_sums.  Phase 1: dsum[0] = sum_n sums[n][2] / P
// (all-reduced by the caller).  Phase 2: weights[n][c][m] += (s1 / P) / (s2 / P + 0.1
// dsum / count); coefs[n][c-1] = (s3 + i s4) / esum      (probe.py:450-476,
// lstsq.py:740-761 projection coefficients).
__global__ __launch_bounds__(256) void eigen_dsum_kernel(const float* __restrict__ sums, int B,
                                                         float inv_P, float* __restrict__ dsum) {
  __shared__ float red[4];
  float a = 0.f;
  for (int n = threadIdx.x; n < B; n += 256) a += sums[5 * n + 2] * inv_P;
  a = tk_block_sum256(a, red);
  if (threadIdx.x == 0) dsum[0] = a;
}

__global__ __launch_bounds__(256) void eigen_weights_kernel(const float* __restrict__ sums, int B,
                                                            float inv_P,
                                                            const float* __restrict__ dsum,
                                                            float inv_count,
                                                            float* __restrict__ weights_c,
                                                            long row, cf* __restrict__ coefs_c,
                                                            int coef_stride,
                                                            const float* __restrict__ esum) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= B) return;
  const float* s = sums + 5 * n;
  const float d_mean = dsum[0] * inv_count;
  weights_c[n * row] += (s[1] * inv_P) / (s[2] * inv_P + 0.1f * d_mean);
  if (coefs_c) {
    const float inv = 1.0f / esum[0];
    coefs_c[(long)n * coef_stride] = mk(s[3] * inv, s[4] * inv);
  }
}

extern "C" int tike_eigen_dsum(const float* sums, int B, long P, float* dsum, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(B >= 0 && P > 0 && dsum && (B == 0 || sums));
  hipLaunchKernelGGL(eigen_dsum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, B,
                     1.0f / (float)P, dsum);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_eigen_weights(const float* sums, int B, long P, const float* dsum,
                                  double count, float* weights_c, long weights_row,
                                  void* coefs_c, int coef_stride, const float* esum,
                                  void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(B >= 0 && P > 0 && count > 0);
  if (B == 0) return TK_OK;
  TK_CHECK_ARG(sums && dsum && weights_c && (!coefs_c || esum));
  hipLaunchKernelGGL(eigen_weights_kernel, dim3((B + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, sums, B, 1.0f / (float)P, dsum, (float)(1.0 / count),
                     weights_c, weights_row, (cf*)coefs_c, coef_stride, esum);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------------- the packed minibatch tail
// The same arithmetic as the entries above, grouped so that a minibatch needs
// three launches between the step statistics and the next forward pass, and
// (several ranks) two small all-reduces besides the gradient's:
//   tike_lstsq_step_stats  ->  tike_eigen_pixel_update1 (+ sums3)
//     [all-reduce { sums3 ; update }]
//   tike_lstsq_tail_mid     2x2 solves; the eigen probe updated and normalised
//   tike_eigen_position_sums1
//     [all-reduce { sum step_o, sum step_p, dsum }]
//   tike_lstsq_tail_finish  weights, probe update, step lengths

// Workgroups [0, gridDim.x - 1): nacc[0..2] += { sum |update|^2, sum |E|^2,
// sum Re(conj(E) update) } (E == NULL: none).  Last workgroup: the 2x2 solves of
// the minibatch (step_solve_kernel), with sums3 = { sum (A1 + eps), sum (A4 +
// eps), sum cost } over ALL ranks: tail3[0..1] = { sum 0.9 max(0, Re x1),
// sum 0.9 max(0, Re x2) } over the local positions.
__global__ __launch_bounds__(256) void lstsq_tail_mid_kernel(
    const cf* __restrict__ E, const cf* __restrict__ update, int npix, float* __restrict__ nacc,
    const float* __restrict__ stats, int B, float eps, const float* __restrict__ sums3,
    float inv_count, int recover_psi, int recover_probe, float* __restrict__ tail3,
    float* __restrict__ part) {
  __shared__ float red[4];
  if (blockIdx.x + 1 < gridDim.x) {
    float uu = 0.f, ee = 0.f, eu = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += (gridDim.x - 1) * 256) {
      const cf e = E[i], u = update[i];
      uu += norm2(u);
      ee += norm2(e);
      eu += e.x * u.x + e.y * u.y;
    }
    uu = tk_block_sum256(uu, red);
    ee = tk_block_sum256(ee, red);
    eu = tk_block_sum256(eu, red);
    if (threadIdx.x == 0) {
      if (part != nullptr) {  // deterministic mode: added in workgroup order afterwards
        part[3 * blockIdx.x] = uu;
        part[3 * blockIdx.x + 1] = ee;
        part[3 * blockIdx.x + 2] = eu;
      } else {
        unsafeAtomicAdd(&nacc[0], uu);
        unsafeAtomicAdd(&nacc[1], ee);
        unsafeAtomicAdd(&nacc[2], eu);
      }
    }
    return;
  }
  const float r1 = 0.5f * sums3[0] * inv_count, r4 = 0.5f * sums3[1] * inv_count;
  float so = 0.f, sp = 0.f;
  for (int n = threadIdx.x; n < B; n += 256) {
    const float* s = stats + 8 * n;
    const float A1 = s[0] + eps + r1, A4 = s[1] + eps + r4;
    const float b1 = s[4], b2 = s[5];
    float x1 = 0.f, x2 = 0.f;
    if (recover_psi && recover_probe) {
      const float det = A1 * A4 - (s[2] * s[2] + s[3] * s[3]);
      x1 = -(s[2] * b2 - A4 * b1) / det;
      x2 = (A1 * b2 - s[2] * b1) / det;
    } else if (recover_psi) {
      x1 = b1 / A1;
    } else if (recover_probe) {
      x2 = b2 / A4;
    }
    so += 0.9f * fmaxf(x1, 0.f);
    sp += 0.9f * fmaxf(x2, 0.f);
  }
  so = tk_block_sum256(so, red);
  sp = tk_block_sum256(sp, red);
  if (threadIdx.x == 0) {
    tail3[0] = so;
    tail3[1] = sp;
  }
}

// E <- (E + k update) / mnorm(.), k = beta / mnorm(update / count) / count
// (probe.py:440-448), from nacc = { sum |update|^2, sum |E|^2, sum Re(conj(E) update) }
__global__ __launch_bounds__(256) void eigen_apply1_kernel(cf* __restrict__ E,
                                                           const cf* __restrict__ update,
                                                           const float* __restrict__ nacc,
                                                           float inv_count, float beta,
                                                           int npix) {
  const float uu = nacc[0], ee = nacc[1], eu = nacc[2];
  const float mu = sqrtf(uu * inv_count * inv_count / (float)npix);
  const float k = beta / mu * inv_count;
  const float inv = 1.0f / sqrtf((ee + 2.0f * k * eu + k * k * uu) / (float)npix);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256)
    E[i] = (E[i] + update[i] * k) * inv;
}

extern "C" int tike_lstsq_tail_mid(void* eigen0, const void* update, int npix, float* nacc,
                                   float beta_eigen, const float* stats, int B, float eps,
                                   const float* sums3, double count, int recover_psi,
                                   int recover_probe, float* tail3, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(B >= 0 && count > 0 && sums3 && tail3 && (B == 0 || stats));
  TK_CHECK_ARG(!eigen0 || (update && nacc && npix >= 1));
  // (deterministic mode: the summing workgroups leave their three sums in the
  // caller's scratch buffer, added in workgroup order by tk_ordered_sum; ONE
  // summing workgroup when there is no room)
  int grid = !eigen0 ? 0 : (npix >= 256 * 64 ? 64 : (npix + 255) / 256);
  float* part = nullptr;
  if (grid > 0 && tk_deterministic()) {
    part = tk_det_scratch(sizeof(float) * 3 * (size_t)grid);
    if (part == nullptr) grid = 1;
  }
  hipLaunchKernelGGL(lstsq_tail_mid_kernel, dim3(grid + 1), dim3(256), 0, (hipStream_t)stream,
                     (const cf*)eigen0, (const cf*)update, npix, nacc, stats, B, eps, sums3,
                     (float)(1.0 / count), recover_psi, recover_probe, tail3, part);
  if (part != nullptr) {
    int rc = tk_ordered_sum(nacc, part, 3, grid, true, (hipStream_t)stream);
    if (rc) return rc;
  }
  if (eigen0)
    hipLaunchKernelGGL(eigen_apply1_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (cf*)eigen0, (const cf*)update, nacc, (float)(1.0 / count), beta_eigen,
                       npix);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// Everything that closes a minibatch, element-wise over max(probe, pixels,
// positions), with tail3 = { sum step_o, sum step_p, dsum } over ALL ranks:
//   steps[0..4] = { tail3[0], tail3[1], beta_object, beta_probe, mean cost }
//   probe += beta_probe mpu;  combined += beta_probe mpu / num_batch   (lstsq.py:177-181)
//   weights[n][0][m] += 0.1 stats[n][6] / stats[n][7]               (lstsq.py:721-738)
//   weights[n][1][m] += (s1 / P) / (s2 / P + 0.1 dsum / count)      (probe.py:450-476)
__global__ __launch_bounds__(256) void lstsq_tail_finish_kernel(
    const float* __restrict__ tail3, const float* __restrict__ sums3, float inv_count,
    float* __restrict__ steps, cf* __restrict__ probe, cf* __restrict__ combined,
    const cf* __restrict__ mpu, float inv_num_batch, long nprobe, float* __restrict__ weights,
    long row, int S, int m, const float* __restrict__ stats, const float* __restrict__ sums5,
    int B, int npix) {
  const float bp = tail3[1] * inv_count;
  const long stride = (long)gridDim.x * 256;
  const long i0 = blockIdx.x * 256L + threadIdx.x;
  if (i0 == 0) {
    steps[0] = tail3[0];
    steps[1] = tail3[1];
    steps[2] = tail3[0] * inv_count;
    steps[3] = bp;
    steps[4] = sums3[2] * inv_count;
  }
  if (probe != nullptr) {
    for (long i = i0; i < nprobe; i += stride) {
      const cf d = mpu[i] * bp;
      probe[i] = probe[i] + d;
      if (combined) combined[i] = combined[i] + d * inv_num_batch;
    }
  }
  if (weights != nullptr) {
    const float inv_P = 1.0f / (float)npix;
    const float d_mean = sums5 ? tail3[2] * inv_count : 0.f;
    for (long n = i0; n < B; n += stride) {
      weights[n * row + m] += 0.1f * stats[8 * n + 6] / stats[8 * n + 7];
      if (sums5) {
        const float* s = sums5 + 5 * n;
        weights[n * row + S + m] += (s[1] * inv_P) / (s[2] * inv_P + 0.1f * d_mean);
      }
    }
  }
}

extern "C" int tike_lstsq_tail_finish(const float* tail3, const float* sums3, double count,
                                      float* steps, void* probe, void* combined, const void* mpu,
                                      float inv_num_batch, long nprobe, float* weights,
                                      long weights_row, int S, int m, const float* stats,
                                      const float* sums5, int B, int npix, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(tail3 && sums3 && steps && count > 0 && B >= 0 && nprobe >= 0 && npix >= 0);
  TK_CHECK_ARG(!probe || mpu);
  if (B == 0) {  // an empty share of the minibatch: no per-position rows to update
    weights = nullptr;
    sums5 = nullptr;
  }
  TK_CHECK_ARG(!weights || (stats && weights_row >= 1 && S >= 1 && m >= 0 && m < S));
  TK_CHECK_ARG(!sums5 || (weights && weights_row >= 2L * S && npix >= 1));
  long work = 1;
  if (probe && nprobe > work) work = nprobe;
  if (weights && B > work) work = B;
  hipLaunchKernelGGL(lstsq_tail_finish_kernel, dim3(tk_grid((work + 255) / 256, 8)), dim3(256), 0,
                     (hipStream_t)stream, tail3, sums3, (float)(1.0 / count), steps, (cf*)probe,
                     (cf*)combined, (const cf*)mpu, inv_num_batch, nprobe, weights, weights_row,
                     S, m, stats, sums5, B, npix);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------- the whole chunk body in one call
// (lstsq.py:422-579): the five stage entries, stream-ordered.
extern "C" int tike_lstsq_chunk_gradients(
    const void* psi, const float* scan, const void* probe, const void* eigen_probe,
    const float* eigen_weights, int num_eigen, int eigen_modes, const void* data, int data_u16,
    const unsigned char* measured, int model, float unmeasured_scaling, long num_measured,
    void* scratch, void* work, float* gscale, void* patches, float* costs, void* objproj,
    void* chi0, void* m_probe_update, float mpu_scale, float* object_acc, int nscan, int S,
    int det, int H, int W, float fwd_scale, float inv_scale, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && H >= 1 && W >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(psi && scan && probe && data && scratch && work && patches);
  TK_CHECK_ARG(scratch != work && (object_acc == nullptr) == (objproj == nullptr));
  if (!((det == 128 && S <= 8) || (det == 256 && S <= 8) || (det == 512 && S <= 4)))
    return TK_ERR_UNSUPPORTED;
  TK_CHECK_ARG(gscale || det == 256);  // 256^2: the factor stays in registers
  if (det == 128) {
    // the far plane is kept at this size (whole tile in LDS): forward +
    // intensity -> gradient factor + costs -> scaled inverse pass 1 -> pass 2 +
    // gradients -> scatter.  gscale holds TWO (nscan,det,det) tables here: the
    // factor, then the intensity.  float32 data only.
    if (data_u16 || (eigen_weights && eigen_modes > 0)) return TK_ERR_UNSUPPORTED;
    float* inten = gscale + (size_t)nscan * det * det;
    int rc = tike_ptycho_fwd_intensity(psi, scan, probe, 0, nullptr, eigen_weights, num_eigen,
                                       eigen_modes, scratch, inten, patches, nscan, S, det, det,
                                       H, W, fwd_scale, stream);
    if (rc) return rc;
    rc = tike_gradient_scale(inten, (const float*)data, measured, gscale, costs, nscan, det,
                             model, unmeasured_scaling, num_measured, stream);
    if (rc) return rc;
    rc = tike_ifft2_pass1_scaled(scratch, gscale, nullptr, nullptr, S, work, (long)nscan * S, det,
                                 stream);
    if (rc) return rc;
    rc = tike_ifft2_pass2_gradients(work, patches, probe, eigen_probe, eigen_weights, num_eigen,
                                    eigen_modes, objproj, chi0, m_probe_update, mpu_scale, nscan,
                                    S, det, inv_scale, stream);
    if (rc) return rc;
    if (object_acc) rc = tike_scatter_patches(objproj, scan, object_acc, nscan, det, H, W, stream);
    return rc;
  }
  int rc = tike_fwd_pass1(psi, scan, probe, 0, nullptr, eigen_probe, eigen_weights, num_eigen,
                          eigen_modes, scratch, patches, nscan, S, det, det, H, W, stream);
  if (rc) return rc;
  if (det == 256) {
    rc = tike_fwd_grad_ifft2_pass1(scratch, data, data_u16, measured, costs, work, nscan, S, det,
                                   fwd_scale, model, unmeasured_scaling, num_measured, stream);
  } else {
    rc = tike_fwd_gradient_scale(scratch, data, data_u16, measured, gscale, nullptr, costs,
                                 nullptr, nscan, S, det, fwd_scale, model, unmeasured_scaling,
                                 num_measured, stream);
    if (rc) return rc;
    rc = tike_grad_ifft2_pass1(scratch, gscale, nullptr, nullptr, S, work, (long)nscan * S, det,
                               fwd_scale, stream);
  }
  if (rc) return rc;
  rc = tike_ifft2_pass2_gradients(work, patches, probe, eigen_probe, eigen_weights, num_eigen,
                                  eigen_modes, objproj, chi0, m_probe_update, mpu_scale, nscan,
                                  S, det, inv_scale, stream);
  if (rc) return rc;
  if (object_acc) rc = tike_scatter_patches(objproj, scan, object_acc, nscan, det, H, W, stream);
  return rc;
}

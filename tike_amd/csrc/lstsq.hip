// Kernels of the least-squares + gradient update loop (lstsq_grad) for gfx950.
//
// Reference: src/tike/ptycho/solvers/lstsq.py (one minibatch):
//   :506-520  object gradient  sum_s conj(P_n,s) chi_n,s scattered   -> tike_object_grad
//   :524-539  probe gradient   sum_n conj(O_n) chi_n,s               -> tike_probe_grad
//   :619-718  step-size normal equations, per position               -> tike_lstsq_step_stats
//   :721-738  eigen-probe intensity coefficients                     -> (same kernel)
//   solvers/_preconditioner.py:116-167 probe preconditioner          -> tike_probe_preconditioner
// where chi is the exit-wave update (IFFT2 of the far-plane gradient cropped
// to the probe window), P_n,s the probe at position n (shared probe plus
// eigen probes synthesised on the fly) and O_n the bilinear object patch.
#include "internal.h"
#include "tike_amd.h"

// ----------------------------------------------------------- object gradient
extern "C" int tike_object_grad(const void* chi, const float* scan, const void* probe,
                                int probe_per_scan, const void* eigen_probe,
                                const float* eigen_weights, int num_eigen, int eigen_modes,
                                void* object_upd_sum, int nscan, int S, int pw, int H, int W,
                                void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(!(eigen_weights && probe_per_scan));
  return tk_conv_adj((const cf*)chi, scan,
                     tk_make_probe(probe, probe_per_scan, eigen_probe, eigen_weights, num_eigen,
                                   eigen_modes, S, pw),
                     (cf*)object_upd_sum, nscan, S, pw, pw, H, W, (hipStream_t)stream);
}

// ------------------------------------------------------------ probe gradient
// One thread per probe pixel, a workgroup walks a chunk of positions keeping S
// complex accumulators in registers; one atomic pair per (pixel, mode, chunk).
// Optionally stores the object patches (B, pw, pw) for later passes.
constexpr int TK_MAX_MODES = 16;

template <bool WITH_CHI>
__global__ __launch_bounds__(256) void probe_grad_kernel(
    const cf* __restrict__ chi, const float* __restrict__ scan, const cf* __restrict__ psi,
    cf* __restrict__ patches, float* __restrict__ out, int nscan, int S, int pw, int H, int W,
    int chunk) {
  const long P = (long)pw * pw;
  const long total = (long)H * W;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b0 = blockIdx.y * chunk;
  const int b1 = min(nscan, b0 + chunk);
  if (p >= P) return;
  const int py = (int)(p / pw), px = (int)(p % pw);
  cf acc[TK_MAX_MODES];
#pragma unroll
  for (int s = 0; s < TK_MAX_MODES; ++s) acc[s] = mk(0.f, 0.f);
  for (int b = b0; b < b1; ++b) {
    const TkCorner c = tk_corner(scan, b);
    const int y = c.sy + py, x = c.sx + px;
    cf o = mk(0.f, 0.f);
    if (y >= 0 && y < H && x >= 0 && x < W) o = tk_gather(psi, (long)y * W + x, W, total, c);
    if (patches) patches[b * P + p] = o;
    if (WITH_CHI) {
      const cf oc = conjf(o);
#pragma unroll
      for (int s = 0; s < TK_MAX_MODES; ++s)
        if (s < S) acc[s] = acc[s] + oc * chi[((long)b * S + s) * P + p];
    } else {
      acc[0].x += norm2(o);
    }
  }
  if (WITH_CHI) {
#pragma unroll
    for (int s = 0; s < TK_MAX_MODES; ++s)
      if (s < S) {
        unsafeAtomicAdd(&out[2 * (s * P + p)], acc[s].x);
        unsafeAtomicAdd(&out[2 * (s * P + p) + 1], acc[s].y);
      }
  } else {
    unsafeAtomicAdd(&out[2 * p], acc[0].x);
  }
}

static int probe_chunk(int nscan) {
  // enough position chunks to fill the chip, at least 8 positions each
  int chunk = (nscan + 31) / 32;
  return chunk < 8 ? 8 : chunk;
}

extern "C" int tike_probe_grad(const void* chi, const float* scan, const void* psi,
                               void* patches, void* m_probe_update, int nscan, int S, int pw,
                               int H, int W, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && S <= TK_MAX_MODES && pw >= 1 && H >= 1 && W >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(chi && scan && psi && m_probe_update);
  const long P = (long)pw * pw;
  const int chunk = probe_chunk(nscan);
  dim3 grid((unsigned)((P + 255) / 256), (unsigned)((nscan + chunk - 1) / chunk));
  hipLaunchKernelGGL((probe_grad_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream,
                     (const cf*)chi, scan, (const cf*)psi, (cf*)patches, (float*)m_probe_update,
                     nscan, S, pw, H, W, chunk);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// probe preconditioner: out (pw,pw) complex (imaginary part untouched) +=
// sum_n |patch_n(psi)|^2   (_preconditioner.py:136-144)
extern "C" int tike_probe_preconditioner(const float* scan, const void* psi, void* out,
                                         int nscan, int pw, int H, int W, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && pw >= 1 && H >= 1 && W >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(scan && psi && out);
  const long P = (long)pw * pw;
  const int chunk = probe_chunk(nscan);
  dim3 grid((unsigned)((P + 255) / 256), (unsigned)((nscan + chunk - 1) / chunk));
  hipLaunchKernelGGL((probe_grad_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream,
                     (const cf*)nullptr, scan, (const cf*)psi, (cf*)nullptr, (float*)out, nscan,
                     1, pw, H, W, chunk);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------------- step-size normal equations
// One workgroup per position.  With m = 0 (lstsq.py:169):
//   dOP = patch_n(g) * P_n,0        g = preconditioned object update
//   dPO = mpu_0 * O_n               mpu = common probe update, O_n = patch_n(psi)
//   stats[n] = { sum|dOP|^2, sum|dPO|^2, Re sum dOP conj(dPO), Im sum dOP conj(dPO),
//                sum Re(conj(dOP) chi_n,0), sum Re(conj(dPO) chi_n,0),
//                sum Re(conj(O_n P_0) chi_n,0), sum |O_n P_0|^2 }
// (the last two feed _get_coefs_intensity, lstsq.py:721-738, which uses the
// SHARED probe P_0).  eps terms (:641,661,667) are added by the solver.
__global__ __launch_bounds__(256) void step_stats_kernel(
    const cf* __restrict__ chi, const float* __restrict__ scan, const cf* __restrict__ psi,
    const cf* __restrict__ gobj, const TkProbe probe, const cf* __restrict__ mpu,
    float* __restrict__ stats, int nscan, int chi_modes, int pw, int H, int W) {
  __shared__ float red[4];
  const long P = (long)pw * pw;
  const long total = (long)H * W;
  for (int n = blockIdx.x; n < nscan; n += gridDim.x) {
    const TkCorner c = tk_corner(scan, n);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long p = threadIdx.x; p < P; p += blockDim.x) {
      const int py = (int)(p / pw), px = (int)(p % pw);
      const int y = c.sy + py, x = c.sx + px;
      cf o = mk(0.f, 0.f), g = mk(0.f, 0.f);
      if (y >= 0 && y < H && x >= 0 && x < W) {
        const long ii = (long)y * W + x;
        o = tk_gather(psi, ii, W, total, c);
        if (gobj) g = tk_gather(gobj, ii, W, total, c);
      }
      const cf x0 = chi[((long)n * chi_modes) * P + p];
      const cf dOP = g * probe.at(n, 0, p);
      const cf dPO = mpu ? mpu[p] * o : mk(0.f, 0.f);
      const cf OP = o * probe.probe[p];
      a[0] += norm2(dOP);
      a[1] += norm2(dPO);
      const cf a2 = dOP * conjf(dPO);
      a[2] += a2.x;
      a[3] += a2.y;
      a[4] += dOP.x * x0.x + dOP.y * x0.y;
      a[5] += dPO.x * x0.x + dPO.y * x0.y;
      a[6] += OP.x * x0.x + OP.y * x0.y;
      a[7] += norm2(OP);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float v = tk_block_sum256(a[k], red);
      if (threadIdx.x == 0) stats[(long)n * 8 + k] = v;
    }
  }
}

extern "C" int tike_lstsq_step_stats(const void* chi, const float* scan, const void* psi,
                                     const void* object_update_precond, const void* probe,
                                     const void* eigen_probe, const float* eigen_weights,
                                     int num_eigen, int eigen_modes, const void* m_probe_update,
                                     float* stats, int nscan, int S, int chi_modes, int pw,
                                     int H, int W, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && chi_modes >= 1 && pw >= 1 && H >= 1 && W >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(chi && scan && psi && probe && stats);
  hipLaunchKernelGGL(step_stats_kernel, dim3(tk_grid(nscan, 16)), dim3(256), 0,
                     (hipStream_t)stream, (const cf*)chi, scan, (const cf*)psi,
                     (const cf*)object_update_precond,
                     tk_make_probe(probe, 0, eigen_probe, eigen_weights, num_eigen, eigen_modes, S,
                                   pw),
                     (const cf*)m_probe_update, stats, nscan, chi_modes, pw, H, W);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

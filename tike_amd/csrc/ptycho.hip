// Fused ptychography forward / adjoint kernels for gfx950.
//
//   tike_ptycho_fwd      bilinear patch gather * probe -> zero pad -> FFT2
//                        (reference ptycho.py:114-129 = convolution.py:58-101
//                        + propagation.py:43-57) in ONE kernel: no patch
//                        array, no memset, no separate multiply pass.
//   tike_ifft2_crop      IFFT2 -> crop to the probe window
//                        (propagation.py:59-73 + lstsq.py:504-507).
//   tike_farplane_gradient  intensity, per-pattern cost and the far-plane
//                        gradient in one pass (ptycho.py:18-23,
//                        objective.py:11-124, lstsq.py:444-502).
//
// One workgroup owns one (position, mode) tile and runs the row pass and the
// column pass back to back; the intermediate lives in the tile's own output
// (L2 / Infinity Cache resident between the passes).
#include "fft_engine2.h"
#include "internal.h"
#include "tike_amd.h"

// --------------------------------------------------------------- forward
template <int N>
__global__ __launch_bounds__(FftPlan<N>::NT, FftPlan<N>::MINW) void ptycho_fwd_kernel(
    const cf* __restrict__ psi, const float* __restrict__ scan, const TkProbe probe,
    cf* __restrict__ farplane, long ntile, int S, int pw, int H, int W, float scale,
    const cf* __restrict__ twtab) {
  using G = FftGeom<N>;
  __shared__ cf lds[G::LDS_ELEMS];
  FftTw<N> tw;
  const int pad = (N - pw) / 2;
  const int end = pad + pw;
  const long total = (long)H * W;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const long n = tile / S;
    const int s = (int)(tile % S);
    const TkCorner c = tk_corner(scan, n);
    cf* __restrict__ dst = farplane + tile * (long)N * N;
    const FftLane<N, false> row = fft_lane<N, false>();
    tw.init(twtab, row.j);
    for (int g = 0; g < N; g += G::L) {
      if (g + G::L <= pad || g >= end) {
        // rows entirely inside the zero padding transform to zero
        for (int i = threadIdx.x; i < G::L * N; i += G::NT) dst[g * N + i] = mk(0.f, 0.f);
        continue;
      }
      // Stage the L input rows (patch * probe, zero padded) into the LDS line
      // buffers with row-contiguous, fully coalesced loads: thread -> column,
      // loop over rows, re-using the lower taps of one row as the upper taps
      // of the next (2 new object loads per pixel instead of 4).
      for (int x0 = 0; x0 < N; x0 += G::NT) {
        const int dx = x0 + threadIdx.x;
        const int px = dx - pad;
        const int x = c.sx + px;
        const bool col_ok = dx < N && px >= 0 && px < pw && x >= 0 && x < W;
        cf t0 = mk(0.f, 0.f), t1 = mk(0.f, 0.f);  // taps of the current object row
        bool have = false;
#pragma unroll 4
        for (int line = 0; line < G::L; ++line) {
          const int py = g + line - pad;
          const int y = c.sy + py;
          cf o = mk(0.f, 0.f);
          if (col_ok && py >= 0 && py < pw && y >= 0 && y < H) {
            const long ii = (long)y * W + x;
            if (!have) {
              t0 = psi[ii];
              t1 = (ii + 1 < total) ? psi[ii + 1] : mk(0.f, 0.f);
            }
            cf b0 = mk(0.f, 0.f), b1 = mk(0.f, 0.f);
            // lower taps are loaded whenever they lie inside the allocation
            // (they become the next row's upper taps); a zero weight makes
            // their contribution exactly zero, as in the reference kernel.
            if (ii + W < total) {
              b0 = psi[ii + W];
              if (ii + W + 1 < total) b1 = psi[ii + W + 1];
            }
            cf v = mk(t0.x * c.w00, t0.y * c.w00);
            v.x += t1.x * c.w01;
            v.y += t1.y * c.w01;
            v.x += b0.x * c.w10;
            v.y += b0.y * c.w10;
            v.x += b1.x * c.w11;
            v.y += b1.y * c.w11;
            o = v * probe.at(n, s, (long)py * pw + px);
            t0 = b0;
            t1 = b1;
            have = true;
          } else {
            have = false;
          }
          if (dx < N) lds[line * G::LS + tk_pad16(dx)] = o;
        }
      }
      __syncthreads();
      fft_lines<N, false, false>(
          lds, row, tw,
          [&](int line, int e) { return lds[line * G::LS + tk_pad16(e)]; },
          [&](int line, int e, cf v) { dst[(g + line) * N + e] = v; }, /*sync_after_load=*/true);
    }
    __syncthreads();
    const FftLane<N, true> col = fft_lane<N, true>();
    tw.init(twtab, col.j);
    for (int g = 0; g < N; g += G::L) {
      fft_lines<N, false, true>(
          lds, col, tw, [&](int line, int e) { return dst[e * N + g + line]; },
          [&](int line, int e, cf v) { dst[e * N + g + line] = v * scale; });
    }
    __syncthreads();
  }
}

// v2 structure (fft_engine2.h): N threads per workgroup, thread = column.
template <int N>
__global__ __launch_bounds__(N, TK_V2_MINW(N)) void ptycho_fwd_v2_kernel(
    const cf* __restrict__ psi, const float* __restrict__ scan, const TkProbe probe,
    cf* __restrict__ farplane, long ntile, int S, int pw, int H, int W, float scale,
    const cf* __restrict__ twtab) {
  using G2 = Fft2Geom<N>;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int pad = (N - pw) / 2;
  const long total = (long)H * W;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const long n = tile / S;
    const int s = (int)(tile % S);
    const TkCorner c = tk_corner(scan, n);
    cf* __restrict__ dst = farplane + tile * (long)N * N;
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    const int t = threadIdx.x;
    const int px = t - pad;
    const int x = c.sx + px;
    const bool col_ok = px >= 0 && px < pw && x >= 0 && x < W;
    // clamped coordinates: every load is unconditional and in bounds, the
    // padding / out-of-image pixels are zeroed by a select afterwards
    const int pxc = px < 0 ? 0 : (px >= pw ? pw - 1 : px);
    const int xc = c.sx + pxc < 0 ? 0 : (c.sx + pxc >= W ? W - 1 : c.sx + pxc);
    // Fast path (every position that passes check_allowed_positions): all four
    // taps of every patch pixel lie inside the image, so rows are addressed as
    // uniform row offset + per-thread column with no clamping, and the probe
    // weights of this (position, mode) are hoisted into scalars.
    const bool interior = c.sy >= 0 && c.sx >= 0 && c.sy + pw < H && c.sx + pw < W;
    const long PP = (long)pw * pw;
    const cf* __restrict__ Pn = probe.probe + n * probe.pos_stride + s * PP;
    float w0 = 1.0f;
    int nE = 0;
    if (probe.weights != nullptr) {
      w0 = probe.weights[n * (long)(probe.C + 1) * probe.S + s];
      if (probe.eigen != nullptr && s < probe.Sm) nE = probe.C;
    }
    for (int r = 0; r < G2::RB; ++r) {
      // stage the 16 rows {r + RB*l} of patch * probe (zero padded) into LDS
      if (interior) {
#pragma unroll 8
        for (int l = 0; l < 16; ++l) {
          const int py = r + G2::RB * l - pad;           // uniform
          const bool row_ok = py >= 0 && py < pw;        // uniform
          const int pyc = py < 0 ? 0 : (py >= pw ? pw - 1 : py);
          const cf* __restrict__ q = psi + (long)(c.sy + pyc) * W + c.sx + pxc;
          const cf a = q[0], b = q[1], d = q[W], e = q[W + 1];
          const long pi = (long)pyc * pw + pxc;
          cf pr = Pn[pi] * w0;
          for (int k = 0; k < nE; ++k) {
            const cf ev = probe.eigen[((long)k * probe.Sm + s) * PP + pi];
            const float wk =
                probe.weights[n * (long)(probe.C + 1) * probe.S + (k + 1) * probe.S + s];
            pr.x += wk * ev.x;
            pr.y += wk * ev.y;
          }
          cf o = mk(a.x * c.w00, a.y * c.w00);
          o.x += b.x * c.w01;
          o.y += b.y * c.w01;
          o.x += d.x * c.w10;
          o.y += d.y * c.w10;
          o.x += e.x * c.w11;
          o.y += e.y * c.w11;
          o = o * pr;
          lds[l * G2::LS + tk_pad16(t)] = (row_ok && col_ok) ? o : mk(0.f, 0.f);
        }
      } else {
#pragma unroll 4
        for (int l = 0; l < 16; ++l) {
          const int py = r + G2::RB * l - pad;
          const int y = c.sy + py;
          const bool ok = col_ok && py >= 0 && py < pw && y >= 0 && y < H;
          const int pyc = py < 0 ? 0 : (py >= pw ? pw - 1 : py);
          const int yc = c.sy + pyc < 0 ? 0 : (c.sy + pyc >= H ? H - 1 : c.sy + pyc);
          const cf o = tk_gather(psi, (long)yc * W + xc, W, total, c) *
                       probe.at(n, s, (long)pyc * pw + pxc);
          lds[l * G2::LS + tk_pad16(t)] = ok ? o : mk(0.f, 0.f);
        }
      }
      __syncthreads();
      fft2_pass1<N, false>(lds, twtab, tw, line, j, r,
                           [&](int y, int e, auto) { return lds[line * G2::LS + tk_pad16(e)]; }, dst);
    }
    __syncthreads();
    for (int k1 = 0; k1 < 16; ++k1)
      fft2_pass2<N, false>(dst, k1, [&](int ky, int tt, cf v) { dst[ky * N + tt] = v * scale; });
    __syncthreads();
  }
}

// Position-major forward: one workgroup owns ALL S modes of a position, so the
// bilinear patch is gathered once per 16-row group (registers) and re-used by
// every mode, and the intensity sum_s |F_s|^2 accumulates in registers during
// pass 2 -- the far-plane is never re-read to form it (ptycho.py:18-23,
// lstsq.py:444-447).
#ifndef TK_POS_WAVES
#define TK_POS_WAVES 2
#endif
// Optional epilogue of the intensity-only forward kernel: the far-plane
// gradient factor and the per-pattern cost (objective.py:11-124,
// lstsq.py:444-502) straight from the intensity in registers.
struct TkGradScale {
  const float* data;           // (nscan, det, det) or nullptr: epilogue off
  const unsigned char* mask;   // (det, det) or nullptr (all measured)
  float* gscale;               // (nscan, det, det)
  float* costs;                // (nscan) or nullptr
  int model;                   // 0 gaussian, 1 poisson
  float unmeasured_scaling;
  float inv_nmeasured;
};

// STORE = false: the far-plane waves are formed in registers for the intensity
// only; `farplane` then keeps the INPUT of the column pass (rows 16r + k1 of
// fft2_pass1), which tike_grad_ifft2_crop consumes.
template <int N, bool STORE = true>
__global__ __launch_bounds__(N, (N <= 256 ? TK_POS_WAVES : 2)) void ptycho_fwd_pos_kernel(
    const cf* __restrict__ psi, const float* __restrict__ scan, const TkProbe probe,
    cf* __restrict__ farplane, float* __restrict__ intensity, int nscan, int S, int pw, int H,
    int W, float scale, const cf* __restrict__ twtab, const TkGradScale gsc,
    cf* __restrict__ patches) {
  using G2 = Fft2Geom<N>;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int pad = (N - pw) / 2;
  const long total = (long)H * W;
  const long PP = (long)pw * pw;
  for (long n = blockIdx.x; n < nscan; n += gridDim.x) {
    const TkCorner c = tk_corner(scan, n);
    cf* __restrict__ dst0 = farplane + n * S * (long)N * N;
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    const int t = threadIdx.x;
    for (int r = 0; r < G2::RB; ++r) {
      // Patch values of row y = r + RB*line in the FFT register layout
      // (element e = j + i*T), gathered once and shared by all S modes: the
      // row FFT consumes them straight from registers (no LDS staging pass).
      const int py = r + G2::RB * line - pad;
      const int y = c.sy + py;
      const bool row_ok = py >= 0 && py < pw && y >= 0 && y < H;
      const int pyc = py < 0 ? 0 : (py >= pw ? pw - 1 : py);
      const int yc = c.sy + pyc < 0 ? 0 : (c.sy + pyc >= H ? H - 1 : c.sy + pyc);
      cf pv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int px = j + i * G2::T - pad;
        const int x = c.sx + px;
        const bool ok = row_ok && px >= 0 && px < pw && x >= 0 && x < W;
        const int pxc = px < 0 ? 0 : (px >= pw ? pw - 1 : px);
        const int xc = c.sx + pxc < 0 ? 0 : (c.sx + pxc >= W ? W - 1 : c.sx + pxc);
        const cf o = tk_gather(psi, (long)yc * W + xc, W, total, c);
        pv[i] = ok ? o : mk(0.f, 0.f);
        // bound the taps in flight (4 elements = 16 loads) and with them the
        // register footprint of this phase
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      if (patches != nullptr && py >= 0 && py < pw) {
        // the object patch O_n (Patch.fwd, lstsq.py:524-531) for the gradient
        // and step-size passes, while it is in registers
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int px = j + i * G2::T - pad;
          if (px >= 0 && px < pw) patches[n * PP + (long)py * pw + px] = pv[i];
        }
      }
      for (int s = 0; s < S; ++s) {
        // probe of this (position, mode): a base pointer and a scale, both
        // uniform -- either the shared probe times its weight or the varying
        // probe synthesised beforehand by tike_varying_probe
        const cf* __restrict__ Pn = probe.probe + n * probe.pos_stride + s * PP;
        float w0 = 1.0f;
        if (probe.weights != nullptr) {
          if (probe.unique != nullptr && s < probe.Sm)
            Pn = probe.unique + (n * probe.Sm + s) * PP;
          else
            w0 = probe.weights[n * (long)(probe.C + 1) * probe.S + s];
        }
        fft2_pass1<N, false>(
            lds, twtab, tw, line, j, r,
            [&](int, int e, auto I) {
              constexpr int i = decltype(I)::value;
              const int px = e - pad;
              const int pxc = px < 0 ? 0 : (px >= pw ? pw - 1 : px);
              return pv[i] * (Pn[pyc * pw + pxc] * w0);
            },
            dst0 + s * (long)N * N);
      }
    }
    __syncthreads();
    if constexpr (!STORE) {
      // intensity only: nothing is stored per tile, so the column pass is a
      // pure read stream -- keep the rows of the NEXT (k1, mode) in flight
      // while the current ones go through the butterfly
      cf nxt[G2::RB];
      float cost = 0.f;
#pragma unroll
      for (int r = 0; r < G2::RB; ++r) nxt[r] = dst0[(16 * r) * N + t];
      for (int k1 = 0; k1 < 16; ++k1) {
        float I[G2::RB];
#pragma unroll
        for (int k2 = 0; k2 < G2::RB; ++k2) I[k2] = 0.f;
        for (int s = 0; s < S; ++s) {
          cf u[G2::RB];
#pragma unroll
          for (int r = 0; r < G2::RB; ++r) u[r] = nxt[r];
          const int s2 = s + 1 < S ? s + 1 : 0;
          const int k2n = s + 1 < S ? k1 : (k1 + 1 < 16 ? k1 + 1 : k1);
          const cf* __restrict__ nsrc = dst0 + s2 * (long)N * N;
#pragma unroll
          for (int r = 0; r < G2::RB; ++r) nxt[r] = nsrc[(16 * r + k2n) * N + t];
          Dft<G2::RB, false>::run(u);
#pragma unroll
          for (int k2 = 0; k2 < G2::RB; ++k2) I[k2] += norm2(u[k2] * scale);
        }
        if (intensity) {
#pragma unroll
          for (int k2 = 0; k2 < G2::RB; ++k2)
            tk_st_stream(intensity + n * (long)N * N + (k1 + 16 * k2) * N + t, I[k2]);
        }
        if (gsc.data) {
          // gradient factor and cost from the intensity in registers
#pragma unroll
          for (int k2 = 0; k2 < G2::RB; ++k2) {
            const long p = (long)(k1 + 16 * k2) * N + t;
            float g = gsc.unmeasured_scaling - 1.0f;
            if (gsc.mask == nullptr || gsc.mask[p]) {
              const float dv = gsc.data[n * (long)N * N + p];
              if (gsc.model == 0) {
                const float sI = sqrtf(I[k2]), sd = sqrtf(dv);
                const float diff = sI - sd;
                cost += diff * diff;
                g = -(1.0f - sd / (sI + 1e-9f));
              } else {
                cost += I[k2] - dv * logf(I[k2] + 1e-9f);
                g = -(1.0f - dv / (I[k2] + 1e-9f));
              }
            }
            gsc.gscale[n * (long)N * N + p] = g;
          }
        }
      }
      if (gsc.data && gsc.costs) {
        // block sum through the (now idle) FFT exchange area
        float* red = reinterpret_cast<float*>(lds);
        __syncthreads();
        cost = tk_wave_sum(cost);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cost;
        __syncthreads();
        if (threadIdx.x == 0) {
          float tot = 0.f;
          for (int w = 0; w < N / 64; ++w) tot += red[w];
          gsc.costs[n] = tot * gsc.inv_nmeasured;
        }
      }
    } else
    for (int k1 = 0; k1 < 16; ++k1) {
      float I[G2::RB];
#pragma unroll
      for (int k2 = 0; k2 < G2::RB; ++k2) I[k2] = 0.f;
      for (int s = 0; s < S; ++s) {
        cf* __restrict__ dst = dst0 + s * (long)N * N;
        fft2_pass2<N, false>(dst, k1, [&](int ky, int tt, cf v) {
          const cf o = v * scale;
          if (STORE) tk_st_stream(dst + ky * N + tt, o);
          I[(ky - k1) >> 4] += norm2(o);
        });
      }
      if (intensity) {
#pragma unroll
        for (int k2 = 0; k2 < G2::RB; ++k2)
          tk_st_stream(intensity + n * (long)N * N + (k1 + 16 * k2) * N + t, I[k2]);
      }
    }
    __syncthreads();
  }
}

template <int N, bool STORE = true>
static int launch_fwd_pos(const cf* psi, const float* scan, const TkProbe& probe, cf* farplane,
                          float* intensity, int nscan, int S, int pw, int H, int W, float scale,
                          hipStream_t stream, const TkGradScale* gsc = nullptr,
                          cf* patches = nullptr) {
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
  TkGradScale g = {};
  if (gsc) g = *gsc;
  hipLaunchKernelGGL((ptycho_fwd_pos_kernel<N, STORE>), dim3(tk_grid(nscan, 4)), dim3(N), 0,
                     stream, psi, scan, probe, farplane, intensity, nscan, S, pw, H, W, scale, tw,
                     g, patches);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

#ifndef TK_LDS128_MAX_MODES
#define TK_LDS128_MAX_MODES 8
#endif
constexpr int TK_FG_PIX = 1024;  // pixels per workgroup of the stored-far-plane cost kernels
static int launch_fwd128_lds(const cf* psi, const float* scan, const TkProbe& probe, cf* farplane,
                             float* intensity, int nscan, int S, int H, int W, float scale,
                             hipStream_t stream, cf* patches, const int* skip = nullptr);
static int tk_farplane_gradient(void* farplane, const float* data, const unsigned char* measured,
                                float* intensity, float* costs, int nscan, int S, int det,
                                int model, int apply_gradient, float unmeasured_scaling,
                                long num_measured, hipStream_t stream, const int* skip);

extern "C" int tike_ptycho_fwd_intensity(const void* psi, const float* scan, const void* probe,
                                         int probe_per_scan, const void* unique_probe,
                                         const float* eigen_weights, int num_eigen,
                                         int eigen_modes, void* farplane, float* intensity,
                                         void* patches, int nscan, int S, int pw, int det, int H,
                                         int W, float scale, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 1 && det >= pw && H >= 1 && W >= 1);
  TK_CHECK_ARG(!(eigen_weights && probe_per_scan));
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(psi && scan && probe && farplane);
  TK_CHECK_ARG(!(eigen_weights && eigen_modes > 0 && !unique_probe));
  const TkProbe P = tk_make_probe(probe, probe_per_scan, nullptr, eigen_weights, num_eigen,
                                  eigen_modes, S, pw, unique_probe);
  switch (det) {
    case 128:
      // probe window = detector: the whole-tile-in-LDS kernel (no intermediate
      // in memory); a few modes only -- it re-reads nothing, but keeps the
      // patch in registers across the modes at the 128-register cap
      if (pw == 128 && intensity != nullptr && S <= TK_LDS128_MAX_MODES)
        return launch_fwd128_lds((const cf*)psi, scan, P, (cf*)farplane, intensity, nscan, S, H,
                                 W, scale, stream, (cf*)patches);
      return launch_fwd_pos<128>((const cf*)psi, scan, P, (cf*)farplane, intensity, nscan, S, pw,
                                 H, W, scale, stream, nullptr, (cf*)patches);
    case 256:
      return launch_fwd_pos<256>((const cf*)psi, scan, P, (cf*)farplane, intensity, nscan, S, pw,
                                 H, W, scale, stream, nullptr, (cf*)patches);
    case 512:
      return launch_fwd_pos<512>((const cf*)psi, scan, P, (cf*)farplane, intensity, nscan, S, pw,
                                 H, W, scale, stream, nullptr, (cf*)patches);
    default:
      return TK_ERR_UNSUPPORTED;
  }
}

// Forward model for the intensity only (det = 256): `scratch` (nscan,S,det,det)
// receives the column-pass input of every tile instead of the far-plane waves.
extern "C" int tike_ptycho_fwd_intensity_only(const void* psi, const float* scan,
                                              const void* probe, int probe_per_scan,
                                              const void* unique_probe,
                                              const float* eigen_weights, int num_eigen,
                                              int eigen_modes, void* scratch, float* intensity,
                                              int nscan, int S, int pw, int det, int H, int W,
                                              float scale, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 1 && det >= pw && H >= 1 && W >= 1);
  TK_CHECK_ARG(!(eigen_weights && probe_per_scan));
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(psi && scan && probe && scratch && intensity);
  TK_CHECK_ARG(!(eigen_weights && eigen_modes > 0 && !unique_probe));
  if (det != 256) return TK_ERR_UNSUPPORTED;
  const TkProbe P = tk_make_probe(probe, probe_per_scan, nullptr, eigen_weights, num_eigen,
                                  eigen_modes, S, pw, unique_probe);
  return launch_fwd_pos<256, false>((const cf*)psi, scan, P, (cf*)scratch, intensity, nscan, S,
                                    pw, H, W, scale, stream);
}

// tike_ptycho_fwd_intensity_only + tike_gradient_scale in one launch: the
// gradient factor and the per-pattern cost are formed from the intensity while
// it is still in registers (intensity itself is stored only if asked for).
extern "C" int tike_ptycho_fwd_gradient_scale(
    const void* psi, const float* scan, const void* probe, int probe_per_scan,
    const void* unique_probe, const float* eigen_weights, int num_eigen, int eigen_modes,
    void* scratch, float* intensity, void* patches, const float* data,
    const unsigned char* measured, float* gscale, float* costs, int nscan, int S, int pw, int det,
    int H, int W, float scale, int model, float unmeasured_scaling, long num_measured,
    void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 1 && det >= pw && H >= 1 && W >= 1);
  TK_CHECK_ARG(!(eigen_weights && probe_per_scan) && (model == 0 || model == 1) &&
               num_measured > 0);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(psi && scan && probe && scratch && data && gscale);
  TK_CHECK_ARG(!(eigen_weights && eigen_modes > 0 && !unique_probe));
  if (det != 256) return TK_ERR_UNSUPPORTED;
  const TkProbe P = tk_make_probe(probe, probe_per_scan, nullptr, eigen_weights, num_eigen,
                                  eigen_modes, S, pw, unique_probe);
  TkGradScale g;
  g.data = data;
  g.mask = measured;
  g.gscale = gscale;
  g.costs = costs;
  g.model = model;
  g.unmeasured_scaling = unmeasured_scaling;
  g.inv_nmeasured = 1.0f / (float)num_measured;
  return launch_fwd_pos<256, false>((const cf*)psi, scan, P, (cf*)scratch, intensity, nscan, S,
                                    pw, H, W, scale, stream, &g, (cf*)patches);
}

// ---- the 256^2 forward split in two launches (both far-plane free) ----------
// Pass 1 alone: bilinear gather * probe -> row transforms -> radix-16 column
// stage; `scratch` receives the column-pass input of every tile and `patches`
// the object patches.  A work item is (position, 16-row group) and covers all S
// modes (the patch of the group is gathered once and shared by the modes).
// 167 VGPRs at 256^2: three waves per SIMD hide the probe loads of a mode behind
// the other waves' butterflies and stores (an earlier version requested the
// next mode's probe values ahead of the stores instead, at 255 VGPRs and two
// waves: 13 % slower).
// FULL: probe window = detector (pw == N, no padding): every probe / patch
// access of a thread is `uniform base + one 32-bit lane offset + 8 T i` bytes.
// KEEP: plain stores of the hand-off instead of non-temporal ones -- they stay
// in the Infinity Cache for a consumer that follows within ~256 MiB (the
// forward operator's sub-batched column pass); the solver's minibatch-sized
// hand-off does not fit and keeps the non-temporal stores.
template <int N, bool FULL, bool KEEP = false>
__global__ __launch_bounds__(N, N == 512 ? 2 : 3) void fwd_pass1_kernel(
    const cf* __restrict__ psi, const float* __restrict__ scan, const TkProbe probe,
    cf* __restrict__ scratch, cf* __restrict__ patches, int nscan, int S, int pw, int H, int W,
    const cf* __restrict__ twtab, const int* __restrict__ skip) {
  using G2 = Fft2Geom<N>;
  constexpr unsigned EB = G2::T * sizeof(cf);  // bytes between a thread's elements
  // a speculative launch (device-side line search) whose result is not needed
  if (skip != nullptr && *skip != 0) return;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int pad = FULL ? 0 : (N - pw) / 2;
  const long total = (long)H * W;
  const long PP = (long)pw * pw;
  const int t = threadIdx.x;
  auto at = [](const cf* base, unsigned byte_off) -> const cf* {
    return reinterpret_cast<const cf*>(reinterpret_cast<const char*>(base) + byte_off);
  };
  // work item = (position, 16-row group): small launches (the sub-chunked
  // pipeline keeps the hand-off inside the Infinity Cache) still fill the chip
  for (long item = blockIdx.x; item < (long)nscan * G2::RB; item += gridDim.x) {
    const long n = item / G2::RB;
    const int r = (int)(item % G2::RB);
    const TkCorner c = tk_corner(scan, n);
    cf* __restrict__ dst0 = scratch + n * S * (long)N * N;
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    const float* __restrict__ wn =
        probe.weights ? probe.weights + n * (long)(probe.C + 1) * probe.S : nullptr;
    // every tap of every patch pixel inside the image (any position that
    // passes check_allowed_positions) and 32-bit byte offsets suffice
    const bool interior = FULL && c.sy >= 0 && c.sx >= 0 && c.sy + pw < H && c.sx + pw < W &&
                          total < (1L << 28);
    {
      // patch values of row y = r + RB*line in the FFT register layout
      // (element e = j + i*T), gathered once and shared by all S modes
      const int py = r + G2::RB * line - pad;
      const int pyc = py < 0 ? 0 : (py >= pw ? pw - 1 : py);
      // byte offset of this thread's first pixel inside a probe mode / patch
      const unsigned pbo = (unsigned)(pyc * pw + j) * (unsigned)sizeof(cf);
      cf pv[16];
      if (interior) {
        const unsigned g0 = (unsigned)((c.sy + py) * W + c.sx + j) * (unsigned)sizeof(cf);
        const unsigned g1 = g0 + (unsigned)W * (unsigned)sizeof(cf);
        // The two taps of a row are adjacent complex values: one 16-byte load.
        // 8 elements = 16 loads are requested together; the empty asm reads
        // all of them, so none can be sunk next to its use (which would
        // cost one L2 round trip per element).
        typedef float tk_v4f __attribute__((ext_vector_type(4)));
        auto ld4 = [](const cf* base, unsigned byte_off) {
          tk_v4f v;
          __builtin_memcpy(&v, reinterpret_cast<const char*>(base) + byte_off, sizeof(v));
          return v;
        };
#pragma unroll
        for (int h = 0; h < 16; h += 8) {
          tk_v4f u[8], l[8];  // upper row (a, b), lower row (d, e)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            u[i] = ld4(psi, g0 + EB * (h + i));
            l[i] = ld4(psi, g1 + EB * (h + i));
          }
          asm volatile(""
                       : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]),
                         "+v"(u[6]), "+v"(u[7]), "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]),
                         "+v"(l[4]), "+v"(l[5]), "+v"(l[6]), "+v"(l[7]));
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            cf o = mk(u[i].x * c.w00, u[i].y * c.w00);
            o.x += u[i].z * c.w01;
            o.y += u[i].w * c.w01;
            o.x += l[i].x * c.w10;
            o.y += l[i].y * c.w10;
            o.x += l[i].z * c.w11;
            o.y += l[i].w * c.w11;
            pv[h + i] = o;
          }
        }
      } else {
        const int y = c.sy + py;
        const bool row_ok = py >= 0 && py < pw && y >= 0 && y < H;
        const int yc = c.sy + pyc < 0 ? 0 : (c.sy + pyc >= H ? H - 1 : c.sy + pyc);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int px = j + i * G2::T - pad;
          const int x = c.sx + px;
          const bool ok = row_ok && px >= 0 && px < pw && x >= 0 && x < W;
          const int pxc = px < 0 ? 0 : (px >= pw ? pw - 1 : px);
          const int xc = c.sx + pxc < 0 ? 0 : (c.sx + pxc >= W ? W - 1 : c.sx + pxc);
          const cf o = tk_gather(psi, (long)yc * W + xc, W, total, c);
          pv[i] = ok ? o : mk(0.f, 0.f);
          __builtin_amdgcn_sched_barrier(0);  // rare path: one element in flight
        }
      }
      if (patches != nullptr && py >= 0 && py < pw) {
        cf* __restrict__ On = patches + n * PP;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int px = j + i * G2::T - pad;
          if (FULL)
            tk_st_stream(const_cast<cf*>(at(On, pbo + EB * i)), pv[i]);
          else if (px >= 0 && px < pw)
            tk_st_stream(On + (long)py * pw + px, pv[i]);
        }
      }
      // probe of (position, mode): shared probe times its weight, plus the
      // eigen probes of the first Sm modes (probe.py:272-303) -- from the
      // synthesised array when given, else on the fly
      auto load_probe = [&](int s, cf (&pn)[16]) {
        const cf* __restrict__ Pn = probe.probe + n * probe.pos_stride + s * PP;
        float w0 = 1.0f;
        int nE = 0;
        if (wn != nullptr) {
          if (probe.unique != nullptr && s < probe.Sm) {
            Pn = probe.unique + (n * probe.Sm + s) * PP;
          } else {
            w0 = wn[s];
            if (probe.eigen != nullptr && s < probe.Sm) nE = probe.C;
          }
        }
        auto pix = [&](const cf* base, int i) {
          if (FULL) return *at(base, pbo + EB * i);
          const int px = j + i * G2::T - pad;
          const int pxc = px < 0 ? 0 : (px >= pw ? pw - 1 : px);
          return base[pyc * pw + pxc];
        };
#pragma unroll
        for (int i = 0; i < 16; ++i) pn[i] = pix(Pn, i) * w0;
        for (int k = 0; k < nE; ++k) {  // uniform, rare (modes owning eigen probes)
          const cf* __restrict__ E = probe.eigen + ((long)k * probe.Sm + s) * PP;
          const float wk = wn[(k + 1) * probe.S + s];
          cf e[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) e[i] = pix(E, i);
          __builtin_amdgcn_sched_barrier(0);  // all 16 in flight before the first use
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            pn[i].x += wk * e[i].x;
            pn[i].y += wk * e[i].y;
          }
        }
      };
      for (int s = 0; s < S; ++s) {
        // (requesting the next mode's probe values ahead of this mode's stores
        // hid one load latency but cost 32 registers: without it the kernel
        // fits 167 VGPRs = 3 waves/SIMD at 256^2 and runs 13 % faster.  At
        // 512^2, where one 512-thread workgroup owns the CU either way and the
        // registers are there, the same request ahead is 5 % slower too:
        // 2.77 -> 2.91 ms per 1000 positions x 4 modes, round 5)
        cf pn[16];
        load_probe(s, pn);
        cf v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = pv[i] * pn[i];
        cf* lbase = lds + line * G2::LS;
        FftStageWave<N, false, 0>::run(v, lbase, j, tw);
#pragma unroll
        for (int i = 0; i < 16; ++i) lbase[tk_pad16(j + i * G2::T)] = v[i];
        __syncthreads();
#pragma unroll
        for (int y2 = 0; y2 < 16; ++y2) v[y2] = lds[y2 * G2::LS + tk_pad16(t)];
        __syncthreads();
        Dft<16, false>::run(v);
        cf* __restrict__ mid = dst0 + s * (long)N * N + (long)(16 * r) * N + t;
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
          cf o = v[k1];
          if (k1 > 0) o = mul_tw<false>(o, twtab[N + r * k1]);  // uniform -> scalar load
          if (KEEP)
            mid[k1 * N] = o;
          else
            tk_st_stream(mid + k1 * N, o);
        }
      }
    }
  }
}

// ---- pass 1 at 512^2, one WAVE per row (round 6) ----------------
// fwd_pass1_kernel<512> holds 16 elements per thread (32 threads per row, one
// 512-thread workgroup per CU at 204-244 VGPRs, two waves per SIMD) and moves
// its bytes at 4.1 TB/s where the 256^2 form reaches 5.4.  Here a thread holds
// EIGHT elements (e = lane + 64 i): a row is one wave, 512 = 8 x 8 x 8 in three
// radix-8 stages with two exchanges inside the wave through the row's LDS
// slot; 1024 threads per workgroup = the 16 rows of a group, and in the column
// phase TWO threads share a column (the radix-16 over y2 as one
// decimation-in-frequency step and a radix-8 each: even / odd k1).
// <= 128 VGPRs: four waves per SIMD.  LDS: 16 rows x 545 + 1024 twiddles.
struct TkRow8 {
  static constexpr int N = 512, T = 64, LS = N + N / 16 + 1;
  // twiddles of stages 1 and 2: tab[((s - 1) * 8 + r) * 64 + lane]
  static constexpr int TW_ELEMS = 2 * 8 * 64;
  static __device__ __forceinline__ void fill(cf* tab, const cf* __restrict__ g_tw) {
    for (int idx = threadIdx.x; idx < TW_ELEMS; idx += blockDim.x) {
      const int l = idx & 63, r = (idx >> 6) & 7, st = (idx >> 9) + 1;
      const int Ns = st == 1 ? 8 : 64;
      tab[idx] = g_tw[N + (l & (Ns - 1)) * r * (N / (Ns * 8))];
    }
  }
  // v[i] = element lane + 64 i of the row, in and out (natural order)
  template <bool INV>
  static __device__ __forceinline__ void run(cf (&v)[8], cf* __restrict__ lbase, int l,
                                             const cf* __restrict__ tab) {
#pragma unroll
    for (int st = 0; st < 3; ++st) {
      const int Ns = st == 0 ? 1 : (st == 1 ? 8 : 64);
      if (st > 0) {
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = mul_tw<INV>(v[r], tab[((st - 1) * 8 + r) * 64 + l]);
      }
      Dft<8, INV>::run(v);
      if (st < 2) {
        const int k = l & (Ns - 1);
        const int j0 = (l - k) * 8 + k;
#pragma unroll
        for (int r = 0; r < 8; ++r) lbase[tk_pad16(j0 + r * Ns)] = v[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = lbase[tk_pad16(l + 64 * i)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
};

template <bool FULL, bool KEEP>
__global__ __launch_bounds__(1024, 1) void fwd_pass1_512w_kernel(
    const cf* __restrict__ psi, const float* __restrict__ scan, const TkProbe probe,
    cf* __restrict__ scratch, cf* __restrict__ patches, int nscan, int S, int pw, int H, int W,
    const cf* __restrict__ twtab, const int* __restrict__ skip) {
  constexpr int N = 512, RB = 32, LS = TkRow8::LS;
  constexpr unsigned EB = 64 * sizeof(cf);  // bytes between a thread's elements
  if (skip != nullptr && *skip != 0) return;
  // two sets of 16 rows: mode s + 1 writes its rows while the columns of mode s
  // are still being read -- ONE barrier per mode (the only workgroup of its CU
  // has nobody to hide a second one behind)
  __shared__ cf lds[2 * 16 * LS + TkRow8::TW_ELEMS];
  cf* twl = lds + 2 * 16 * LS;
  TkRow8::fill(twl, twtab);
  __syncthreads();
  const int pad = FULL ? 0 : (N - pw) / 2;
  const long total = (long)H * W;
  const long PP = (long)pw * pw;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int l = threadIdx.x & 63;
  asm volatile("" : "+v"(l));
  const int col = threadIdx.x & 511;
  const int q = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 9));  // even / odd k1
  auto at = [](const cf* base, unsigned byte_off) -> const cf* {
    return reinterpret_cast<const cf*>(reinterpret_cast<const char*>(base) + byte_off);
  };
  for (long item = blockIdx.x; item < (long)nscan * RB; item += gridDim.x) {
    const long n = item / RB;
    const int r = (int)(item % RB);
    const TkCorner c = tk_corner(scan, n);
    cf* __restrict__ dst0 = scratch + n * S * (long)N * N;
    const float* __restrict__ wn =
        probe.weights ? probe.weights + n * (long)(probe.C + 1) * probe.S : nullptr;
    const bool interior = FULL && c.sy >= 0 && c.sx >= 0 && c.sy + pw < H && c.sx + pw < W &&
                          total < (1L << 28);
    // this wave's row: y = r + 32 w, elements e = l + 64 i
    const int py = r + RB * w - pad;
    const int pyc = py < 0 ? 0 : (py >= pw ? pw - 1 : py);
    const unsigned pbo = (unsigned)(pyc * pw + l) * (unsigned)sizeof(cf);
    cf pv[8];
    if (interior) {
      const unsigned g0 = (unsigned)((c.sy + py) * W + c.sx + l) * (unsigned)sizeof(cf);
      const unsigned g1 = g0 + (unsigned)W * (unsigned)sizeof(cf);
      typedef float tk_v4f __attribute__((ext_vector_type(4)));
      auto ld4 = [](const cf* base, unsigned byte_off) {
        tk_v4f v;
        __builtin_memcpy(&v, reinterpret_cast<const char*>(base) + byte_off, sizeof(v));
        return v;
      };
      tk_v4f u[8], lo[8];  // upper row (a, b), lower row (d, e)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        u[i] = ld4(psi, g0 + EB * i);
        lo[i] = ld4(psi, g1 + EB * i);
      }
      asm volatile(""
                   : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]),
                     "+v"(u[6]), "+v"(u[7]), "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]),
                     "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        cf o = mk(u[i].x * c.w00, u[i].y * c.w00);
        o.x += u[i].z * c.w01;
        o.y += u[i].w * c.w01;
        o.x += lo[i].x * c.w10;
        o.y += lo[i].y * c.w10;
        o.x += lo[i].z * c.w11;
        o.y += lo[i].w * c.w11;
        pv[i] = o;
      }
    } else {
      const int y = c.sy + py;
      const bool row_ok = py >= 0 && py < pw && y >= 0 && y < H;
      const int yc = c.sy + pyc < 0 ? 0 : (c.sy + pyc >= H ? H - 1 : c.sy + pyc);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int px = l + 64 * i - pad;
        const int x = c.sx + px;
        const bool ok = row_ok && px >= 0 && px < pw && x >= 0 && x < W;
        const int pxc = px < 0 ? 0 : (px >= pw ? pw - 1 : px);
        const int xc = c.sx + pxc < 0 ? 0 : (c.sx + pxc >= W ? W - 1 : c.sx + pxc);
        const cf o = tk_gather(psi, (long)yc * W + xc, W, total, c);
        pv[i] = ok ? o : mk(0.f, 0.f);
        __builtin_amdgcn_sched_barrier(0);  // rare path: one element in flight
      }
    }
    if (patches != nullptr && py >= 0 && py < pw) {
      cf* __restrict__ On = patches + n * PP;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int px = l + 64 * i - pad;
        if (FULL)
          tk_st_stream(const_cast<cf*>(at(On, pbo + EB * i)), pv[i]);
        else if (px >= 0 && px < pw)
          tk_st_stream(On + (long)py * pw + px, pv[i]);
      }
    }
    auto pix = [&](const cf* base, int i) {
      if (FULL) return *at(base, pbo + EB * i);
      const int px = l + 64 * i - pad;
      const int pxc = px < 0 ? 0 : (px >= pw ? pw - 1 : px);
      return base[pyc * pw + pxc];
    };
    // probe of (position, mode) (probe.py:272-303, as fwd_pass1_kernel).
    // (Requested one mode ahead, before the transforms of the mode in hand: 12
    // bytes of scratch per lane and 2.76 instead of 2.62 ms per 1000 x 4.)
    auto load_probe = [&](int s, cf (&pn)[8]) {
      const cf* __restrict__ Pn = probe.probe + n * probe.pos_stride + s * PP;
      float w0 = 1.0f;
      int nE = 0;
      if (wn != nullptr) {
        if (probe.unique != nullptr && s < probe.Sm) {
          Pn = probe.unique + (n * probe.Sm + s) * PP;
        } else {
          w0 = wn[s];
          if (probe.eigen != nullptr && s < probe.Sm) nE = probe.C;
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) pn[i] = pix(Pn, i) * w0;
      for (int k = 0; k < nE; ++k) {  // uniform, rare (modes owning eigen probes)
        const cf* __restrict__ E = probe.eigen + ((long)k * probe.Sm + s) * PP;
        const float wk = wn[(k + 1) * probe.S + s];
        cf e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = pix(E, i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          pn[i].x += wk * e[i].x;
          pn[i].y += wk * e[i].y;
        }
      }
    };
    for (int s = 0; s < S; ++s) {
      cf v[8];
      load_probe(s, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = pv[i] * v[i];
      cf* const set = lds + (s & 1) * 16 * LS;
      cf* lbase = set + w * LS;
      TkRow8::run<false>(v, lbase, l, twl);
#pragma unroll
      for (int i = 0; i < 8; ++i) lbase[tk_pad16(l + 64 * i)] = v[i];
      __syncthreads();
      // column `col` of the 16 rows: the radix-16 over y2 split between TWO
      // threads by one decimation-in-frequency step -- q = 0 the even k1 (radix-8
      // of x[n] + x[n + 8]), q = 1 the odd ones (radix-8 of (x[n] - x[n + 8])
      // w_16^n): all 1024 threads work, eight values each
      cf t[8];
      {
        constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f;
        constexpr float H2 = 0.70710678118654752f;
        const cf w16[8] = {mk(1.f, 0.f),  mk(C1, -S1),  mk(H2, -H2),  mk(S1, -C1),
                           mk(0.f, -1.f), mk(-S1, -C1), mk(-H2, -H2), mk(-C1, -S1)};
#pragma unroll
        for (int y2 = 0; y2 < 8; ++y2) {
          const cf a = set[y2 * LS + tk_pad16(col)];
          const cf b = set[(y2 + 8) * LS + tk_pad16(col)];
          t[y2] = q == 0 ? a + b : (a - b) * w16[y2];
        }
      }
      Dft<8, false>::run(t);
      cf* __restrict__ mid = dst0 + s * (long)N * N + (long)(16 * r) * N + col;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int k1 = 2 * m + q;
        cf o = t[m];
        o = mul_tw<false>(o, twtab[N + r * k1]);  // uniform -> scalar load (k1 = 0: 1)
        if (KEEP)
          mid[k1 * N] = o;
        else
          tk_st_stream(mid + k1 * N, o);
      }
    }
    // (the next item starts at set 0 again: after an odd number of modes that
    // is the set just read)
    if (S & 1) __syncthreads();
  }
}

// unique_probe: the varying probe of the first eigen_modes modes from
// tike_varying_probe, or NULL with eigen_probe given: formed on the fly.
static int tk_fwd_pass1(const void* psi, const float* scan, const void* probe,
                        int probe_per_scan, const void* unique_probe, const void* eigen_probe,
                        const float* eigen_weights, int num_eigen, int eigen_modes, void* scratch,
                        void* patches, int nscan, int S, int pw, int det, int H, int W,
                        hipStream_t stream, const int* skip, bool keep = false) {
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 1 && det >= pw && H >= 1 && W >= 1);
  TK_CHECK_ARG(!(eigen_weights && probe_per_scan));
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(psi && scan && probe && scratch);
  TK_CHECK_ARG(!(eigen_weights && eigen_modes > 0 && !unique_probe && !eigen_probe));
  if (det != 128 && det != 256 && det != 512) return TK_ERR_UNSUPPORTED;
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
  const TkProbe P = tk_make_probe(probe, probe_per_scan, unique_probe ? nullptr : eigen_probe,
                                  eigen_weights, num_eigen, eigen_modes, S, pw, unique_probe);
#define TK_F1K(N, FULL, KEEP)                                                                   \
  hipLaunchKernelGGL((fwd_pass1_kernel<N, FULL, KEEP>),                                        \
                     dim3(tk_grid((long)nscan * (N / 16), N == 512 ? 1 : (N == 256 ? 8 : 16))), \
                     dim3(N), 0, stream, (const cf*)psi, scan, P, (cf*)scratch, (cf*)patches,   \
                     nscan, S, pw, H, W, tw, skip)
#define TK_F1(N, FULL)      \
  do {                      \
    if (keep)               \
      TK_F1K(N, FULL, true);  \
    else                    \
      TK_F1K(N, FULL, false); \
  } while (0)
  if (det == 128 && pw == det)  // (a multislice object: tike_slice_step's partner)
    TK_F1(128, true);
  else if (det == 128)
    TK_F1(128, false);
  else if (det == 256 && pw == det)
    TK_F1(256, true);
  else if (det == 256)
    TK_F1(256, false);
  else {
    // 512^2: one wave per row, 1024 threads (fwd_pass1_kernel<512>: 2.77 ms per
    // 1000 positions x 4 modes; this one 2.61)
#define TK_F1W(FULL, KEEP)                                                                      \
  hipLaunchKernelGGL((fwd_pass1_512w_kernel<FULL, KEEP>), dim3(tk_grid((long)nscan * 32, 1)),   \
                     dim3(1024), 0, stream, (const cf*)psi, scan, P, (cf*)scratch,              \
                     (cf*)patches, nscan, S, pw, H, W, tw, skip)
    if (pw == det && keep)
      TK_F1W(true, true);
    else if (pw == det)
      TK_F1W(true, false);
    else if (keep)
      TK_F1W(false, true);
    else
      TK_F1W(false, false);
#undef TK_F1W
  }
#undef TK_F1
#undef TK_F1K
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_fwd_pass1(const void* psi, const float* scan, const void* probe,
                              int probe_per_scan, const void* unique_probe,
                              const void* eigen_probe, const float* eigen_weights,
                              int num_eigen, int eigen_modes, void* scratch, void* patches,
                              int nscan, int S, int pw, int det, int H, int W, void* stream_) {
  TK_ENTER();
  return tk_fwd_pass1(psi, scan, probe, probe_per_scan, unique_probe, eigen_probe, eigen_weights,
                      num_eigen, eigen_modes, scratch, patches, nscan, S, pw, det, H, W,
                      (hipStream_t)stream_, nullptr);
}

// Counts and mask bits of the RB pixels (k1 + 16 k2, t) of position n of an
// N x N pattern, requested TOGETHER and unconditionally (a branch per pixel
// around its load makes RB serial memory round trips of them); unmeasured
// pixels may hold NaN: they are selected away by the mask bit, never multiplied.
template <int N, int RB, class DT>
__device__ __forceinline__ void tk_request_data(const DT* __restrict__ data,
                                                const unsigned char* __restrict__ mask, long n,
                                                int k1, int t, DT (&raw)[RB], unsigned& bits) {
  static_assert(RB <= 32, "one mask bit per pixel");
  const DT* __restrict__ d = data + n * (long)N * N + k1 * N;  // uniform
  const unsigned lo = (unsigned)t * (unsigned)sizeof(DT);
#pragma unroll
  for (int k2 = 0; k2 < RB; ++k2) raw[k2] = *tk_at_pinned(d + (16 * k2) * N, lo);
  bits = 0xffffffffu;
  if (mask) {  // uniform
    unsigned char mb[RB];
#pragma unroll
    for (int k2 = 0; k2 < RB; ++k2) mb[k2] = mask[(k1 + 16 * k2) * N + t];
    bits = 0;
#pragma unroll
    for (int k2 = 0; k2 < RB; ++k2) bits |= (mb[k2] ? 1u : 0u) << k2;
  }
}
template <class DT>
__device__ __forceinline__ void tk_request_data16(const DT* __restrict__ data,
                                                  const unsigned char* __restrict__ mask, long n,
                                                  int k1, int t, DT (&raw)[16], unsigned& bits) {
  tk_request_data<256, 16>(data, mask, n, k1, t, raw, bits);
}

// I[k2] (intensity) -> g * fwd_scale, returns this thread's cost terms.
template <int MODEL, int RB, class DT>
__device__ __forceinline__ float tk_gradient_factor(float (&I)[RB], const DT (&raw)[RB],
                                                    unsigned bits, float unmeasured_scaling,
                                                    float fwd_scale) {
  float cost = 0.f;
#pragma unroll
  for (int k2 = 0; k2 < RB; ++k2) {
    const bool meas = (bits >> k2) & 1u;
    const float dv = (float)raw[k2];
    float term, g;
    if (MODEL == 0) {
      const float sI = sqrtf(I[k2]), sd = sqrtf(dv);
      const float diff = sI - sd;
      term = diff * diff;
      g = -(1.0f - sd / (sI + 1e-9f));
    } else {
      term = I[k2] - dv * logf(I[k2] + 1e-9f);
      g = -(1.0f - dv / (I[k2] + 1e-9f));
    }
    cost += meas ? term : 0.f;
    I[k2] = (meas ? g : unmeasured_scaling - 1.0f) * fwd_scale;
  }
  return cost;
}
template <int MODEL, class DT>
__device__ __forceinline__ float tk_gradient_factor16(float (&I)[16], const DT (&raw)[16],
                                                      unsigned bits, float unmeasured_scaling,
                                                      float fwd_scale) {
  return tk_gradient_factor<MODEL, 16>(I, raw, bits, unmeasured_scaling, fwd_scale);
}

// The column pass as a pure read stream: one workgroup per (position, k1)
// forms F[k1 + 16 k2] of every mode in registers (radix-16 over the rows
// 16 r + k1 of the hand-off), accumulates I = sum_s |F_s|^2 and emits the
// gradient factor and the cost share of those 16 rows (objective.py:11-124,
// lstsq.py:444-502).  costs must be zero on entry (accumulated by atomics).
// DT: float, or unsigned short for detector counts kept as they arrived
// (16-bit data stays 16-bit in HBM, reference ptycho.py:383-390).
template <int N, int MODEL, class DT>
__global__ __launch_bounds__(256, N == 256 ? 4 : 2) void fwd_gradient_scale_kernel(
    const cf* __restrict__ colin, const DT* __restrict__ data,
    const unsigned char* __restrict__ mask, float* __restrict__ gscale,
    float* __restrict__ intensity, const TkCostSink costs, cf* __restrict__ farplane,
    long nitem, int S, float scale, float unmeasured_scaling, float inv_nmeasured,
    const int* __restrict__ skip) {
  constexpr int RB = N / 16;    // radix of the column pass
  constexpr int NH = N / 256;   // 256-column blocks per row
  __shared__ float red[4];
  if (skip != nullptr && *skip != 0) return;  // speculative launch, not needed
  const float s2 = scale * scale;
  for (long v = blockIdx.x; v < nitem; v += gridDim.x) {
    // item = (position, k1, column block)
    const int hb = (int)(v % NH);
    const int k1 = (int)((v / NH) & 15);
    // positions in DESCENDING order: the last hand-off tiles forward pass 1
    // wrote (ascending) are still in the 256 MB Infinity Cache when this
    // kernel starts, and the ones read last here are the first the next
    // kernel (ascending again) asks for
    const long n = nitem / (16 * NH) - 1 - v / (16 * NH);
    const int t = hb * 256 + threadIdx.x;
    float I[RB];
#pragma unroll
    for (int k2 = 0; k2 < RB; ++k2) I[k2] = 0.f;
    // 256^2: software pipelined over the modes -- the rows of mode s + 1 are
    // requested before the butterflies of mode s (124 VGPRs, still 4 waves per
    // SIMD; 1.00 -> 0.89 ms per 1000 positions).  At 512^2 (radix 32) the second
    // set of rows costs a wave per SIMD and loses (0.85 -> 0.92 ms).
    constexpr bool PIPE = N == 256;
    cf un[PIPE ? RB : 1];
    if (PIPE) {
      const cf* __restrict__ src0 = colin + (n * S) * (long)N * N + k1 * N + t;
#pragma unroll
      for (int r = 0; r < RB; ++r) un[PIPE ? r : 0] = tk_ld_stream(src0 + (long)(16 * r) * N);
    }
    for (int s = 0; s < S; ++s) {
      cf u[RB];
      if (PIPE) {
#pragma unroll
        for (int r = 0; r < RB; ++r) u[r] = un[PIPE ? r : 0];
        if (s + 1 < S) {
          const cf* __restrict__ src = colin + (n * S + s + 1) * (long)N * N + k1 * N + t;
#pragma unroll
          for (int r = 0; r < RB; ++r)
            un[PIPE ? r : 0] = tk_ld_stream(src + (long)(16 * r) * N);
        }
      } else {
        const cf* __restrict__ src = colin + (n * S + s) * (long)N * N + k1 * N + t;
#pragma unroll
        for (int r = 0; r < RB; ++r) u[r] = tk_ld_stream(src + (long)(16 * r) * N);
      }
      Dft<RB, false>::run(u);
#pragma unroll
      for (int k2 = 0; k2 < RB; ++k2) I[k2] += norm2(u[k2]) * s2;
      if (farplane != nullptr) {
        // the far-plane wave itself, for the pipelines that keep it
        cf* __restrict__ dst = farplane + (n * S + s) * (long)N * N + k1 * N + t;
#pragma unroll
        for (int k2 = 0; k2 < RB; ++k2) tk_st_stream(dst + (long)(16 * k2) * N, u[k2] * scale);
      }
    }
    // counts and mask requested together, the factor selected (not branched)
    DT raw[RB];
    unsigned bits;
    tk_request_data<N, RB>(data, mask, n, k1, t, raw, bits);
    if (intensity) {  // uniform
#pragma unroll
      for (int k2 = 0; k2 < RB; ++k2)
        tk_st_stream(intensity + n * (long)N * N + (long)(k1 + 16 * k2) * N + t, I[k2]);
    }
    float cost = tk_gradient_factor<MODEL, RB>(I, raw, bits, unmeasured_scaling, 1.0f);
    if (gscale) {  // uniform
#pragma unroll
      for (int k2 = 0; k2 < RB; ++k2)
        gscale[n * (long)N * N + (long)(k1 + 16 * k2) * N + t] = I[k2];
    }
    if (costs.costs) {
      cost = tk_block_sum256(cost, red);
      if (threadIdx.x == 0) tk_cost_add(costs, n, k1 * NH + hb, cost * inv_nmeasured);
    }
  }
}

// scratch: from tike_fwd_pass1 (UNSCALED column-pass input; `scale` is the
// forward FFT normalisation applied here).  intensity / costs may be NULL, and
// so may gscale when only the costs are wanted (a line-search probe).
static int tk_fwd_gradient_scale(const void* scratch, const void* data, int data_u16,
                                 const unsigned char* measured, float* gscale, float* intensity,
                                 float* costs, void* farplane, int nscan, int S, int det,
                                 float scale, int model, float unmeasured_scaling,
                                 long num_measured, hipStream_t stream, const int* skip) {
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && det >= 1 && (model == 0 || model == 1) &&
               num_measured > 0);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(scratch && data && (gscale || costs) && farplane != scratch);
  if (det != 256 && det != 512) return TK_ERR_UNSUPPORTED;
  // (a speculative launch may return at once: its cost slots would be stale,
  // so the line searches keep the zeroed atomics)
  TkCostSink sink{costs, nullptr, 0};
  if (skip == nullptr) {
    int rc = tk_cost_sink(costs, nscan, 16 * (det / 256), stream, &sink);
    if (rc) return rc;
  } else if (costs) {
    hipError_t e = hipMemsetAsync(costs, 0, sizeof(float) * (size_t)nscan, stream);
    if (e != hipSuccess) return (int)e;
  }
  const long nitem = (long)nscan * 16 * (det / 256);
  const float inv = 1.0f / (float)num_measured;
  const dim3 grid(tk_grid(nitem, 32)), block(256);
#define TK_FGS(N, M, DT)                                                                     \
  hipLaunchKernelGGL((fwd_gradient_scale_kernel<N, M, DT>), grid, block, 0, stream,             \
                     (const cf*)scratch, (const DT*)data, measured, gscale, intensity, sink,    \
                     (cf*)farplane, nitem, S, scale, unmeasured_scaling, inv, skip)
#define TK_FGS_N(N)                     \
  do {                                  \
    if (model == 0 && data_u16)         \
      TK_FGS(N, 0, unsigned short);     \
    else if (model == 0)                \
      TK_FGS(N, 0, float);              \
    else if (data_u16)                  \
      TK_FGS(N, 1, unsigned short);     \
    else                                \
      TK_FGS(N, 1, float);              \
  } while (0)
  if (det == 256)
    TK_FGS_N(256);
  else
    TK_FGS_N(512);
#undef TK_FGS_N
#undef TK_FGS
  TK_LAUNCH_CHECK();
  return tk_cost_finish(sink, nscan, stream);
}

extern "C" int tike_fwd_gradient_scale(const void* scratch, const void* data, int data_u16,
                                       const unsigned char* measured, float* gscale,
                                       float* intensity, float* costs, void* farplane, int nscan,
                                       int S, int det, float scale, int model,
                                       float unmeasured_scaling, long num_measured,
                                       void* stream_) {
  TK_ENTER();
  return tk_fwd_gradient_scale(scratch, data, data_u16, measured, gscale, intensity, costs,
                               farplane, nscan, S, det, scale, model, unmeasured_scaling,
                               num_measured, (hipStream_t)stream_, nullptr);
}

// ------------------------------------------- line search decided on the device
// Backtracking line search of the conjugate-gradient solver (reference
// opt.py:216-278 line_search, as composed by solvers/cgrad.py): try
// x + step d, x + step/2 d, ... until the gaussian cost of the minibatch is no
// larger than at x.  Every trial is a cost-only forward pass; its launches are
// enqueued for `nslots` step lengths AHEAD of the decisions, and a trial whose
// predecessor was accepted returns at once (the `skip` word the kernels read):
// no host round trip per trial.
// state (device, double[5]): { fx = mean cost at x, step, done, trials, failures }.
//   in : fx, step (first step length to try)
//   out: accepted -> fx = mean cost there, step = that step length, done = 1
//        otherwise  step = the next step length to try (step / 2^nslots), done = 0,
//        failures += 1 (a caller that chains searches reads it once at the end)
// xs receives x + step d of the LAST trial made (accepted: the new iterate).
__global__ __launch_bounds__(256) void ls_trial_kernel(const cf* __restrict__ x,
                                                       const cf* __restrict__ d,
                                                       cf* __restrict__ xs, long n,
                                                       const double* __restrict__ state,
                                                       float shrink,
                                                       const int* __restrict__ skip) {
  if (*skip != 0) return;
  const float a = (float)state[1] * shrink;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const cf v = d[i];
    xs[i] = mk(x[i].x + a * v.x, x[i].y + a * v.y);
  }
}

// One workgroup: mean cost of the trial; accept if it is no larger than fx.
__global__ __launch_bounds__(256) void ls_decide_kernel(const float* __restrict__ costs, int n,
                                                        double inv_count, float shrink,
                                                        int last, double* __restrict__ state,
                                                        int* __restrict__ skip) {
  if (*skip != 0) return;
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += (double)costs[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double f = red[0] * inv_count;
    state[3] += 1.0;
    if (f <= state[0]) {
      state[0] = f;
      state[1] = (double)((float)state[1] * shrink);
      state[2] = 1.0;
      *skip = 1;
    } else if (last) {
      state[1] = (double)((float)state[1] * shrink * 0.5f);
      state[4] += 1.0;
    }
  }
}

// ------------------------------------------- conjugate direction on the device
// Dai-Yuan direction of the conjugate-gradient solver (reference opt.py:281-301
// direction_dy as solvers/cgrad.py composes it) in two kernels instead of a
// dozen element-wise launches:
//   g1 = -(accumulated update)            (object: planar (2, n) float32;
//                                          probe: interleaved complex (n))
//   first:  d = -g1
//   else:   d = -g1 + d |g1|^2 / (sum conj(d) (g1 - g0) + 1e-32)
//   g0 <- g1;  first: state[0] = sum(costs) / count   (the cost at x)
// sums[0..3] (double, zeroed here): |g1|^2, Re / Im of the denominator, sum(costs)
__global__ __launch_bounds__(256) void cg_sums_kernel(const float* __restrict__ planar,
                                                      const cf* __restrict__ inter,
                                                      const cf* __restrict__ g0,
                                                      const cf* __restrict__ d, long n, int first,
                                                      const float* __restrict__ costs, int ncost,
                                                      double* __restrict__ sums) {
  __shared__ float red[4];
  __shared__ double redd[256];
  float nn = 0.f, dr = 0.f, di = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const cf a = planar ? mk(planar[i], planar[n + i]) : inter[i];
    const cf g1 = mk(-a.x, -a.y);
    nn += norm2(g1);
    if (!first) {
      const cf y = mk(g1.x - g0[i].x, g1.y - g0[i].y);
      const cf t = conjf(d[i]) * y;
      dr += t.x;
      di += t.y;
    }
  }
  nn = tk_block_sum256(nn, red);
  dr = tk_block_sum256(dr, red);
  di = tk_block_sum256(di, red);
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(&sums[0], (double)nn);
    if (!first) {
      unsafeAtomicAdd(&sums[1], (double)dr);
      unsafeAtomicAdd(&sums[2], (double)di);
    }
  }
  if (first && costs != nullptr) {  // uniform: the mean cost, summed in double
    double cs = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < ncost; i += gridDim.x * 256L)
      cs += (double)costs[i];
    redd[threadIdx.x] = cs;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) redd[threadIdx.x] += redd[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0 && redd[0] != 0.0) unsafeAtomicAdd(&sums[3], redd[0]);
  }
}

__global__ __launch_bounds__(256) void cg_direction_kernel(const float* __restrict__ planar,
                                                           const cf* __restrict__ inter,
                                                           cf* __restrict__ g0, cf* __restrict__ d,
                                                           long n, int first, int have_costs,
                                                           double inv_count,
                                                           const double* __restrict__ sums,
                                                           double* __restrict__ state) {
  cf beta = mk(0.f, 0.f);
  if (!first) {
    // |g1|^2 / (den + 1e-32), complex
    const float nr = (float)sums[0];
    const float er = (float)sums[1] + 1e-32f, ei = (float)sums[2];
    const float m = er * er + ei * ei;
    beta = mk(nr * er / m, -nr * ei / m);
  }
  if (first && have_costs && blockIdx.x == 0 && threadIdx.x == 0) state[0] = sums[3] * inv_count;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const cf a = planar ? mk(planar[i], planar[n + i]) : inter[i];
    const cf g1 = mk(-a.x, -a.y);
    cf nd = mk(-g1.x, -g1.y);
    if (!first) {
      const cf t = d[i] * beta;
      nd = mk(t.x - g1.x, t.y - g1.y);
    }
    d[i] = nd;
    g0[i] = g1;
  }
}

extern "C" int tike_cgrad_direction(const float* update_planar, const void* update_complex,
                                    void* gradient, void* direction, long n, int first,
                                    const float* costs, int ncost, double count, double* state,
                                    double* sums, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(n >= 1 && gradient && direction && sums);
  TK_CHECK_ARG((update_planar != nullptr) != (update_complex != nullptr));
  TK_CHECK_ARG(!(first && costs != nullptr) || (ncost >= 1 && count > 0 && state != nullptr));
  hipError_t e = hipMemsetAsync(sums, 0, 4 * sizeof(double), stream);
  if (e != hipSuccess) return (int)e;
  const dim3 grid(tk_grid((n + 255) / 256, 4)), block(256);
  // deterministic mode: ONE summing workgroup (its tree is fixed; the double
  // atomics of several workgroups arrive in any order)
  hipLaunchKernelGGL(cg_sums_kernel, tk_deterministic() ? dim3(1) : grid, block, 0, stream,
                     update_planar,
                     (const cf*)update_complex, (const cf*)gradient, (const cf*)direction, n,
                     first, first ? costs : nullptr, ncost, sums);
  hipLaunchKernelGGL(cg_direction_kernel, grid, block, 0, stream, update_planar,
                     (const cf*)update_complex, (cf*)gradient, (cf*)direction, n, first,
                     (int)(first && costs != nullptr), count > 0 ? 1.0 / count : 0.0, sums, state);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_cgrad_line_search(int variable, const void* x, const void* d, void* xs,
                                      const void* other, const float* scan, const void* data,
                                      int data_u16, void* scratch, float* costs, int nscan,
                                      int chunk, int S, int det, int H, int W, float fwd_scale,
                                      double count, double* state, int* skip, int nslots,
                                      void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 1 && chunk >= 1 && S >= 1 && H >= 1 && W >= 1 && nslots >= 1 &&
               nslots <= 30 && count > 0 && (variable == 0 || variable == 1));
  TK_CHECK_ARG(x && d && xs && other && scan && data && scratch && costs && state && skip);
  if (det != 128 && det != 256 && det != 512) return TK_ERR_UNSUPPORTED;
  if (det == 128 && data_u16) return TK_ERR_UNSUPPORTED;  // the 128^2 cost kernel reads float32
  const long n = variable == 0 ? (long)H * W : (long)S * det * det;
  hipError_t e = hipMemsetAsync(skip, 0, sizeof(int), stream);
  if (e == hipSuccess) e = hipMemsetAsync(state + 2, 0, sizeof(double), stream);  // done = 0
  if (e != hipSuccess) return (int)e;
  const size_t dsz = data_u16 ? 2 : 4;
  float shrink = 1.0f;
  for (int k = 0; k < nslots; ++k, shrink *= 0.5f) {
    hipLaunchKernelGGL(ls_trial_kernel, dim3(tk_grid((n + 255) / 256, 8)), dim3(256), 0, stream,
                       (const cf*)x, (const cf*)d, (cf*)xs, n, state, shrink, skip);
    const void* psi = variable == 0 ? xs : other;
    const void* probe = variable == 0 ? other : xs;
    for (int lo = 0; lo < nscan; lo += chunk) {
      const int m = nscan - lo < chunk ? nscan - lo : chunk;
      if (det == 128) {
        // whole-tile forward (far plane stored) + the cost of that far plane:
        // the two launches of a host-side trial at this size
        const TkProbe P = tk_make_probe(probe, 0, nullptr, nullptr, 0, 0, S, det);
        int rc = launch_fwd128_lds((const cf*)psi, scan + 2L * lo, P, (cf*)scratch, nullptr, m, S,
                                   H, W, fwd_scale, stream, nullptr, skip);
        if (rc) return rc;
        rc = tk_farplane_gradient(scratch, (const float*)data + (size_t)lo * det * det, nullptr,
                                  nullptr, costs + lo, m, S, det, 0, 0, 1.0f, (long)det * det,
                                  stream, skip);
        if (rc) return rc;
        continue;
      }
      int rc = tk_fwd_pass1(psi, scan + 2L * lo, probe, 0, nullptr, nullptr, nullptr, 0, 0,
                            scratch, nullptr, m, S, det, det, H, W, stream, skip);
      if (rc) return rc;
      rc = tk_fwd_gradient_scale(scratch, (const char*)data + dsz * (size_t)lo * det * det,
                                 data_u16, nullptr, nullptr, nullptr, costs + lo, nullptr, m, S,
                                 det, fwd_scale, 0, 1.0f, (long)det * det, stream, skip);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(ls_decide_kernel, dim3(1), dim3(256), 0, stream, costs, nscan,
                       1.0 / count, shrink, k + 1 == nslots, state, skip);
  }
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------- the same line search, all steps at once
// The far plane is LINEAR in the variable a line search moves along: with
// A = F(x) and B = F(d) (the forward model applied to the direction in place
// of the object, or of the probe), F(x + s d) = A + s B for every step length
// s.  A is the hand-off the gradient pass at x has just left behind; B costs
// ONE forward pass 1; the intensity of a trial is the quadratic
//   I(s) = sum_m |A_m|^2 + 2 s sum_m Re(conj(A_m) B_m) + s^2 sum_m |B_m|^2
// in s per pixel, so one column pass over the TWO hand-offs gives the costs of
// x and of x + step d, x + step/2 d, ... (TK_LS_STEPS of them) together, and
// one small kernel takes the decision of the backtracking search
// (opt.py:216-278): the first of those step lengths whose cost is no larger
// than the cost at x.  Same candidates, same rule, same result as
// tike_cgrad_line_search up to float32 rounding -- for one forward pass and
// one two-stream column pass instead of a forward pass per trial.
constexpr int TK_LS_STEPS = 8;   // step lengths per pass over the hand-offs
constexpr int TK_LS_PASSES = 2;  // passes enqueued (the second returns at once if the first accepted)
constexpr int TK_LS_ROWS = TK_LS_STEPS * TK_LS_PASSES + 1;  // cost rows: x, then every step

// gaussian cost terms of RB pixels at step0 / 2^k, k < K (rows 1..K) and, FIRST,
// at step 0 (row 0).  v_sqrt_f32 (1 ulp) instead of the correctly rounded sqrtf
// (a dozen instructions each): K x RB square roots per thread are what this
// kernel issues most, and the cost at x it is compared with is formed the same way.
template <int K, int RB, bool FIRST, class DT>
__device__ __forceinline__ void tk_ksteps_costs(const float (&I0)[RB], const float (&C)[RB],
                                                const float (&I1)[RB], const DT (&raw)[RB],
                                                float step0, float (&acc)[K + 1]) {
#pragma unroll
  for (int p = 0; p < RB; ++p) {
    const float sd = __builtin_amdgcn_sqrtf((float)raw[p]);
    if (FIRST) {
      const float t0 = __builtin_amdgcn_sqrtf(I0[p]) - sd;
      acc[0] = fmaf(t0, t0, acc[0]);
    }
    const float c2 = 2.0f * C[p];
    float s = step0;
#pragma unroll
    for (int k = 0; k < K; ++k, s *= 0.5f) {
      const float I = fmaxf(fmaf(s, fmaf(s, I1[p], c2), I0[p]), 0.0f);
      const float t = __builtin_amdgcn_sqrtf(I) - sd;
      acc[k + 1] = fmaf(t, t, acc[k + 1]);
    }
  }
}

// per-thread sums -> one atomic each into costs_k[row * stride + n]; acc[0] is
// row 0 (FIRST only), acc[1..K] are rows row1 .. row1 + K - 1
// (deterministic mode: `part` != nullptr receives the contribution of slot
// `slot` of `nslots` per (row, pattern) -- part[(slot * TK_LS_ROWS + row) *
// stride + n] -- and ls_costs_finish_kernel adds the slots in order)
template <int K, bool FIRST>
__device__ __forceinline__ void tk_ksteps_emit(float (&acc)[K + 1], float (*red)[K + 1],
                                               float* __restrict__ costs_k, long stride, long n,
                                               int row1, float inv_nmeasured,
                                               float* __restrict__ part = nullptr,
                                               int slot = 0) {
#pragma unroll
  for (int k = FIRST ? 0 : 1; k <= K; ++k) acc[k] = tk_wave_sum(acc[k]);
  __syncthreads();  // the previous item's sums have been read
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k <= K; ++k) red[threadIdx.x >> 6][k] = acc[k];
  }
  __syncthreads();
  const int k = threadIdx.x;
  if (k <= K && (FIRST || k > 0)) {
    const int row = k == 0 ? 0 : row1 + k - 1;
    const float v = (red[0][k] + red[1][k] + red[2][k] + red[3][k]) * inv_nmeasured;
    if (part != nullptr)
      part[((long)slot * TK_LS_ROWS + row) * stride + n] = v;
    else
      unsafeAtomicAdd(&costs_k[row * stride + n], v);
  }
}

// 256^2 / 512^2: the column pass of fwd_gradient_scale_kernel over the
// hand-offs of x (col_a) and of the direction (col_b)
template <int N, class DT, bool FIRST>
__global__ __launch_bounds__(256, 2) void ls_ksteps_colpass_kernel(
    const cf* __restrict__ col_a, const cf* __restrict__ col_b, const DT* __restrict__ data,
    float* __restrict__ costs_k, long stride, long nitem, int S, float scale,
    float inv_nmeasured, int row1, const double* __restrict__ state,
    float* __restrict__ part) {
  constexpr int RB = N / 16, NH = N / 256, K = TK_LS_STEPS;
  __shared__ float red[4][K + 1];
  if (!FIRST && state[2] != 0.0) return;  // an earlier pass has accepted a step
  const float s2 = scale * scale;
  const float step0 = (float)state[1];
  for (long v = blockIdx.x; v < nitem; v += gridDim.x) {
    const int hb = (int)(v % NH);
    const int k1 = (int)((v / NH) & 15);
    const long n = nitem / (16 * NH) - 1 - v / (16 * NH);  // descending, as its siblings
    const int t = hb * 256 + threadIdx.x;
    float I0[RB], C[RB], I1[RB];
#pragma unroll
    for (int k2 = 0; k2 < RB; ++k2) I0[k2] = C[k2] = I1[k2] = 0.f;
    for (int s = 0; s < S; ++s) {
      const long off = (n * S + s) * (long)N * N + k1 * N + t;
      cf a[RB], b[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) a[r] = tk_ld_stream(col_a + off + (long)(16 * r) * N);
#pragma unroll
      for (int r = 0; r < RB; ++r) b[r] = tk_ld_stream(col_b + off + (long)(16 * r) * N);
      Dft<RB, false>::run(a);
      Dft<RB, false>::run(b);
#pragma unroll
      for (int k2 = 0; k2 < RB; ++k2) {
        I0[k2] += norm2(a[k2]) * s2;
        C[k2] += (a[k2].x * b[k2].x + a[k2].y * b[k2].y) * s2;
        I1[k2] += norm2(b[k2]) * s2;
      }
    }
    DT raw[RB];
    unsigned bits;
    tk_request_data<N, RB>(data, nullptr, n, k1, t, raw, bits);
    float acc[K + 1];
#pragma unroll
    for (int k = 0; k <= K; ++k) acc[k] = 0.f;
    tk_ksteps_costs<K, RB, FIRST>(I0, C, I1, raw, step0, acc);
    tk_ksteps_emit<K, FIRST>(acc, red, costs_k, stride, n, row1, inv_nmeasured, part,
                             k1 * NH + hb);
  }
}

// deterministic mode: costs_k[row][n] = sum over the slots, in slot order
__global__ __launch_bounds__(256) void ls_costs_finish_kernel(float* __restrict__ costs_k,
                                                              const float* __restrict__ part,
                                                              long stride, int n0, int n1,
                                                              int row_first, int row1,
                                                              int nslots) {
  constexpr int K = TK_LS_STEPS;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < (long)(K + 1) * (n1 - n0);
       i += gridDim.x * 256L) {
    const int k = (int)(i / (n1 - n0));
    const long n = n0 + i % (n1 - n0);
    if (k == 0 && !row_first) continue;
    const int row = k == 0 ? 0 : row1 + k - 1;
    float s = 0.f;
    for (int c = 0; c < nslots; ++c) s += part[((long)c * TK_LS_ROWS + row) * stride + n];
    costs_k[row * stride + n] = s;
  }
}

// stored far planes (128^2): a workgroup covers TK_FG_PIX pixels of one position
template <bool FIRST>
__global__ __launch_bounds__(256) void ls_ksteps_farplane_kernel(
    const cf* __restrict__ far_a, const cf* __restrict__ far_b, const float* __restrict__ data,
    float* __restrict__ costs_k, long stride, int S, long npix, float inv_nmeasured, int row1,
    const double* __restrict__ state, float* __restrict__ part) {
  constexpr int K = TK_LS_STEPS;
  __shared__ float red[4][K + 1];
  if (!FIRST && state[2] != 0.0) return;
  const float step0 = (float)state[1];
  const long n = blockIdx.y;
  const cf* __restrict__ FA = far_a + n * S * npix;
  const cf* __restrict__ FB = far_b + n * S * npix;
  const long p0 = (long)blockIdx.x * TK_FG_PIX;
  const long p1 = p0 + TK_FG_PIX < npix ? p0 + TK_FG_PIX : npix;
  float acc[K + 1];
#pragma unroll
  for (int k = 0; k <= K; ++k) acc[k] = 0.f;
  for (long p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    float I0[1] = {0.f}, C[1] = {0.f}, I1[1] = {0.f};
    const float raw[1] = {data[n * npix + p]};
    for (int s = 0; s < S; ++s) {
      const cf a = FA[s * npix + p], b = FB[s * npix + p];
      I0[0] += norm2(a);
      C[0] += a.x * b.x + a.y * b.y;
      I1[0] += norm2(b);
    }
    tk_ksteps_costs<K, 1, FIRST>(I0, C, I1, raw, step0, acc);
  }
  tk_ksteps_emit<K, FIRST>(acc, red, costs_k, stride, n, row1, inv_nmeasured, part,
                           (int)blockIdx.x);
}

// One workgroup: the means of a pass's cost rows, then the backtracking
// decision.  state { fx, step, done, trials, failures } as in ls_decide_kernel.
// First pass: fx on entry is ignored -- the cost at x is row 0, formed with the
// same arithmetic as the trials it is compared with -- and kept in state[0] for
// the passes behind it.  A pass that accepts nothing leaves step = the next
// length to try; the last one also counts a failure.
__global__ __launch_bounds__(256) void ls_pick_kernel(const float* __restrict__ costs_k,
                                                      long stride, int n, double inv_count,
                                                      int row1, int first, int last,
                                                      double* __restrict__ state,
                                                      int* __restrict__ accepted) {
  constexpr int K = TK_LS_STEPS;
  __shared__ double red[256];
  __shared__ double mean[K + 1];
  if (!first && state[2] != 0.0) return;
  for (int k = first ? 0 : 1; k <= K; ++k) {
    const float* __restrict__ row = costs_k + (k == 0 ? 0 : row1 + k - 1) * stride;
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += (double)row[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) mean[k] = red[0] * inv_count;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double fx = first ? mean[0] : state[0];
    float s = (float)state[1];
    int pick = -1;
    for (int k = 0; k < K; ++k, s *= 0.5f) {
      if (mean[k + 1] <= fx) {
        pick = k;
        break;
      }
    }
    if (pick >= 0) {
      state[0] = mean[pick + 1];
      state[1] = (double)s;
      state[2] = 1.0;
      state[3] += (double)(pick + 1);
      *accepted = 1;
    } else {
      state[0] = fx;
      state[1] = (double)s;  // step / 2^K: the next length to try
      state[2] = 0.0;
      state[3] += (double)K;
      if (last) state[4] += 1.0;
    }
  }
}

// Several ranks: the sums of a pass's cost rows over THIS rank's positions,
// to be all-reduced between the cost pass and the decision (row 0 only for
// the first pass).  One workgroup; sums (TK_LS_ROWS doubles).
__global__ __launch_bounds__(256) void ls_rowsum_kernel(const float* __restrict__ costs_k,
                                                        long stride, int n, int row1, int first,
                                                        const double* __restrict__ state,
                                                        double* __restrict__ sums) {
  constexpr int K = TK_LS_STEPS;
  __shared__ double red[256];
  const bool skip = !first && state[2] != 0.0;  // accepted already: leave zeros
  for (int k = first ? 0 : 1; k <= K; ++k) {
    const int rowi = k == 0 ? 0 : row1 + k - 1;
    double a = 0.0;
    if (!skip)
      for (int i = threadIdx.x; i < n; i += 256) a += (double)costs_k[rowi * stride + i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) sums[rowi] = red[0];
    __syncthreads();
  }
}

// The decision of ls_pick_kernel from (all-reduced) row sums.
__global__ void ls_pick_sums_kernel(const double* __restrict__ sums, double inv_count, int row1,
                                    int first, int last, double* __restrict__ state,
                                    int* __restrict__ accepted) {
  constexpr int K = TK_LS_STEPS;
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (!first && state[2] != 0.0) return;
  const double fx = first ? sums[0] * inv_count : state[0];
  float s = (float)state[1];
  int pick = -1;
  for (int k = 0; k < K; ++k, s *= 0.5f) {
    if (sums[row1 + k] * inv_count <= fx) {
      pick = k;
      break;
    }
  }
  if (pick >= 0) {
    state[0] = sums[row1 + pick] * inv_count;
    state[1] = (double)s;
    state[2] = 1.0;
    state[3] += (double)(pick + 1);
    *accepted = 1;
  } else {
    state[0] = fx;
    state[1] = (double)s;
    state[2] = 0.0;
    state[3] += (double)K;
    if (last) state[4] += 1.0;
  }
}

// xs = x + step d with the accepted step (x itself when none was)
__global__ __launch_bounds__(256) void ls_apply_kernel(const cf* __restrict__ x,
                                                       const cf* __restrict__ d,
                                                       cf* __restrict__ xs, long n,
                                                       const double* __restrict__ state) {
  const float a = state[2] != 0.0 ? (float)state[1] : 0.0f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const cf v = d[i];
    xs[i] = mk(x[i].x + a * v.x, x[i].y + a * v.y);
  }
}

extern "C" int tike_cgrad_line_search_linear(int variable, const void* x, const void* d, void* xs,
                                             const void* other, const float* scan,
                                             const void* data, int data_u16, void* far_a,
                                             int a_valid, void* far_b, float* costs_k,
                                             int nscan, int chunk, int S, int det, int H, int W,
                                             float fwd_scale, double count, double* state,
                                             int stage, double* sums, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 1 && chunk >= 1 && S >= 1 && H >= 1 && W >= 1 && count > 0 &&
               (variable == 0 || variable == 1));
  TK_CHECK_ARG(x && d && xs && other && scan && data && far_a && far_b && far_a != far_b &&
               costs_k && state);
  // stage 0: the whole search (one rank).  Several ranks, whose cost sums
  // must be all-reduced between a cost pass and its decision: 1 = first cost
  // pass -> sums; 2 = first decision from sums; 3 = second cost pass -> sums;
  // 4 = second decision from sums, then xs.
  TK_CHECK_ARG(stage >= 0 && stage <= 4 && (stage == 0 || sums != nullptr));
  if (det != 128 && det != 256 && det != 512) return TK_ERR_UNSUPPORTED;
  if (det == 128 && data_u16) return TK_ERR_UNSUPPORTED;  // the 128^2 cost kernel reads float32
  const long n = variable == 0 ? (long)H * W : (long)S * det * det;
  // (+ one word behind the rows: raised once a step is accepted -- the forward
  // passes of a later pass over a several-chunk minibatch read it and return)
  int* accepted = reinterpret_cast<int*>(costs_k + (size_t)TK_LS_ROWS * nscan);
  if (stage <= 1) {
    hipError_t e = hipMemsetAsync(
        costs_k, 0, sizeof(float) * ((size_t)TK_LS_ROWS * nscan + 1), stream);
    if (e != hipSuccess) return (int)e;
  }
  // deterministic mode: every (row, pattern) cost has `nslots` contributors --
  // their values go to the caller's scratch buffer and are added in slot order
  const int nslots = det == 128 ? (int)(((long)det * det + TK_FG_PIX - 1) / TK_FG_PIX)
                                : 16 * (det / 256);
  float* part = nullptr;
  if (tk_deterministic()) {
    part = tk_det_scratch(sizeof(float) * (size_t)nslots * TK_LS_ROWS * nscan);
    if (part == nullptr) return TK_ERR_ARG;  // scratch buffer too small
  }
  const bool reuse = a_valid && nscan <= chunk;  // the gradient pass left F(x) in far_a
  const bool resident = nscan <= chunk;          // one chunk: both hand-offs stay put
  const size_t dsz = data_u16 ? 2 : 4;
  const float inv = 1.0f / (float)((long)det * det);
  // forward model of the direction: d in place of the variable
  const void* psi_b = variable == 0 ? d : other;
  const void* probe_b = variable == 0 ? other : d;
  const void* psi_a = variable == 0 ? x : other;
  const void* probe_a = variable == 0 ? other : x;
  static_assert(TK_LS_PASSES == 2, "stages 1-4 name two passes");
  for (int pass = 0; pass < TK_LS_PASSES; ++pass) {
    const int row1 = 1 + pass * TK_LS_STEPS;
    const bool costs_now = stage == 0 || stage == 1 + 2 * pass;
    const bool decide_now = stage == 0 || stage == 2 + 2 * pass;
    if (!costs_now && !decide_now) continue;
    for (int lo = 0; costs_now && lo < nscan; lo += chunk) {
      const int m = nscan - lo < chunk ? nscan - lo : chunk;
      const float* sc = scan + 2L * lo;
      // the hand-offs of a chunk: formed in the first pass; a later pass (rare:
      // the first one accepted nothing) finds them in place unless the
      // minibatch has several chunks, which share the two buffers
      const bool form = pass == 0 || !resident;
      if (det == 128) {
        if (form && !(reuse && pass == 0)) {
          const TkProbe PA = tk_make_probe(probe_a, 0, nullptr, nullptr, 0, 0, S, det);
          int rc = launch_fwd128_lds((const cf*)psi_a, sc, PA, (cf*)far_a, nullptr, m, S, H, W,
                                     fwd_scale, stream, nullptr, pass ? accepted : nullptr);
          if (rc) return rc;
        }
        if (form) {
          const TkProbe PB = tk_make_probe(probe_b, 0, nullptr, nullptr, 0, 0, S, det);
          int rc = launch_fwd128_lds((const cf*)psi_b, sc, PB, (cf*)far_b, nullptr, m, S, H, W,
                                     fwd_scale, stream, nullptr, pass ? accepted : nullptr);
          if (rc) return rc;
        }
        const long npix = (long)det * det;
        const dim3 grid((unsigned)((npix + TK_FG_PIX - 1) / TK_FG_PIX), (unsigned)m);
        const float* dchunk = (const float*)data + (size_t)lo * npix;
        if (pass == 0)
          hipLaunchKernelGGL(ls_ksteps_farplane_kernel<true>, grid, dim3(256), 0, stream,
                             (const cf*)far_a, (const cf*)far_b, dchunk, costs_k + lo,
                             (long)nscan, S, npix, inv, row1, state, part ? part + lo : part);
        else
          hipLaunchKernelGGL(ls_ksteps_farplane_kernel<false>, grid, dim3(256), 0, stream,
                             (const cf*)far_a, (const cf*)far_b, dchunk, costs_k + lo,
                             (long)nscan, S, npix, inv, row1, state, part ? part + lo : part);
        if (part)
          hipLaunchKernelGGL(ls_costs_finish_kernel, dim3(tk_grid((long)(m * 9 + 255) / 256, 8)),
                             dim3(256), 0, stream, costs_k, part, (long)nscan, lo, lo + m,
                             (int)(pass == 0), row1, nslots);
        continue;
      }
      if (form && !(reuse && pass == 0)) {
        int rc = tk_fwd_pass1(psi_a, sc, probe_a, 0, nullptr, nullptr, nullptr, 0, 0, far_a,
                              nullptr, m, S, det, det, H, W, stream, pass ? accepted : nullptr);
        if (rc) return rc;
      }
      if (form) {
        int rc = tk_fwd_pass1(psi_b, sc, probe_b, 0, nullptr, nullptr, nullptr, 0, 0, far_b,
                              nullptr, m, S, det, det, H, W, stream, pass ? accepted : nullptr);
        if (rc) return rc;
      }
      const long nitem = (long)m * 16 * (det / 256);
      const dim3 grid(tk_grid(nitem, 32)), block(256);
      const char* dchunk = (const char*)data + dsz * (size_t)lo * det * det;
#define TK_LSK(N, DT, FIRST)                                                                  \
  hipLaunchKernelGGL((ls_ksteps_colpass_kernel<N, DT, FIRST>), grid, block, 0, stream,           \
                     (const cf*)far_a, (const cf*)far_b, (const DT*)dchunk, costs_k + lo,        \
                     (long)nscan, nitem, S, fwd_scale, inv, row1, state, part ? part + lo : part)
#define TK_LSK_N(N, DT)      \
  do {                       \
    if (pass == 0)           \
      TK_LSK(N, DT, true);   \
    else                     \
      TK_LSK(N, DT, false);  \
  } while (0)
      if (det == 256 && data_u16)
        TK_LSK_N(256, unsigned short);
      else if (det == 256)
        TK_LSK_N(256, float);
      else if (data_u16)
        TK_LSK_N(512, unsigned short);
      else
        TK_LSK_N(512, float);
#undef TK_LSK_N
#undef TK_LSK
      if (part)
        hipLaunchKernelGGL(ls_costs_finish_kernel, dim3(tk_grid((long)(m * 9 + 255) / 256, 8)),
                           dim3(256), 0, stream, costs_k, part, (long)nscan, lo, lo + m,
                           (int)(pass == 0), row1, nslots);
    }
    if (stage == 0)
      hipLaunchKernelGGL(ls_pick_kernel, dim3(1), dim3(256), 0, stream, costs_k, (long)nscan,
                         nscan, 1.0 / count, row1, (int)(pass == 0),
                         (int)(pass + 1 == TK_LS_PASSES), state, accepted);
    else if (costs_now)
      hipLaunchKernelGGL(ls_rowsum_kernel, dim3(1), dim3(256), 0, stream, costs_k, (long)nscan,
                         nscan, row1, (int)(pass == 0), state, sums);
    else
      hipLaunchKernelGGL(ls_pick_sums_kernel, dim3(1), dim3(64), 0, stream, sums, 1.0 / count,
                         row1, (int)(pass == 0), (int)(pass + 1 == TK_LS_PASSES), state,
                         accepted);
  }
  if (stage != 0 && stage != 4) {
    TK_LAUNCH_CHECK();
    return TK_OK;
  }
  hipLaunchKernelGGL(ls_apply_kernel, dim3(tk_grid((n + 255) / 256, 8)), dim3(256), 0, stream,
                     (const cf*)x, (const cf*)d, (cf*)xs, n, state);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------- 128^2: the whole tile in LDS
// A 128 x 128 complex tile is 128 KiB: it fits the 160 KiB LDS of a CU, so the
// 2-D transform needs NO intermediate in memory -- the only HBM traffic of the
// forward operator is its output, written once, straight from registers.
// One workgroup of 1024 threads (one per CU) per position; per mode:
//   rows     thread (row, j) gathers its 16 pixels (bilinear taps * probe)
//            straight into the FFT register layout, runs the 128-point row
//            transform (radix 16 x 8, the exchange stays inside its wave) and
//            leaves the row spectrum in the LDS tile;
//   columns  the same 8 threads then own COLUMN `line`: its 128 elements
//            (stride LS) go through the same in-wave radix 16 x 8 plan, the
//            column itself being the exchange buffer -- no workgroup barrier
//            and no second trip of the tile through LDS -- and leave for the
//            far plane from registers (8 rows x 64 contiguous bytes per wave
//            store; the neighbouring wave writes the other half of each line).
//            Two workgroup barriers per mode (rows done / columns done).
// The intensity sum_s |F_s|^2 accumulates in registers across the modes.
// LDS row stride: 136 elements = 272 dwords = 16 (mod 64 banks), so the four
// rows a 32-lane read group touches (8 lanes x 16 dwords each) tile the 64
// banks exactly; 136 also holds the padded row (127 + 127/16 = 134).
// (TK_L128_LS, tk_l128_swizzle: fft_engine2.h -- csrc/pfa.hip runs the same tile)
template <bool WITH_I>
__global__ __launch_bounds__(1024, 4) void fwd128_lds_kernel(
    const cf* __restrict__ psi, const float* __restrict__ scan, const TkProbe probe,
    cf* __restrict__ farplane, float* __restrict__ intensity, cf* __restrict__ patches, int nscan,
    int S, int H, int W, float scale, const cf* __restrict__ twtab,
    const int* __restrict__ skip) {
  constexpr int N = 128, T = 8, LS = TK_L128_LS;
  static_assert(FftPlan<N>::E == 16 && LS >= N + N / 16, "row plan: 16 elements x 8 threads");
  __shared__ cf lds[N * LS + FftTwLds<N>::ELEMS];
  if (skip != nullptr && *skip != 0) return;  // speculative launch, not needed
  cf* twl = lds + N * LS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  typedef float tk_v4f __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  int line = tid / T, j = tid % T;
  asm volatile("" : "+v"(line), "+v"(j));
  const FftTwLds<N> tw{twl, j};
  __builtin_assume(j >= 0 && j < T && line >= 0 && line < N);
  const long PP = (long)N * N;
  const long total = (long)H * W;
  auto at = [](const cf* base, unsigned byte_off) -> const cf* {
    return reinterpret_cast<const cf*>(reinterpret_cast<const char*>(base) + byte_off);
  };
  for (long n = blockIdx.x; n < nscan; n += gridDim.x) {
    const TkCorner c = tk_corner(scan, n);
    const bool interior = c.sy >= 0 && c.sx >= 0 && c.sy + N < H && c.sx + N < W &&
                          total < (1L << 28);
    // ---- bilinear patch of row `line`, elements e = j + 8 i
    cf pv[16];
    if (interior) {
      // One 8-byte load per pixel and row: the tap to the right (x + 1) is the
      // neighbouring lane's pixel (lane + 1 holds x + 1 for j < 7; for j == 7
      // it is slot i + 1 of the lane with j = 0, seven lanes down), fetched
      // with DPP row shifts.  Slot 16 is the pixel x = 128 + j that closes the
      // row (only j = 0's is used).  Two batches of loads: row y, then row y+1.
      const unsigned g0 = (unsigned)((c.sy + line) * W + c.sx + j) * (unsigned)sizeof(cf);
      const unsigned g1 = g0 + (unsigned)W * (unsigned)sizeof(cf);
      // slot 16 of the lanes j > 0 would lie past x = 128: read x = 128 too
      const unsigned gx = (unsigned)(128 - j) * (unsigned)sizeof(cf);
      auto right = [&](const cf (&a)[17], int i) {
        // value of pixel x + 1 for slot i
        auto dpp = [](float v, int ctrl_shl) {
          return ctrl_shl
                     ? __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                                                     0, __builtin_bit_cast(int, v), 0x101, 0xF,
                                                     0xF, true))   // row_shl:1  (lane + 1)
                     : __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                                                     0, __builtin_bit_cast(int, v), 0x117, 0xF,
                                                     0xF, true));  // row_shr:7  (lane - 7)
        };
        const cf nx = mk(dpp(a[i].x, 1), dpp(a[i].y, 1));
        const cf wr = mk(dpp(a[i + 1].x, 0), dpp(a[i + 1].y, 0));
        return j < 7 ? nx : wr;
      };
      cf up[17], lo[17];
#pragma unroll
      for (int i = 0; i < 16; ++i) up[i] = *at(psi, g0 + 64 * i);
      up[16] = *at(psi, g0 + gx);
      asm volatile(""
                   : "+v"(up[0].x), "+v"(up[1].x), "+v"(up[2].x), "+v"(up[3].x), "+v"(up[4].x),
                     "+v"(up[5].x), "+v"(up[6].x), "+v"(up[7].x), "+v"(up[8].x), "+v"(up[9].x),
                     "+v"(up[10].x), "+v"(up[11].x), "+v"(up[12].x), "+v"(up[13].x),
                     "+v"(up[14].x), "+v"(up[15].x), "+v"(up[16].x));
#pragma unroll
      for (int i = 0; i < 16; ++i) lo[i] = *at(psi, g1 + 64 * i);
      lo[16] = *at(psi, g1 + gx);
      asm volatile(""
                   : "+v"(lo[0].x), "+v"(lo[1].x), "+v"(lo[2].x), "+v"(lo[3].x), "+v"(lo[4].x),
                     "+v"(lo[5].x), "+v"(lo[6].x), "+v"(lo[7].x), "+v"(lo[8].x), "+v"(lo[9].x),
                     "+v"(lo[10].x), "+v"(lo[11].x), "+v"(lo[12].x), "+v"(lo[13].x),
                     "+v"(lo[14].x), "+v"(lo[15].x), "+v"(lo[16].x));
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const cf b = right(up, i), e = right(lo, i);
        cf o = mk(up[i].x * c.w00, up[i].y * c.w00);
        o.x += b.x * c.w01;
        o.y += b.y * c.w01;
        o.x += lo[i].x * c.w10;
        o.y += lo[i].y * c.w10;
        o.x += e.x * c.w11;
        o.y += e.y * c.w11;
        pv[i] = o;
      }
    } else {
      const int y = c.sy + line;
      const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int x = c.sx + j + i * T;
        const bool ok = y >= 0 && y < H && x >= 0 && x < W;
        const int xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
        const cf o = tk_gather(psi, (long)yc * W + xc, W, total, c);
        pv[i] = ok ? o : mk(0.f, 0.f);
        __builtin_amdgcn_sched_barrier(0);  // rare path: one element in flight
      }
    }
    if (patches != nullptr) {
      // O_n for the gradient kernels (uniform branch; the solver's 128^2 path)
      cf* __restrict__ On = patches + n * PP + line * N + j;
#pragma unroll
      for (int i = 0; i < 16; ++i) tk_st_stream(On + i * T, pv[i]);
    }
    float I[16];
    if (WITH_I) {
#pragma unroll
      for (int i = 0; i < 16; ++i) I[i] = 0.f;
    }
    const unsigned pbo = (unsigned)(line * N + j) * (unsigned)sizeof(cf);
    for (int s = 0; s < S; ++s) {
      // probe of (position, mode)
      const cf* __restrict__ Pn = probe.probe + n * probe.pos_stride + s * PP;
      float w0 = 1.0f;
      if (probe.weights != nullptr) {
        if (probe.unique != nullptr && s < probe.Sm)
          Pn = probe.unique + (n * probe.Sm + s) * PP;
        else
          w0 = probe.weights[n * (long)(probe.C + 1) * probe.S + s];
      }
      cf v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = pv[i] * (*at(Pn, pbo + 64 * i) * w0);
      // ---- row transform (the row of the tile is its own exchange buffer),
      // spectrum into the tile at the swizzled columns
      cf* lbase = lds + line * LS;
      FftStageWave<N, false, 0>::run(v, lbase, j, tw);
      const int rsw = tk_l128_swizzle(line);
#pragma unroll
      for (int i = 0; i < 16; ++i) lbase[(j + i * T + rsw) & (N - 1)] = v[i];
      __syncthreads();
      // ---- column transform: the 8 threads that shared row `line` now share
      // COLUMN `line`; its 128 elements (stride LS) are their exchange buffer
      const int col = line;
      auto cat = [&](int e) { return e * LS + ((col + tk_l128_swizzle(e)) & (N - 1)); };
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = lds[cat(j + i * T)];
      FftStageWave<N, false, 0>::run_at(v, lds, j, tw, cat);
      cf* __restrict__ dst = farplane + (n * S + s) * PP + col;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const cf o = v[i] * scale;
        if (farplane != nullptr) tk_st_stream(dst + (long)(j + i * T) * N, o);
        if (WITH_I) I[i] += norm2(o);
      }
      __syncthreads();  // the next mode's rows overwrite the tile
    }
    if (WITH_I) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        tk_st_stream(intensity + n * PP + (long)(j + i * T) * N + line, I[i]);
    }
  }
}

static int launch_fwd128_lds(const cf* psi, const float* scan, const TkProbe& probe, cf* farplane,
                             float* intensity, int nscan, int S, int H, int W, float scale,
                             hipStream_t stream, cf* patches, const int* skip) {
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
  const dim3 grid(tk_grid(nscan, 1)), block(1024);
  if (intensity)
    hipLaunchKernelGGL((fwd128_lds_kernel<true>), grid, block, 0, stream, psi, scan, probe,
                       farplane, intensity, patches, nscan, S, H, W, scale, tw, skip);
  else
    hipLaunchKernelGGL((fwd128_lds_kernel<false>), grid, block, 0, stream, psi, scan, probe,
                       farplane, intensity, patches, nscan, S, H, W, scale, tw, skip);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

template <int N>
static int launch_fwd_v2(const cf* psi, const float* scan, const TkProbe& probe, cf* farplane,
                         long ntile, int S, int pw, int H, int W, float scale,
                         hipStream_t stream) {
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
  hipLaunchKernelGGL((ptycho_fwd_v2_kernel<N>), dim3(tk_grid(ntile, 4)), dim3(N), 0, stream, psi,
                     scan, probe, farplane, ntile, S, pw, H, W, scale, tw);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

template <int N>
static int launch_fwd(const cf* psi, const float* scan, const TkProbe& probe, cf* farplane,
                      long ntile, int S, int pw, int H, int W, float scale, hipStream_t stream) {
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
  hipLaunchKernelGGL((ptycho_fwd_kernel<N>), dim3(tk_grid(ntile, N >= 512 ? 2 : 4)),
                     dim3(FftPlan<N>::NT), 0, stream, psi, scan, probe, farplane, ntile, S, pw, H,
                     W, scale, tw);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// The forward column pass alone, in place on the far-plane array: rows
// {k1 + 16 r} of a tile in, the same rows out (fft_engine2.h pass 2), so a
// tile written by forward pass 1 becomes the far plane without a second array.
// Work item = (tile, k1, 256-column block); tiles in descending order (pass 1
// wrote them ascending: its last tiles are still in the Infinity Cache).
template <int N>
__global__ __launch_bounds__(256, N == 256 ? 4 : 2) void fwd_colpass_inplace_kernel(
    cf* far, long ntile, float scale) {
  constexpr int RB = N / 16, NH = N / 256;
  const long nitem = ntile * 16 * NH;
  for (long v = blockIdx.x; v < nitem; v += gridDim.x) {
    const int hb = (int)(v % NH);
    const int k1 = (int)((v / NH) & 15);
    const long tile = ntile - 1 - v / (16 * NH);
    cf* p = far + tile * (long)N * N + (long)k1 * N + hb * 256 + threadIdx.x;
    cf u[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) u[r] = p[(long)(16 * r) * N];
    Dft<RB, false>::run(u);
#pragma unroll
    for (int k2 = 0; k2 < RB; ++k2) tk_st_stream(p + (long)(16 * k2) * N, u[k2] * scale);
  }
}

extern "C" int tike_ptycho_fwd(const void* psi, const float* scan, const void* probe,
                               int probe_per_scan, const void* eigen_probe,
                               const float* eigen_weights, int num_eigen, int eigen_modes,
                               void* farplane, int nscan, int S, int pw, int det, int H, int W,
                               float scale, int sub_batch, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(psi && scan && probe && farplane);
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 1 && det >= pw && H >= 1 && W >= 1);
  TK_CHECK_ARG(!(eigen_weights && probe_per_scan));
  if (nscan == 0) return TK_OK;
  const TkProbe P = tk_make_probe(probe, probe_per_scan, eigen_probe, eigen_weights, num_eigen,
                                  eigen_modes, S, pw);
  const long ntile = (long)nscan * S;
  const cf* psi_ = (const cf*)psi;
  cf* far = (cf*)farplane;
  if (det == 128 && pw == 128 && !(eigen_weights && eigen_modes > 0))
    return launch_fwd128_lds(psi_, scan, P, far, nullptr, nscan, S, H, W, scale, stream, nullptr);
  if (det == 512 || det == 256) {
    // forward pass 1 (the patch of a row group gathered once for all modes,
    // eigen probes on the fly) straight into the far-plane array, then the
    // column pass in place -- two streaming kernels; faster than one workgroup
    // per tile / per position (below) for every mode count: 256^2 x 1 mode
    // 2.57 -> 2.75 M tiles/s, x 3 modes +22 %, 512^2 +47 %
    // ... in sub-batches of about 256 MiB of far plane: pass 1 keeps its
    // hand-off in the Infinity Cache (plain stores), the column pass that
    // follows reads it from there and overwrites it in place, so HBM sees the
    // far plane once (256^2 x 1 mode: 2.83 -> 2.95 M patterns/s; smaller
    // sub-batches lose more to their launches than the cache returns,
    // profiles/r04_experiments.md).  sub_batch = positions per sub-batch
    // (0: the 256 MiB default; < 0: one batch) -- an argument, the entry
    // reads no environment.
    const size_t tile_bytes = sizeof(cf) * (size_t)det * det;
    long sub = sub_batch > 0   ? sub_batch
               : sub_batch < 0 ? nscan
                               : (256L << 20) / (long)(tile_bytes * S);
    if (sub < 1) sub = 1;
    const bool keep = sub < nscan;
    for (long lo = 0; lo < nscan; lo += sub) {
      const int m = (int)(nscan - lo < sub ? nscan - lo : sub);
      int rc = tk_fwd_pass1(psi, scan + 2 * lo,
                            (const char*)probe +
                                (probe_per_scan ? sizeof(cf) * (size_t)S * pw * pw * lo : 0),
                            probe_per_scan, nullptr, eigen_probe,
                            eigen_weights ? eigen_weights + lo * (num_eigen + 1) * S : nullptr,
                            num_eigen, eigen_modes, (char*)farplane + tile_bytes * S * lo,
                            nullptr, m, S, pw, det, H, W, stream, nullptr, keep);
      if (rc) return rc;
      const long mt = (long)m * S;
      const long nitem = mt * 16 * (det / 256);
      cf* fm = far + lo * S * det * det;
      if (det == 256)
        hipLaunchKernelGGL((fwd_colpass_inplace_kernel<256>), dim3(tk_grid(nitem, 32)),
                           dim3(256), 0, stream, fm, mt, scale);
      else
        hipLaunchKernelGGL((fwd_colpass_inplace_kernel<512>), dim3(tk_grid(nitem, 32)),
                           dim3(256), 0, stream, fm, mt, scale);
    }
    TK_LAUNCH_CHECK();
    return TK_OK;
  }
  if (S > 1 && !(eigen_weights && eigen_modes > 0)) {
    // position-major kernel (patch gathered once per position and shared by
    // the modes, straight-line loader; with a single mode there is nothing to
    // share and the tile-major kernel below, with its higher occupancy, is
    // 7 % faster); the varying-probe case needs tike_varying_probe first and
    // is served by tike_ptycho_fwd_intensity
    // (256 / 512 never get here: the two streaming kernels above)
    if (det == 128)
      return launch_fwd_pos<128>(psi_, scan, P, far, nullptr, nscan, S, pw, H, W, scale, stream);
  }
  if (det == 128)
    return launch_fwd_v2<128>(psi_, scan, P, far, ntile, S, pw, H, W, scale, stream);
  switch (det) {
    case 32: return launch_fwd<32>(psi_, scan, P, far, ntile, S, pw, H, W, scale, stream);
    case 64: return launch_fwd<64>(psi_, scan, P, far, ntile, S, pw, H, W, scale, stream);
    case 1024: return launch_fwd<1024>(psi_, scan, P, far, ntile, S, pw, H, W, scale, stream);
    default: break;
  }
  // any other detector size: unfused gather*probe, then the generic DFT in place
  int rc = tk_conv_fwd(psi_, scan, P, far, nscan, S, pw, det, H, W, stream);
  if (rc) return rc;
  return tk_fft2(far, far, ntile, det, 0, scale, stream);
}

// ------------------------------------------------------- inverse + crop
template <int N>
__global__ __launch_bounds__(FftPlan<N>::NT, FftPlan<N>::MINW) void ifft2_crop_kernel(
    const cf* farplane, cf* work, cf* chi, long ntile,
    int pw, float scale, const cf* __restrict__ twtab) {
  using G = FftGeom<N>;
  __shared__ cf lds[G::LDS_ELEMS];
  FftTw<N> tw;
  const int pad = (N - pw) / 2;
  const int end = pad + pw;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const cf* src = farplane + tile * (long)N * N;
    cf* mid = work + tile * (long)N * N;
    cf* dst = chi + tile * (long)pw * pw;
    const FftLane<N, false> row = fft_lane<N, false>();
    tw.init(twtab, row.j);
    for (int g = 0; g < N; g += G::L) {
      fft_lines<N, true, false>(
          lds, row, tw, [&](int line, int e) { return src[(g + line) * N + e]; },
          [&](int line, int e, cf v) { mid[(g + line) * N + e] = v; });
    }
    __syncthreads();
    const FftLane<N, true> col = fft_lane<N, true>();
    tw.init(twtab, col.j);
    for (int g = 0; g < N; g += G::L) {
      if (g + G::L <= pad || g >= end) continue;  // columns outside the crop
      fft_lines<N, true, true>(
          lds, col, tw, [&](int line, int e) { return mid[e * N + g + line]; },
          [&](int line, int e, cf v) {
            const int py = e - pad, px = g + line - pad;
            if (py >= 0 && py < pw && px >= 0 && px < pw) tk_st_stream(dst + py * pw + px, v * scale);
          });
    }
    __syncthreads();
  }
}

// v2: pass 1 farplane -> work (must not alias), pass 2 in place / cropped.
#ifndef TK_ICROP_WAVES
#define TK_ICROP_WAVES 4
#endif
// PASS2 = false: stop after pass 1 (`work` then holds the input of the column
// pass, which tike_ifft2_pass2_gradients consumes).
template <int N, int MODE, bool PASS2 = true>
__global__ __launch_bounds__(N, (N <= 256 ? TK_ICROP_WAVES : 2)) void ifft2_crop_v2_kernel(
    const cf* __restrict__ farplane, cf* work, cf* chi, long ntile, int pw, float scale,
    const cf* __restrict__ twtab, const float* __restrict__ gscale, int S,
    const float* __restrict__ mode_scale, const unsigned char* __restrict__ measured) {
  using G2 = Fft2Geom<N>;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int pad = (N - pw) / 2;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const cf* __restrict__ src = farplane + tile * (long)N * N;
    cf* mid = work + tile * (long)N * N;
    cf* dst = chi + tile * (long)pw * pw;
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    // far-plane gradient applied on the fly: F_s * g, g per (position, pixel);
    // MODE 2 (poisson): times the step of this (position, mode) on measured pixels
    const float* __restrict__ gs = MODE ? gscale + (tile / S) * (long)N * N : nullptr;
    const float ms = MODE >= 2 ? mode_scale[tile] : 1.0f;
    for (int r = 0; r < G2::RB; ++r)
      fft2_pass1<N, true, !PASS2>(
          lds, twtab, tw, line, j, r,
          [&](int y, int e, auto) {
            const cf f = tk_ld_stream(src + y * N + e);
            if (MODE == 0) return f;
            float g = gs[y * N + e];
            // (MODE 3 = MODE 2 with a mask: its byte is requested with the
            // factor and the step selected -- never a test around the load)
            if (MODE == 2) g *= ms;
            if (MODE == 3) g *= measured[y * N + e] ? ms : 1.0f;
            return f * g;
          },
          mid);
    if constexpr (PASS2) {
      __syncthreads();
      for (int k1 = 0; k1 < 16; ++k1)
        fft2_pass2<N, true>(mid, k1, [&](int ky, int t, cf v) {
          const int py = ky - pad, px = t - pad;
          if (py >= 0 && py < pw && px >= 0 && px < pw)
            tk_st_stream(dst + py * pw + px, v * scale);
        });
      __syncthreads();
    }
  }
}

template <int N, bool PASS2 = true>
static int launch_icrop_v2(const cf* far, cf* work, cf* chi, long ntile, int pw, float scale,
                           hipStream_t stream, const float* gscale = nullptr, int S = 1,
                           const float* mode_scale = nullptr,
                           const unsigned char* measured = nullptr) {
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
#define TK_ICROP(MODE)                                                                        \
  hipLaunchKernelGGL((ifft2_crop_v2_kernel<N, MODE, PASS2>), dim3(tk_grid(ntile, 4)), dim3(N), \
                     0, stream, far, work, chi, ntile, pw, scale, tw, gscale, S, mode_scale, \
                     measured)
  if (gscale && mode_scale && measured)
    TK_ICROP(3);
  else if (gscale && mode_scale)
    TK_ICROP(2);
  else if (gscale)
    TK_ICROP(1);
  else
    TK_ICROP(0);
#undef TK_ICROP
  TK_LAUNCH_CHECK();
  return TK_OK;
}

__global__ __launch_bounds__(256) void crop_kernel(const cf* __restrict__ src,
                                                   cf* __restrict__ dst, long ntile, int det,
                                                   int pw) {
  const int pad = (det - pw) / 2;
  const long nrow = ntile * pw;
  for (long r = blockIdx.x; r < nrow; r += gridDim.x) {
    const long t = r / pw;
    const int py = (int)(r % pw);
    for (int px = threadIdx.x; px < pw; px += blockDim.x)
      dst[(t * pw + py) * pw + px] = src[(t * det + pad + py) * (long)det + pad + px];
  }
}

template <int N>
static int launch_icrop(const cf* far, cf* work, cf* chi, long ntile, int pw, float scale,
                        hipStream_t stream) {
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
  hipLaunchKernelGGL((ifft2_crop_kernel<N>), dim3(tk_grid(ntile, N >= 512 ? 2 : 4)),
                     dim3(FftPlan<N>::NT), 0, stream, far, work, chi, ntile, pw, scale, tw);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// work: (ntile, det, det) scratch for the intermediate; may alias farplane
// (overwrite) and, when pw == det, chi may alias work.
extern "C" int tike_ifft2_crop(const void* farplane, void* work, void* chi, long ntile, int det,
                               int pw, float scale, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(farplane && work && chi && ntile >= 0 && pw >= 1 && det >= pw);
  TK_CHECK_ARG(!(chi == work && pw != det));
  if (ntile == 0) return TK_OK;
  const cf* far = (const cf*)farplane;
  cf* wk = (cf*)work;
  cf* out = (cf*)chi;
  if ((const cf*)wk != far) {
    switch (det) {
      case 128: return launch_icrop_v2<128>(far, wk, out, ntile, pw, scale, stream);
      case 256: return launch_icrop_v2<256>(far, wk, out, ntile, pw, scale, stream);
      case 512: return launch_icrop_v2<512>(far, wk, out, ntile, pw, scale, stream);
      default: break;
    }
  }
  switch (det) {
    case 32: return launch_icrop<32>(far, wk, out, ntile, pw, scale, stream);
    case 64: return launch_icrop<64>(far, wk, out, ntile, pw, scale, stream);
    case 128: return launch_icrop<128>(far, wk, out, ntile, pw, scale, stream);
    case 256: return launch_icrop<256>(far, wk, out, ntile, pw, scale, stream);
    case 512: return launch_icrop<512>(far, wk, out, ntile, pw, scale, stream);
    case 1024: return launch_icrop<1024>(far, wk, out, ntile, pw, scale, stream);
    default: break;
  }
  int rc = tk_fft2(far, wk, ntile, det, 1, scale, stream);
  if (rc) return rc;
  if (out == wk) return TK_OK;
  hipLaunchKernelGGL(crop_kernel, dim3(tk_grid(ntile * pw, 16)), dim3(256), 0, stream, wk, out,
                     ntile, det, pw);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------- intensity / cost / gradient
// One workgroup per position.  For every detector pixel: I = sum_s |F_s|^2;
// cost contribution on measured pixels; F_s *= g with
//   gaussian: g = -(1 - sqrt(d) / (sqrt(I) + 1e-9))       (objective.py:31-44)
//   poisson:  g = -(1 - d / (I + 1e-9))                   (objective.py:97-109)
// on measured pixels and g = (unmeasured_scaling - 1) elsewhere
// (lstsq.py:491-502).  Unmeasured pixels are selected by the mask and their
// data values (possibly NaN) are never read into the arithmetic.
// Grid: (pixel blocks, positions); a workgroup covers TK_FG_PIX pixels of one
// position and adds its share of the cost with one atomic.

template <int MODEL, bool GRAD>
__global__ __launch_bounds__(256) void farplane_gradient_kernel(
    cf* __restrict__ farplane, const float* __restrict__ data,
    const unsigned char* __restrict__ mask, float* __restrict__ intensity,
    const TkCostSink costs, int nscan, int S, int det, float unmeasured_scaling,
    float inv_nmeasured, const int* __restrict__ skip) {
  __shared__ float red[4];
  if (skip != nullptr && *skip != 0) return;  // speculative launch, not needed
  const long npix = (long)det * det;
  const long n = blockIdx.y;
  cf* __restrict__ F = farplane + n * S * npix;
  const float* __restrict__ d = data + n * npix;
  float cost = 0.f;
  const long p0 = (long)blockIdx.x * TK_FG_PIX;
  const long p1 = p0 + TK_FG_PIX < npix ? p0 + TK_FG_PIX : npix;
  for (long p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    // the count is requested WITH the wave values, not behind the mask test
    // (a load inside a branch waits for everything in front of the branch);
    // an unmeasured pixel may hold NaN: selected away, never multiplied
    const float dv = d[p];
    const bool measured = mask ? mask[p] != 0 : true;
    float I = 0.f;
    for (int s = 0; s < S; ++s) I += norm2(F[s * npix + p]);
    if (intensity) intensity[n * npix + p] = I;
    float g, term;
    if (MODEL == 0) {
      const float sI = sqrtf(I), sd = sqrtf(dv);
      const float diff = sI - sd;
      term = diff * diff;
      g = -(1.0f - sd / (sI + 1e-9f));
    } else {
      term = I - dv * logf(I + 1e-9f);
      g = -(1.0f - dv / (I + 1e-9f));
    }
    cost += measured ? term : 0.f;
    g = measured ? g : unmeasured_scaling - 1.0f;
    if (GRAD)
      for (int s = 0; s < S; ++s) F[s * npix + p] = F[s * npix + p] * g;
  }
  if (costs.costs) {
    cost = tk_block_sum256(cost, red);
    if (threadIdx.x == 0) tk_cost_add(costs, n, (int)blockIdx.x, cost * inv_nmeasured);
  }
}

static int tk_farplane_gradient(void* farplane, const float* data, const unsigned char* measured,
                                float* intensity, float* costs, int nscan, int S, int det,
                                int model, int apply_gradient, float unmeasured_scaling,
                                long num_measured, hipStream_t stream, const int* skip) {
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && det >= 1);
  TK_CHECK_ARG(model == 0 || model == 1);
  TK_CHECK_ARG(num_measured > 0);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(farplane && data);
  const float inv = 1.0f / (float)num_measured;
  const long npix = (long)det * det;
  const dim3 grid((unsigned)((npix + TK_FG_PIX - 1) / TK_FG_PIX), (unsigned)nscan), block(256);
  TkCostSink sink{costs, nullptr, 0};
  if (skip == nullptr) {
    int rc = tk_cost_sink(costs, nscan, (int)grid.x, stream, &sink);
    if (rc) return rc;
  } else if (costs) {
    hipError_t e = hipMemsetAsync(costs, 0, sizeof(float) * (size_t)nscan, stream);
    if (e != hipSuccess) return (int)e;
  }
#define TK_FG(M, G)                                                                          \
  hipLaunchKernelGGL((farplane_gradient_kernel<M, G>), grid, block, 0, stream, (cf*)farplane, \
                     data, measured, intensity, sink, nscan, S, det, unmeasured_scaling, inv,     \
                     skip)
  if (model == 0 && apply_gradient) TK_FG(0, true);
  if (model == 0 && !apply_gradient) TK_FG(0, false);
  if (model == 1 && apply_gradient) TK_FG(1, true);
  if (model == 1 && !apply_gradient) TK_FG(1, false);
#undef TK_FG
#undef TK_FG_K
  TK_LAUNCH_CHECK();
  return tk_cost_finish(sink, nscan, stream);
}

extern "C" int tike_farplane_gradient(void* farplane, const float* data,
                                      const unsigned char* measured, float* intensity,
                                      float* costs, int nscan, int S, int det, int model,
                                      int apply_gradient, float unmeasured_scaling,
                                      long num_measured, void* stream_) {
  TK_ENTER();
  return tk_farplane_gradient(farplane, data, measured, intensity, costs, nscan, S, det, model,
                              apply_gradient, unmeasured_scaling, num_measured,
                              (hipStream_t)stream_, nullptr);
}

// ------------------------------------------------ gradient scale from intensity
// gscale[n][p] = -(1 - sqrt(d)/(sqrt(I)+1e-9))  (gaussian; poisson: -(1 - d/(I+1e-9)))
// on measured pixels, (unmeasured_scaling - 1) elsewhere; costs[n] = mean over
// measured pixels of the per-pixel cost (objective.py:11-124, lstsq.py:444-502).
template <int MODEL>
__global__ __launch_bounds__(256) void gradient_scale_kernel(
    const float* __restrict__ intensity, const float* __restrict__ data,
    const unsigned char* __restrict__ mask, float* __restrict__ gscale,
    const TkCostSink costs, int det, float unmeasured_scaling, float inv_nmeasured) {
  __shared__ float red[4];
  const long npix = (long)det * det;
  const long n = blockIdx.y;
  float cost = 0.f;
  const long p0 = (long)blockIdx.x * TK_FG_PIX;
  const long p1 = p0 + TK_FG_PIX < npix ? p0 + TK_FG_PIX : npix;
  for (long p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    const float I = intensity[n * npix + p];
    const float dv = data[n * npix + p];  // with I, not behind the mask test
    const bool measured = mask ? mask[p] != 0 : true;
    float g, term;
    if (MODEL == 0) {
      const float sI = sqrtf(I), sd = sqrtf(dv);
      const float diff = sI - sd;
      term = diff * diff;
      g = -(1.0f - sd / (sI + 1e-9f));
    } else {
      term = I - dv * logf(I + 1e-9f);
      g = -(1.0f - dv / (I + 1e-9f));
    }
    cost += measured ? term : 0.f;
    gscale[n * npix + p] = measured ? g : unmeasured_scaling - 1.0f;
  }
  if (costs.costs) {
    cost = tk_block_sum256(cost, red);
    if (threadIdx.x == 0) tk_cost_add(costs, n, (int)blockIdx.x, cost * inv_nmeasured);
  }
}

extern "C" int tike_gradient_scale(const float* intensity, const float* data,
                                   const unsigned char* measured, float* gscale, float* costs,
                                   int nscan, int det, int model, float unmeasured_scaling,
                                   long num_measured, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && det >= 1 && (model == 0 || model == 1) && num_measured > 0);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(intensity && data && gscale);
  const float inv = 1.0f / (float)num_measured;
  const long npix = (long)det * det;
  const dim3 grid((unsigned)((npix + TK_FG_PIX - 1) / TK_FG_PIX), (unsigned)nscan), block(256);
  TkCostSink sink;
  int rc = tk_cost_sink(costs, nscan, (int)grid.x, stream, &sink);
  if (rc) return rc;
  if (model == 0)
    hipLaunchKernelGGL((gradient_scale_kernel<0>), grid, block, 0, stream, intensity, data,
                       measured, gscale, sink, det, unmeasured_scaling, inv);
  else
    hipLaunchKernelGGL((gradient_scale_kernel<1>), grid, block, 0, stream, intensity, data,
                       measured, gscale, sink, det, unmeasured_scaling, inv);
  TK_LAUNCH_CHECK();
  return tk_cost_finish(sink, nscan, stream);
}

// IFFT2 + crop of (farplane * gscale): gscale (ntile / S, det, det) f32 is
// shared by the S modes of a position.  work must not alias farplane.
extern "C" int tike_ifft2_crop_scaled(const void* farplane, const float* gscale, int S,
                                      void* work, void* chi, long ntile, int det, int pw,
                                      float scale, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(ntile >= 0 && S >= 1 && pw >= 1 && det >= pw);
  if (ntile == 0) return TK_OK;
  TK_CHECK_ARG(farplane && gscale && work && chi && work != farplane && ntile % S == 0);
  TK_CHECK_ARG(!(chi == work && pw != det));
  switch (det) {
    case 128:
      return launch_icrop_v2<128>((const cf*)farplane, (cf*)work, (cf*)chi, ntile, pw, scale,
                                  stream, gscale, S);
    case 256:
      return launch_icrop_v2<256>((const cf*)farplane, (cf*)work, (cf*)chi, ntile, pw, scale,
                                  stream, gscale, S);
    case 512:
      return launch_icrop_v2<512>((const cf*)farplane, (cf*)work, (cf*)chi, ntile, pw, scale,
                                  stream, gscale, S);
    default:
      return TK_ERR_UNSUPPORTED;
  }
}

// XCD-aware tile order for kernels whose S mode tiles of one position share a
// per-position table (gscale): workgroups are dealt round-robin over the 8
// XCDs, so virtual block v runs on XCD v % 8; giving the S modes of a position
// to consecutive blocks OF ONE XCD lets that XCD's L2 fetch the table once
// instead of every XCD fetching it.  v ranges over ceil(nscan/8)*8*S; returns
// -1 for the padding.  Placement only affects speed, never results.
__device__ __forceinline__ long tk_xcd_tile(long v, int S, long nscan) {
  const long xcd = v & 7, slot = v >> 3;
  const long p = (slot / S) * 8 + xcd;
  return p < nscan ? p * S + slot % S : -1;
}

// ------------------------------------------- gradient + inverse, no far plane
// Consumes the column-pass input left by tike_ptycho_fwd_intensity_only.  Per
// tile and per k1 (a thread owns one column):
//   F[k1 + 16 k2] = radix-16 over r of rows 16r + k1      (forward column pass)
//   G = F * fwd_scale * gscale                            (lstsq.py:491-502)
// and the inverse transform starts on the same registers: with ky = k1 + 16 k2
// and y = ya + 16 yb,
//   w^-(ky y) = w_16^-(k2 ya) * w_N^-(k1 ya) * w_16^-(k1 yb),
//   A[ya] = w_N^-(k1 ya) * radix-16 over k2 of G          (registers)
//   rows (k1, ya): inverse row transforms                 (LDS, in-wave stages)
//   stored as rows 16 k1 + ya of `work`; after a barrier fft2_pass2 (radix-16
//   over k1, in place, rows {ya + 16 yb}) finishes, crops and scales.
// The far-plane waves are therefore never written to or read from memory.
#ifndef TK_GINV_WAVES
#define TK_GINV_WAVES 4
#endif
template <int N, int MODE, bool PASS2 = true>
__global__ __launch_bounds__(N, TK_GINV_WAVES) void grad_ifft2_crop_kernel(
    const cf* __restrict__ colin, cf* work, cf* chi, long ntile, int pw, float fwd_scale,
    float inv_scale, const cf* __restrict__ twtab, const float* __restrict__ gscale, int S,
    const float* __restrict__ mode_scale, const unsigned char* __restrict__ measured,
    int ksplit) {
  using G2 = Fft2Geom<N>;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int pad = (N - pw) / 2;
  const int t = threadIdx.x;
  const long nscan = ntile / S;
  // work item = (tile, group of 16 / ksplit values of k1); ksplit > 1 (only
  // without pass 2) lets a small launch fill the chip
  const long nvirt = ((nscan + 7) / 8) * 8 * S;
  const int kn = 16 / ksplit;
  for (long w = blockIdx.x; w < nvirt * ksplit; w += gridDim.x) {
    const long v = w % nvirt;
    const int kbeg = (int)(w / nvirt) * kn;
    const long tile = tk_xcd_tile(v, S, nscan);
    if (tile < 0) continue;  // uniform
    const cf* __restrict__ src = colin + tile * (long)N * N;
    cf* mid = work + tile * (long)N * N;
    cf* dst = chi + tile * (long)pw * pw;
    const float* __restrict__ gs = gscale + (tile / S) * (long)N * N;
    const float ms = MODE == 2 ? mode_scale[tile] : 1.0f;
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    // software pipelined over k1: the 16 rows of k1 + 1 are requested before
    // the butterflies of k1 (1.64 -> 1.60 ms; 7 registers go to scratch at the
    // 128-register cap, a third wave less per SIMD would cost more)
    cf un[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) un[r] = tk_ld_stream(src + (16 * r + kbeg) * N + t);
    for (int k1 = kbeg; k1 < kbeg + kn; ++k1) {
      cf u[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) u[r] = un[r];
      if (k1 + 1 < kbeg + kn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) un[r] = tk_ld_stream(src + (16 * r + k1 + 1) * N + t);
      }
      Dft<16, false>::run(u);
      if constexpr (MODE == 2) {
        // per-mode step on measured pixels: the 16 factors (and mask bytes)
        // are requested together, the step is SELECTED -- a test per pixel
        // around its mask load made 16 serial round trips of them (3.29 ms
        // per 1000 positions x 8 modes against 1.6 ms without the steps)
        float g[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) g[k2] = gs[(k1 + 16 * k2) * N + t];
        if (measured != nullptr) {  // uniform
          unsigned char mb[16];
#pragma unroll
          for (int k2 = 0; k2 < 16; ++k2) mb[k2] = measured[(k1 + 16 * k2) * N + t];
#pragma unroll
          for (int k2 = 0; k2 < 16; ++k2) g[k2] *= mb[k2] ? ms : 1.0f;
        } else {
#pragma unroll
          for (int k2 = 0; k2 < 16; ++k2) g[k2] *= ms;
        }
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) u[k2] = u[k2] * (g[k2] * fwd_scale);
      } else {
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
          const int p = (k1 + 16 * k2) * N + t;
          float g = gs[p] * fwd_scale;
          u[k2] = u[k2] * g;
        }
      }
      Dft<16, true>::run(u);
#pragma unroll
      for (int ya = 1; ya < 16; ++ya) u[ya] = mul_tw<true>(u[ya], twtab[N + k1 * ya]);
      fft2_rows_from_columns<N, true, !PASS2>(lds, tw, line, j, u, mid + (long)(16 * k1) * N);
    }
    if constexpr (PASS2) {
      __syncthreads();
      for (int ya = 0; ya < 16; ++ya)
        fft2_pass2<N, true>(mid, ya, [&](int y, int x, cf v) {
          const int py = y - pad, px = x - pad;
          if (py >= 0 && py < pw && px >= 0 && px < pw)
            tk_st_stream(dst + py * pw + px, v * inv_scale);
        });
      __syncthreads();
    }
  }
}

// ---- column pass + gradient factor + inverse pass 1 in ONE kernel (256^2)
// tike_fwd_gradient_scale and tike_grad_ifft2_pass1 both stream the hand-off:
// the factor g of rows {k1 + 16 k2} needs |F_s|^2 of ALL modes of exactly those
// rows, and the inverse's pass 1 for (tile, k1) needs exactly that g -- so one
// work item (position, k1) can do both: sweep A re-forms F_s row by row for
// the intensity (F discarded), g stays in 16 registers, sweep B re-reads the
// same 16 rows of every mode -- newest first, they are the likeliest to be
// cached still -- re-forms F_s, scales, and runs the inverse's pass 1 on the
// same registers.  The factor never goes through memory and the second read
// of the hand-off is served partly by the caches.  Gaussian / poisson without
// per-mode steps (those need the intensity between the two sweeps).  The path
// for S < TK_FG_RESIDENT_MIN_MODES; more modes: the resident kernel below.
// measured (tools/fg_probe.py, 8000 tiles): S = 3 1.91 vs 2.22 ms, 4 1.87 / 1.89, 5 1.86 / 1.91,
// 6 1.89 / 1.77, 7 1.86 / 1.78, 8 1.97 / 1.67 (two sweeps / resident)
#define TK_FG_RESIDENT_MIN_MODES 6

//
// BACK (the last slice of a multislice object, rpie.py:444-472): the wave the
// slices in front receive is FresnelSpectProp.adj applied b times to chi =
// IFFT2(G), i.e. IFFT2(conj(H) FFT2(IFFT2(G))) = IFFT2(conj(H)^b G) -- the
// forward transform of the step back cancels against the inverse that formed
// chi (probe window = detector).  G is in registers here, so the kernel also
// emits the inverse's pass 1 of conj(H)^b G, b = 1 .. nback, into work + b *
// back_stride: the steps back cost one more store each instead of a stored
// chi, a forward pass 1 and a column pass.
template <int MODEL, class DT, bool BACK = false, bool MK = true>
__global__ __launch_bounds__(256, BACK ? 2 : 3) void fwd_grad_ifft2_pass1_kernel(
    const cf* __restrict__ colin, const DT* __restrict__ data,
    const unsigned char* __restrict__ mask, const TkCostSink costs, cf* __restrict__ work,
    long nscan, int S, float fwd_scale, float unmeasured_scaling, float inv_nmeasured,
    const cf* __restrict__ twtab, const cf* __restrict__ backprop = nullptr, int nback = 0,
    long back_stride = 0) {
  constexpr int N = 256;
  using G2 = Fft2Geom<N>;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  __shared__ float red[4];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int t = threadIdx.x;
  const float s2 = fwd_scale * fwd_scale;
  for (long v = blockIdx.x; v < nscan * 16; v += gridDim.x) {
    const int k1 = (int)(v & 15);
    const long n = nscan - 1 - (v >> 4);  // descending: see fwd_gradient_scale_kernel
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    // ---- sweep A: intensity of rows k1 + 16 k2, all modes (pipelined loads)
    float I[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) I[k2] = 0.f;
    cf un[16];
    {
      const cf* __restrict__ src0 = colin + (n * S) * (long)N * N + k1 * N + t;
#pragma unroll
      for (int r = 0; r < 16; ++r) un[r] = tk_ld_stream(src0 + (long)(16 * r) * N);
    }
    for (int s = 0; s < S; ++s) {
      cf u[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) u[r] = un[r];
      // next: mode s + 1 of sweep A, or the first mode of sweep B (S - 1 again)
      {
        const int sn = s + 1 < S ? s + 1 : S - 1;
        const cf* __restrict__ src = colin + (n * S + sn) * (long)N * N + k1 * N + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) un[r] = src[(long)(16 * r) * N];
      }
      Dft<16, false>::run(u);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) I[k2] += norm2(u[k2]) * s2;
    }
    // ---- the factor (times the forward scale the inverse applies to F) and the cost
    float cost;
    {
      DT raw[16];
      unsigned bits;
      tk_request_data16(data, !MK ? (const unsigned char*)nullptr : mask, n, k1, t, raw, bits);
      cost = tk_gradient_factor16<MODEL>(I, raw, bits, unmeasured_scaling, fwd_scale);
    }
    if (costs.costs) {
      cost = tk_block_sum256(cost, red);
      if (threadIdx.x == 0) tk_cost_add(costs, n, k1, cost * inv_nmeasured);
    }
    // ---- sweep B: modes S - 1 .. 0, gradient and the inverse's pass 1
    for (int s = S - 1; s >= 0; --s) {
      cf u[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) u[r] = un[r];
      if (s > 0) {
        const cf* __restrict__ src = colin + (n * S + s - 1) * (long)N * N + k1 * N + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) un[r] = src[(long)(16 * r) * N];
      }
      Dft<16, false>::run(u);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) u[k2] = u[k2] * I[k2];
      cf g[BACK ? 16 : 1];
      if (BACK) {
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) g[BACK ? k2 : 0] = u[k2];
      }
      Dft<16, true>::run(u);
#pragma unroll
      for (int ya = 1; ya < 16; ++ya) u[ya] = mul_tw<true>(u[ya], twtab[N + k1 * ya]);
      cf* mid = work + (n * S + s) * (long)N * N;
      fft2_rows_from_columns<N, true, true>(lds, tw, line, j, u, mid + (long)(16 * k1) * N);
      if (BACK) {
        for (int b = 1; b <= nback; ++b) {
#pragma unroll
          for (int k2 = 0; k2 < 16; ++k2) {
            g[BACK ? k2 : 0] = g[BACK ? k2 : 0] * conjf(*tk_at_pinned(backprop + (k1 + 16 * k2) * N, (unsigned)t * 8u));
            u[k2] = g[BACK ? k2 : 0];
          }
          Dft<16, true>::run(u);
#pragma unroll
          for (int ya = 1; ya < 16; ++ya) u[ya] = mul_tw<true>(u[ya], twtab[N + k1 * ya]);
          fft2_rows_from_columns<N, true, true>(lds, tw, line, j, u,
                                                mid + b * back_stride + (long)(16 * k1) * N);
        }
      }
    }
  }
}

// ---- one mode: the column-pass values of the work item are 16 registers per
// thread, so there is no second sweep at all (cgrad's gradient pass, S = 1)
template <int MODEL, class DT, bool MK = true>
__global__ __launch_bounds__(256, 4) void fwd_grad_ifft2_pass1_single_kernel(
    const cf* __restrict__ colin, const DT* __restrict__ data,
    const unsigned char* __restrict__ mask, const TkCostSink costs, cf* __restrict__ work,
    long nscan, float fwd_scale, float unmeasured_scaling, float inv_nmeasured,
    const cf* __restrict__ twtab) {
  constexpr int N = 256;
  using G2 = Fft2Geom<N>;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  __shared__ float red[4];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int t = threadIdx.x;
  const float s2 = fwd_scale * fwd_scale;
  for (long v = blockIdx.x; v < nscan * 16; v += gridDim.x) {
    const int k1 = (int)(v & 15);
    const long n = nscan - 1 - (v >> 4);  // descending: see fwd_gradient_scale_kernel
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    cf u[16];
    {
      const cf* __restrict__ src = colin + n * (long)N * N + k1 * N;  // uniform
#pragma unroll
      for (int r = 0; r < 16; ++r) u[r] = tk_ld_stream(tk_at_pinned(src + (16 * r) * N, t * 8u));
    }
    DT raw[16];
    unsigned bits;
    tk_request_data16(data, !MK ? (const unsigned char*)nullptr : mask, n, k1, t, raw, bits);
    Dft<16, false>::run(u);
    float I[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) I[k2] = norm2(u[k2]) * s2;
    float cost = tk_gradient_factor16<MODEL>(I, raw, bits, unmeasured_scaling, fwd_scale);
    if (costs.costs) {
      cost = tk_block_sum256(cost, red);
      if (threadIdx.x == 0) tk_cost_add(costs, n, k1, cost * inv_nmeasured);
    }
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) u[k2] = u[k2] * I[k2];
    Dft<16, true>::run(u);
#pragma unroll
    for (int ya = 1; ya < 16; ++ya) u[ya] = mul_tw<true>(u[ya], twtab[N + k1 * ya]);
    cf* mid = work + n * (long)N * N;
    fft2_rows_from_columns<N, true, true>(lds, tw, line, j, u, mid + (long)(16 * k1) * N);
  }
}

// ---- the same with the column-pass values RESIDENT IN REGISTERS (256^2)
// The kernel above streams the hand-off twice, and its second sweep misses L2
// (96 work items x 256 KiB per XCD).  Here a 512-thread workgroup -- one per
// CU, 2 waves/SIMD, the whole register file -- owns a work item (position,
// k1): half h of the workgroup holds F of modes [h*MH, h*MH + MH) of its
// column, 32 registers per mode; the halves exchange their partial
// intensities through LDS, form the same g, and each sends its modes through
// the inverse's pass 1 in its own LDS transpose region.  The hand-off is read
// ONCE.  With a single workgroup per CU nothing else hides the memory
// latency, so the loop is rotated: as soon as mode m of this work item has
// left its registers, the rows of mode m of the NEXT work item are requested
// into them -- a full work item (256 KiB per CU) is always in flight.
template <int N, bool INV, class Tw>
__device__ __forceinline__ void fft2_rows_from_columns_half(cf* __restrict__ lds, const Tw& tw,
                                                            int t, int line, int j, cf (&a)[16],
                                                            cf* __restrict__ rows, bool store) {
  using G2 = Fft2Geom<N>;
#pragma unroll
  for (int ya = 0; ya < 16; ++ya) lds[ya * G2::LS + tk_pad16(t)] = a[ya];
  __syncthreads();
  cf v[16];
  cf* lbase = lds + line * G2::LS;
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = lbase[tk_pad16(j + i * G2::T)];
  FftStageWave<N, INV, 0>::run(v, lbase, j, tw);
  if (store) {
    const unsigned lo = (unsigned)(line * N + j) * 8u;  // `rows` is uniform
#pragma unroll
    for (int i = 0; i < 16; ++i) tk_st_stream(tk_at(rows + i * G2::T, lo), v[i]);
  }
  __syncthreads();
}

// STEPS (poisson model, every pixel measured; see
// poisson_sweep2_grad_ifft2_pass1_kernel): with F of all modes in registers the
// sweeps of the per-mode step lengths (exitwave.py:122-184) cost no re-read --
//   1: the FIRST sweep alone: denominators and numerators at alpha = start, the
//      costs; nothing is transformed back or written;
//   2: the SECOND sweep's numerators at alpha[n][s], then the gradient pass as
//      usual (pass 1 of the inverse WITHOUT the step length).
// sums (nscan, S, 2) = { denominator, numerator }, one atomic per wave.
// MK (STEPS only): a mask may be given; without it the selects on `measured`
// and the mask loads are compiled out (2.87 against 3.04 ms per 1000 positions
// for both sweeps).
template <int MH, int MODEL, class DT, int STEPS = 0, bool MK = true>
__global__ __launch_bounds__(512, 1) void fwd_grad_ifft2_pass1_resident_kernel(
    const cf* __restrict__ colin, const DT* __restrict__ data,
    const unsigned char* __restrict__ mask, const TkCostSink costs, cf* __restrict__ work,
    long nscan, int S, float fwd_scale, float unmeasured_scaling, float inv_nmeasured,
    const cf* __restrict__ twtab, const float* __restrict__ alpha = nullptr, float start = 0.f,
    float* __restrict__ sums = nullptr) {
  constexpr int N = 256;
  using G2 = Fft2Geom<N>;
  __shared__ cf lds[2 * G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  // (STEPS: the counts of a work item wait in LDS, a private slot per thread
  // and pixel, while the modes go through their registers)
  __shared__ float dvp[STEPS != 0 ? 16 * 512 : 1];
  __shared__ float ivp[STEPS == 2 ? 16 * 512 : 1];  // ... and, beside the inverse, the intensity
  cf* twl = lds + 2 * G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int h = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const int t = threadIdx.x & 255;
  cf* const mylds = lds + h * G2::LDS_ELEMS;
  int line = t / G2::T, j = t % G2::T;
  asm volatile("" : "+v"(line), "+v"(j));
  const FftTwLds<N> tw{twl, j};
  const int m0 = h * MH;
  const float s2 = fwd_scale * fwd_scale;
  const long total = nscan * 16;
  cf F[MH][16];
  auto request = [&](long v, int m) {  // rows 16 r + k1 of mode m0 + m into F[m]
    const int k1 = (int)(v & 15);
    const long n = nscan - 1 - (v >> 4);
    if (m0 + m < S) {
      const cf* __restrict__ src = colin + (n * S + m0 + m) * (long)N * N + k1 * N;  // uniform
#pragma unroll
      for (int r = 0; r < 16; ++r)
        F[m][r] = tk_ld_stream(tk_at_pinned(src + (16 * r) * N, t * 8u));
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) F[m][r] = cf{0.f, 0.f};
    }
  };
  long v = blockIdx.x;
  if (v < total) {
#pragma unroll
    for (int m = 0; m < MH; ++m) request(v, m);
  }
  for (; v < total; v += gridDim.x) {
    const int k1 = (int)(v & 15);
    const long n = nscan - 1 - (v >> 4);  // descending: see fwd_gradient_scale_kernel
    DT raw[16];
    unsigned bits;
    float I[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) I[k2] = 0.f;
#pragma unroll
    for (int m = 0; m < MH; ++m) {
      // the counts: requested behind the last hand-off rows, used after the
      // last butterfly and the exchange
      if (m == MH - 1)
        tk_request_data16(data, !MK ? (const unsigned char*)nullptr : mask, n, k1, t, raw, bits);
      Dft<16, false>::run(F[m]);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) I[k2] += norm2(F[m][k2]) * s2;
    }
    {
      float* const myI = reinterpret_cast<float*>(mylds);
      const float* const otherI =
          reinterpret_cast<const float*>(lds + (1 - h) * G2::LDS_ELEMS);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) myI[k2 * N + t] = I[k2];
      __syncthreads();
      // both halves add in the same order: they must form the SAME factor
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        const float o = otherI[k2 * N + t];
        I[k2] = h == 0 ? I[k2] + o : o + I[k2];
      }
      __syncthreads();
    }
    float cost;
    if (STEPS == 0) {
      cost = tk_gradient_factor16<MODEL>(I, raw, bits, unmeasured_scaling, fwd_scale);
    } else {
      // (I stays the intensity: the sweeps need it next to every mode; the
      // factor -xi x scale is formed per mode below)
      cost = 0.f;
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        // an unmeasured pixel (its count may be NaN: selected, never used in
        // arithmetic) is parked as -1: no term in any sum, factor 0
        // (unmeasured_pixels_scaling = 1, the only value this path serves)
        const bool meas = !MK || ((bits >> k2) & 1u);
        const float dv = meas ? (float)raw[k2] : -1.0f;
        // (the costs come out of the FIRST sweep's launch: sixteen logarithms
        // beside 128 registers of F are what the second one spilled for)
        if (STEPS == 1) cost += meas ? I[k2] - dv * logf(I[k2] + 1e-9f) : 0.f;
        dvp[k2 * 512 + threadIdx.x] = dv;
        if (STEPS == 2) ivp[k2 * 512 + threadIdx.x] = I[k2];
      }
    }
    if (STEPS != 2 && costs.costs && h == 0) {
      cost = tk_wave_sum(cost);
      if ((threadIdx.x & 63) == 0)
        tk_cost_add(costs, n, k1 * 4 + (int)(threadIdx.x >> 6), cost * inv_nmeasured);
    }
    const long vn = v + gridDim.x;
#pragma unroll
    for (int m = 0; m < MH; ++m) {
      if (STEPS != 0) {
        // the sweep's sums of this mode over this thread's 16 pixels
        const float al = (STEPS == 1 || m0 + m >= S) ? start : alpha[n * S + m0 + m];  // uniform
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
          // (in quarters: all sixteen pairs of parked values in flight at once
          // do not fit next to the other modes)
          if (STEPS == 2 && (k2 & 3) == 0) asm volatile("" ::: "memory");
          const float dv = dvp[k2 * 512 + threadIdx.x];
          const float ie = STEPS == 2 ? ivp[k2 * 512 + threadIdx.x] : I[k2];
          // (v_rcp_f32, 1 ulp: an IEEE division is a dozen instructions and
          // five temporaries, twice per pixel and mode, next to 128 registers
          // of F)
          const bool meas = !MK || dv >= 0.f;
          const float xi = 1.0f - dv * __builtin_amdgcn_rcpf(ie + 1e-9f);
          // (|F|^2 formed AGAIN here: kept from the intensity loop -- the same
          // expression -- sixteen values per mode lived across the exchange,
          // in scratch: 108-140 bytes per lane until round 6)
          if (STEPS == 2) asm volatile("" : "+v"(F[m][k2].x), "+v"(F[m][k2].y));
          const float av = norm2(F[m][k2]) * s2;
          const float xam1 = xi * al - 1.0f;
          const float tn =
              xi * av * (1.0f + dv * xam1 * __builtin_amdgcn_rcpf(av * xam1 * xam1 + ie - av));
          num += meas ? tn : 0.f;
          if (STEPS == 1) den += meas ? xi * xi * av : 0.f;
          if (STEPS == 2) F[m][k2] = F[m][k2] * (meas ? -xi * fwd_scale : 0.f);
        }
        if (m0 + m < S) {  // uniform
          num = tk_wave_sum(num);
          if (STEPS == 1) den = tk_wave_sum(den);
          if ((threadIdx.x & 63) == 0) {
            unsafeAtomicAdd(&sums[(n * S + m0 + m) * 2 + 1], num);
            if (STEPS == 1) unsafeAtomicAdd(&sums[(n * S + m0 + m) * 2], den);
          }
        }
      } else {
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) F[m][k2] = F[m][k2] * I[k2];
      }
      if (STEPS != 1) {
        Dft<16, true>::run(F[m]);
#pragma unroll
        for (int ya = 1; ya < 16; ++ya)
          F[m][ya] = mul_tw<true>(F[m][ya], twtab[N + k1 * ya]);
        cf* mid = work + (n * S + m0 + m) * (long)N * N;
        fft2_rows_from_columns_half<N, true>(mylds, tw, t, line, j, F[m],
                                             mid + (long)(16 * k1) * N, m0 + m < S);
      }
      if (vn < total) request(vn, m);
    }
  }
}

// The same at 512^2 (RB = 32): per tile and k1 a thread owns one column,
//   F[k1 + 16 k2] = radix-32 over r of rows 16 r + k1 of the hand-off, times g;
// the 32 rows it then holds are exactly TWO input groups of the inverse's pass
// 1 (rows of residue k1 and k1 + 16 mod 32: k2 even / odd), so they go through
// LDS into the row layout and through fft2_pass1 -- the result is the
// intermediate tike_ifft2_pass1_scaled would have produced from a stored far
// plane, which tike_ifft2_pass2_gradients finishes.
template <int MODE>
__global__ __launch_bounds__(512, 2) void grad_ifft2_pass1_512_kernel(
    const cf* __restrict__ colin, cf* __restrict__ work, long ntile, float fwd_scale,
    const cf* __restrict__ twtab, const float* __restrict__ gscale, int S,
    const float* __restrict__ mode_scale, const unsigned char* __restrict__ measured) {
  constexpr int N = 512;
  using G2 = Fft2Geom<N>;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int t = threadIdx.x;
  const long nscan = ntile / S;
  const long nvirt = ((nscan + 7) / 8) * 8 * S;
  for (long w = blockIdx.x; w < nvirt * 16; w += gridDim.x) {
    const long tile = tk_xcd_tile(w % nvirt, S, nscan);
    const int k1 = (int)(w / nvirt);
    if (tile < 0) continue;  // uniform
    const cf* __restrict__ src = colin + tile * (long)N * N;
    cf* __restrict__ mid = work + tile * (long)N * N;
    const float* __restrict__ gs = gscale + (tile / S) * (long)N * N;
    const float ms = MODE == 2 ? mode_scale[tile] : 1.0f;
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    cf u[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) u[r] = tk_ld_stream(src + (16 * r + k1) * N + t);
    Dft<32, false>::run(u);
    if constexpr (MODE == 2) {
      // (factors and mask bytes requested together, the step selected: see
      // grad_ifft2_crop_kernel)
      float g[32];
#pragma unroll
      for (int k2 = 0; k2 < 32; ++k2) g[k2] = gs[(k1 + 16 * k2) * N + t];
      if (measured != nullptr) {  // uniform
        unsigned char mb[32];
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) mb[k2] = measured[(k1 + 16 * k2) * N + t];
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) g[k2] *= mb[k2] ? ms : 1.0f;
      } else {
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) g[k2] *= ms;
      }
#pragma unroll
      for (int k2 = 0; k2 < 32; ++k2) u[k2] = u[k2] * (g[k2] * fwd_scale);
    } else {
#pragma unroll
      for (int k2 = 0; k2 < 32; ++k2) {
        const int p = (k1 + 16 * k2) * N + t;
        float g = gs[p] * fwd_scale;
        u[k2] = u[k2] * g;
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // rows k1 + 16 h + 32 y2 (column t) -> row layout of pass-1 group k1 + 16 h
#pragma unroll
      for (int y2 = 0; y2 < 16; ++y2) lds[y2 * G2::LS + tk_pad16(t)] = u[2 * y2 + h];
      __syncthreads();
      const cf* __restrict__ lrow = lds + line * G2::LS;
      fft2_pass1<N, true, true>(
          lds, twtab, tw, line, j, k1 + 16 * h,
          [&](int, int e, auto) { return lrow[tk_pad16(e)]; }, mid);
    }
  }
}

// ---- 512^2: column pass + gradient factor + inverse pass 1 in ONE launch.
// The resident form of 256^2 does not exist here -- F of four modes is 4 x 32
// values per column = 256 registers before anything else, and splitting the
// 512 columns over two half-workgroups (as the modes are split at 256^2) would
// break the inverse's row transforms, which need whole rows -- so this is the
// two-sweep form: sweep A re-forms F_s mode by mode for the intensity (F
// discarded, the factor stays in 32 registers), sweep B re-reads the 32 rows
// of every mode, newest first -- 512 KiB per work item, in flight 128 MiB over
// the chip: the second read is the Infinity Cache's, not HBM's -- applies the
// factor and sends the two 16-row groups it holds (residues k1 and k1 + 16 mod
// 32) through the inverse's pass 1.  Against the two launches it replaces
// (tike_fwd_gradient_scale + tike_grad_ifft2_pass1) the factor never goes
// through memory and HBM sees the hand-off once.

template <int MODEL, class DT, bool MK = true>
__global__ __launch_bounds__(512, 2) void fwd_grad_ifft2_pass1_512_kernel(
    const cf* __restrict__ colin, const DT* __restrict__ data,
    const unsigned char* __restrict__ mask, const TkCostSink costs, cf* __restrict__ work,
    long nscan, int S, float fwd_scale, float unmeasured_scaling, float inv_nmeasured,
    const cf* __restrict__ twtab) {
  constexpr int N = 512;
  using G2 = Fft2Geom<N>;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  __shared__ float red[8];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int t = threadIdx.x;
  const float s2 = fwd_scale * fwd_scale;
  for (long v = blockIdx.x; v < nscan * 16; v += gridDim.x) {
    const int k1 = (int)(v & 15);
    const long n = nscan - 1 - (v >> 4);  // descending: see fwd_gradient_scale_kernel
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    // ---- sweep A: intensity of rows k1 + 16 k2 over the modes
    float I[32];
#pragma unroll
    for (int k2 = 0; k2 < 32; ++k2) I[k2] = 0.f;
    for (int s = 0; s < S; ++s) {
      const cf* __restrict__ src = colin + (n * S + s) * (long)N * N + k1 * N + t;
      cf u[32];
#pragma unroll
      for (int r = 0; r < 32; ++r) u[r] = src[(long)(16 * r) * N];
      Dft<32, false>::run(u);
#pragma unroll
      for (int k2 = 0; k2 < 32; ++k2) I[k2] += norm2(u[k2]) * s2;
    }
    // ---- the factor (times the forward scale the inverse applies to F), the cost
    float cost;
    {
      DT raw[32];
      unsigned bits;
      tk_request_data<N, 32>(data, !MK ? (const unsigned char*)nullptr : mask, n, k1, t, raw, bits);
      cost = tk_gradient_factor<MODEL, 32>(I, raw, bits, unmeasured_scaling, fwd_scale);
    }
    if (costs.costs) {
      cost = tk_block_sum512(cost, red);
      if (threadIdx.x == 0) tk_cost_add(costs, n, k1, cost * inv_nmeasured);
    }
    // ---- sweep B: modes S - 1 .. 0 (the likeliest to be cached still first)
    for (int s = S - 1; s >= 0; --s) {
      const cf* __restrict__ src = colin + (n * S + s) * (long)N * N + k1 * N + t;
      cf* __restrict__ mid = work + (n * S + s) * (long)N * N;
      cf u[32];
#pragma unroll
      for (int r = 0; r < 32; ++r) u[r] = tk_ld_stream(src + (long)(16 * r) * N);
      Dft<32, false>::run(u);
#pragma unroll
      for (int k2 = 0; k2 < 32; ++k2) u[k2] = u[k2] * I[k2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // rows k1 + 16 h + 32 y2 (column t) -> row layout of pass-1 group k1 + 16 h
#pragma unroll
        for (int y2 = 0; y2 < 16; ++y2) lds[y2 * G2::LS + tk_pad16(t)] = u[2 * y2 + h];
        __syncthreads();
        const cf* __restrict__ lrow = lds + line * G2::LS;
        fft2_pass1<N, true, true>(
            lds, twtab, tw, line, j, k1 + 16 * h,
            [&](int, int e, auto) { return lrow[tk_pad16(e)]; }, mid);
      }
    }
  }
}

static int launch_grad_ifft2(const void* colin, const float* gscale, const float* mode_scale,
                             const unsigned char* measured, int S, void* work, void* chi,
                             long ntile, int pw, float fwd_scale, float inv_scale,
                             hipStream_t stream, bool pass2) {
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
  constexpr int N = 256;
  // a multiple of 8 workgroups keeps "virtual block % 8" equal to the XCD of
  // the workgroup across the grid-stride loop
  const long nvirt = ((ntile / S + 7) / 8) * 8 * S;
  // without pass 2 the 16 values of k1 of a tile are independent: split them
  // when there are fewer tiles than the chip holds workgroups
  int ksplit = 1;
  if (!pass2)
    while (ksplit < 16 && nvirt * ksplit < 2048) ksplit *= 2;
  const int grid8 = (tk_grid(nvirt * ksplit, 4) + 7) / 8 * 8;
#define TK_GINV(MODE, P2)                                                                      \
  hipLaunchKernelGGL((grad_ifft2_crop_kernel<N, MODE, P2>), dim3(grid8), dim3(N), 0,            \
                     stream, (const cf*)colin, (cf*)work, (cf*)chi, ntile, pw, fwd_scale,      \
                     inv_scale, tw, gscale, S, mode_scale, measured, ksplit)
  if (mode_scale && pass2)
    TK_GINV(2, true);
  else if (mode_scale)
    TK_GINV(2, false);
  else if (pass2)
    TK_GINV(1, true);
  else
    TK_GINV(1, false);
#undef TK_GINV
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_grad_ifft2_crop(const void* colin, const float* gscale,
                                    const float* mode_scale, const unsigned char* measured,
                                    int S, void* work, void* chi, long ntile, int det, int pw,
                                    float fwd_scale, float inv_scale, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(ntile >= 0 && S >= 1 && pw >= 1 && det >= pw);
  if (ntile == 0) return TK_OK;
  TK_CHECK_ARG(colin && gscale && work && chi && work != colin && ntile % S == 0);
  TK_CHECK_ARG(!(chi == work && pw != det));
  if (det != 256) return TK_ERR_UNSUPPORTED;
  return launch_grad_ifft2(colin, gscale, mode_scale, measured, S, work, chi, ntile, pw,
                           fwd_scale, inv_scale, stream, true);
}

// Pass 1 only of tike_grad_ifft2_crop: `work` receives the input of the inverse
// column pass (rows 16 k1 + ya), consumed by tike_ifft2_pass2_gradients.
// scratch: from tike_fwd_pass1.  costs (may be NULL) must not need zeroing by
// the caller (done here); work must not alias scratch.  det = 256 or 512.
static int launch_fwd_grad_ifft2_pass1(const void* scratch, const void* data, int data_u16,
                                       const unsigned char* measured, float* costs, void* work,
                                       int nscan, int S, int det, float fwd_scale, int model,
                                       float unmeasured_scaling, long num_measured,
                                       const cf* backprop, int nback, hipStream_t stream) {
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && det >= 1 && (model == 0 || model == 1) &&
               num_measured > 0 && nback >= 0);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(scratch && data && work && work != scratch && (nback == 0 || backprop));
  if (det != 256 && (det != 512 || nback > 0)) return TK_ERR_UNSUPPORTED;
  const long back_stride = (long)nscan * S * det * det;
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
  if (det == 512) {
    TkCostSink sink512;
    int rc = tk_cost_sink(costs, nscan, 16, stream, &sink512);
    if (rc) return rc;
    const float inv512 = 1.0f / (float)num_measured;
    const dim3 grid(tk_grid((long)nscan * 16, 2)), block(512);
#define TK_FG512_K(M, DT, MK_)                                                                \
  hipLaunchKernelGGL((fwd_grad_ifft2_pass1_512_kernel<M, DT, MK_>), grid, block, 0, stream,   \
                     (const cf*)scratch, (const DT*)data, measured, sink512, (cf*)work,       \
                     (long)nscan, S, fwd_scale, unmeasured_scaling, inv512, tw)
#define TK_FG512(M, DT)           \
  do {                            \
    if (measured != nullptr)      \
      TK_FG512_K(M, DT, true);    \
    else                          \
      TK_FG512_K(M, DT, false);   \
  } while (0)
    if (model == 0 && data_u16)
      TK_FG512(0, unsigned short);
    else if (model == 0)
      TK_FG512(0, float);
    else if (data_u16)
      TK_FG512(1, unsigned short);
    else
      TK_FG512(1, float);
#undef TK_FG512
#undef TK_FG512_K
    TK_LAUNCH_CHECK();
    return tk_cost_finish(sink512, nscan, stream);
  }
  // (nback > 0: the two-sweep kernel for any number of modes -- a BACK form of
  // the resident kernel, G of the mode in hand parked in LDS, was 10 % slower:
  // 3.13 vs 2.81 ms per 1000 positions x 8 modes x 2 slices)
  const bool resident = S >= TK_FG_RESIDENT_MIN_MODES && S <= 8 && nback == 0;
  // contributors per pattern: (k1, wave of the first half) / (k1)
  TkCostSink sink;
  {
    int rc = tk_cost_sink(costs, nscan, resident ? 64 : 16, stream, &sink);
    if (rc) return rc;
  }
  const float inv = 1.0f / (float)num_measured;
  if (resident) {
    // one 512-thread workgroup per CU (it takes the whole register file)
    const dim3 grid(tk_grid((long)nscan * 16, 1)), block(512);
#define TK_FGR_K(MH, M, DT, MK_)                                                              \
  hipLaunchKernelGGL((fwd_grad_ifft2_pass1_resident_kernel<MH, M, DT, 0, MK_>), grid, block,  \
                     0, stream, (const cf*)scratch, (const DT*)data, measured, sink,          \
                     (cf*)work, (long)nscan, S, fwd_scale, unmeasured_scaling, inv, tw)
#define TK_FGR(MH, M, DT)          \
  do {                             \
    if (measured != nullptr)       \
      TK_FGR_K(MH, M, DT, true);   \
    else                           \
      TK_FGR_K(MH, M, DT, false);  \
  } while (0)
#define TK_FGR_M(MH)                                                                          \
  do {                                                                                        \
    if (model == 0 && data_u16)                                                               \
      TK_FGR(MH, 0, unsigned short);                                                          \
    else if (model == 0)                                                                      \
      TK_FGR(MH, 0, float);                                                                   \
    else if (data_u16)                                                                        \
      TK_FGR(MH, 1, unsigned short);                                                          \
    else                                                                                      \
      TK_FGR(MH, 1, float);                                                                   \
  } while (0)
    if (S == 6)
      TK_FGR_M(3);
    else
      TK_FGR_M(4);
#undef TK_FGR_M
#undef TK_FGR
#undef TK_FGR_K
    TK_LAUNCH_CHECK();
    return tk_cost_finish(sink, nscan, stream);
  }
  const dim3 grid(tk_grid((long)nscan * 16, 12)), block(256);
  if (nback > 0) {  // (one mode too: the two-sweep kernel reads it twice)
#define TK_FGB(M, DT)                                                                         \
  hipLaunchKernelGGL((fwd_grad_ifft2_pass1_kernel<M, DT, true>), grid, block, 0, stream,      \
                     (const cf*)scratch, (const DT*)data, measured, sink, (cf*)work,         \
                     (long)nscan, S, fwd_scale, unmeasured_scaling, inv, tw, backprop, nback, \
                     back_stride)
    if (model == 0 && data_u16)
      TK_FGB(0, unsigned short);
    else if (model == 0)
      TK_FGB(0, float);
    else if (data_u16)
      TK_FGB(1, unsigned short);
    else
      TK_FGB(1, float);
#undef TK_FGB
    TK_LAUNCH_CHECK();
    return tk_cost_finish(sink, nscan, stream);
  }
  if (S == 1) {
#define TK_FG1_K(M, DT, MK_)                                                                  \
  hipLaunchKernelGGL((fwd_grad_ifft2_pass1_single_kernel<M, DT, MK_>), grid, block, 0, stream, \
                     (const cf*)scratch, (const DT*)data, measured, sink, (cf*)work,         \
                     (long)nscan, fwd_scale, unmeasured_scaling, inv, tw)
#define TK_FG1(M, DT)           \
  do {                          \
    if (measured != nullptr)    \
      TK_FG1_K(M, DT, true);    \
    else                        \
      TK_FG1_K(M, DT, false);   \
  } while (0)
    if (model == 0 && data_u16)
      TK_FG1(0, unsigned short);
    else if (model == 0)
      TK_FG1(0, float);
    else if (data_u16)
      TK_FG1(1, unsigned short);
    else
      TK_FG1(1, float);
#undef TK_FG1
#undef TK_FG1_K
    TK_LAUNCH_CHECK();
    return tk_cost_finish(sink, nscan, stream);
  }
#define TK_FG_K(M, DT, MK_)                                                                   \
  hipLaunchKernelGGL((fwd_grad_ifft2_pass1_kernel<M, DT, false, MK_>), grid, block, 0, stream, \
                     (const cf*)scratch, (const DT*)data, measured, sink, (cf*)work,         \
                     (long)nscan, S, fwd_scale, unmeasured_scaling, inv, tw)
#define TK_FG(M, DT)           \
  do {                         \
    if (measured != nullptr)   \
      TK_FG_K(M, DT, true);    \
    else                       \
      TK_FG_K(M, DT, false);   \
  } while (0)
  if (model == 0 && data_u16)
    TK_FG(0, unsigned short);
  else if (model == 0)
    TK_FG(0, float);
  else if (data_u16)
    TK_FG(1, unsigned short);
  else
    TK_FG(1, float);
#undef TK_FG
  TK_LAUNCH_CHECK();
  return tk_cost_finish(sink, nscan, stream);
}

extern "C" int tike_fwd_grad_ifft2_pass1(const void* scratch, const void* data, int data_u16,
                                         const unsigned char* measured, float* costs,
                                         void* work, int nscan, int S, int det, float fwd_scale,
                                         int model, float unmeasured_scaling, long num_measured,
                                         void* stream) {
  TK_ENTER();
  return launch_fwd_grad_ifft2_pass1(scratch, data, data_u16, measured, costs, work, nscan, S,
                                     det, fwd_scale, model, unmeasured_scaling, num_measured,
                                     nullptr, 0, (hipStream_t)stream);
}

// The last slice of a multislice object: work (nslices, nscan, S, det, det);
// work[b] = the inverse's pass 1 of conj(propagator)^b x (far-plane gradient)
// -- what tike_ifft2_pass2_products of slice nslices - 1 - b finishes.
extern "C" int tike_fwd_grad_ifft2_pass1_slices(const void* scratch, const void* data,
                                                int data_u16, const unsigned char* measured,
                                                float* costs, void* work, int nscan, int S,
                                                int det, float fwd_scale, int model,
                                                float unmeasured_scaling, long num_measured,
                                                const void* propagator, int nslices,
                                                void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nslices >= 1);
  return launch_fwd_grad_ifft2_pass1(scratch, data, data_u16, measured, costs, work, nscan, S,
                                     det, fwd_scale, model, unmeasured_scaling, num_measured,
                                     (const cf*)propagator, nslices - 1, (hipStream_t)stream);
}

extern "C" int tike_grad_ifft2_pass1(const void* colin, const float* gscale,
                                     const float* mode_scale, const unsigned char* measured,
                                     int S, void* work, long ntile, int det, float fwd_scale,
                                     void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(ntile >= 0 && S >= 1 && det >= 1);
  if (ntile == 0) return TK_OK;
  TK_CHECK_ARG(colin && gscale && work && work != colin && ntile % S == 0);
  if (det == 512) {
    const cf* tw = tk_twiddles();
    if (!tw) return (int)hipErrorNotInitialized;
    const long nvirt = ((ntile / S + 7) / 8) * 8 * S;
    const int grid8 = (tk_grid(nvirt * 16, 2) + 7) / 8 * 8;
    if (mode_scale)
      hipLaunchKernelGGL((grad_ifft2_pass1_512_kernel<2>), dim3(grid8), dim3(512), 0, stream,
                         (const cf*)colin, (cf*)work, ntile, fwd_scale, tw, gscale, S, mode_scale,
                         measured);
    else
      hipLaunchKernelGGL((grad_ifft2_pass1_512_kernel<1>), dim3(grid8), dim3(512), 0, stream,
                         (const cf*)colin, (cf*)work, ntile, fwd_scale, tw, gscale, S, mode_scale,
                         measured);
    TK_LAUNCH_CHECK();
    return TK_OK;
  }
  if (det != 256) return TK_ERR_UNSUPPORTED;
  return launch_grad_ifft2(colin, gscale, mode_scale, measured, S, work, work, ntile, det,
                           fwd_scale, 1.0f, stream, false);
}

// Pass 1 only of tike_ifft2_crop_scaled / _scaled_modes (stored far plane).
extern "C" int tike_ifft2_pass1_scaled(const void* farplane, const float* gscale,
                                       const float* mode_scale, const unsigned char* measured,
                                       int S, void* work, long ntile, int det, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(ntile >= 0 && S >= 1 && det >= 1);
  if (ntile == 0) return TK_OK;
  TK_CHECK_ARG(farplane && gscale && work && work != farplane && ntile % S == 0);
  const cf* far = (const cf*)farplane;
  cf* wk = (cf*)work;
  switch (det) {
    case 128:
      return launch_icrop_v2<128, false>(far, wk, wk, ntile, det, 1.0f, stream, gscale, S,
                                         mode_scale, measured);
    case 256:
      return launch_icrop_v2<256, false>(far, wk, wk, ntile, det, 1.0f, stream, gscale, S,
                                         mode_scale, measured);
    case 512:
      return launch_icrop_v2<512, false>(far, wk, wk, ntile, det, 1.0f, stream, gscale, S,
                                         mode_scale, measured);
    default:
      return TK_ERR_UNSUPPORTED;
  }
}

// Poisson variant (lstsq.py:454-489): the gradient factor of mode s at a
// measured pixel is gscale[n][p] * mode_scale[n][s]; unmeasured pixels keep
// gscale alone.  measured == NULL: every pixel is measured.
extern "C" int tike_ifft2_crop_scaled_modes(const void* farplane, const float* gscale,
                                            const float* mode_scale,
                                            const unsigned char* measured, int S, void* work,
                                            void* chi, long ntile, int det, int pw, float scale,
                                            void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(ntile >= 0 && S >= 1 && pw >= 1 && det >= pw);
  if (ntile == 0) return TK_OK;
  TK_CHECK_ARG(farplane && gscale && mode_scale && work && chi && work != farplane &&
               ntile % S == 0);
  TK_CHECK_ARG(!(chi == work && pw != det));
  switch (det) {
    case 128:
      return launch_icrop_v2<128>((const cf*)farplane, (cf*)work, (cf*)chi, ntile, pw, scale,
                                  stream, gscale, S, mode_scale, measured);
    case 256:
      return launch_icrop_v2<256>((const cf*)farplane, (cf*)work, (cf*)chi, ntile, pw, scale,
                                  stream, gscale, S, mode_scale, measured);
    case 512:
      return launch_icrop_v2<512>((const cf*)farplane, (cf*)work, (cf*)chi, ntile, pw, scale,
                                  stream, gscale, S, mode_scale, measured);
    default:
      return TK_ERR_UNSUPPORTED;
  }
}

// ---------------------------------------------------- poisson step lengths
// exitwave.py:122-234.  One workgroup per (position, mode) tile; three sweeps
// over the measured pixels (denominator, then two fixed-point updates of the
// step), each closed by a block reduction.  xi = 1 - d / (I + 1e-9).
//   all modes:     denom = sum xi^2 a,   a = |F_s|^2
//                  numer = sum xi a (1 + d (xi alpha - 1) / (a (xi alpha - 1)^2 + I - a))
//   dominant mode: denom = sum xi^2 I
//                  numer = sum xi (I - d / (1 - alpha xi))      (same for all modes)
//   alpha <- (1 - w) alpha + w numer / denom
template <bool DOMINANT>
__global__ __launch_bounds__(256) void poisson_steps_kernel(
    const cf* __restrict__ farplane, const float* __restrict__ intensity,
    const float* __restrict__ data, const unsigned char* __restrict__ mask,
    float* __restrict__ steps, int S, long npix, float start, float w) {
  __shared__ float red[4];
  const long tile = blockIdx.x;  // DOMINANT: position; else position * S + mode
  const long n = DOMINANT ? tile : tile / S;
  const cf* __restrict__ F = farplane + tile * npix;
  const float* __restrict__ I = intensity + n * npix;
  const float* __restrict__ d = data + n * npix;
  float denom = 0.f;
  for (long p = threadIdx.x; p < npix; p += blockDim.x) {
    if (mask && !mask[p]) continue;
    const float Ie = I[p];
    const float xi = 1.0f - d[p] / (Ie + 1e-9f);
    denom += xi * xi * (DOMINANT ? Ie : norm2(F[p]));
  }
  denom = tk_block_sum256(denom, red);
  float alpha = start;
  for (int it = 0; it < 2; ++it) {
    float numer = 0.f;
    for (long p = threadIdx.x; p < npix; p += blockDim.x) {
      if (mask && !mask[p]) continue;
      const float Ie = I[p], Im = d[p];
      const float xi = 1.0f - Im / (Ie + 1e-9f);
      if (DOMINANT) {
        numer += xi * (Ie - Im / (1.0f - alpha * xi));
      } else {
        const float a = norm2(F[p]);
        const float xam1 = xi * alpha - 1.0f;
        numer += xi * a * (1.0f + Im * xam1 / (a * xam1 * xam1 + Ie - a));
      }
    }
    numer = tk_block_sum256(numer, red);
    alpha = alpha * (1.0f - w) + (numer / denom) * w;
  }
  if (threadIdx.x == 0) {
    if (DOMINANT) {
      for (int s = 0; s < S; ++s) steps[n * S + s] = alpha;
    } else {
      steps[tile] = alpha;
    }
  }
}

// All modes of a position in ONE workgroup and TWO sweeps (S <= 8, even pixel
// count): the intensity and the counts of a pixel are read once for its S
// modes, and the first fixed-point update (alpha = start: known) shares its
// sweep with the denominator -- 9 MiB instead of 24 MiB per position at
// 256^2 x 8.  A thread takes U pairs of neighbouring pixels per trip (16-byte
// loads of the waves, 8-byte loads of intensity and counts), every operand
// requested before the first is used: one workgroup per position, nothing else
// hides the latency.  Unmeasured pixels (their counts may be NaN) are selected
// away, never multiplied.
template <int MAXS, int U>
__global__ __launch_bounds__(256) void poisson_steps_allmodes_kernel(
    const cf* __restrict__ farplane, const float* __restrict__ intensity,
    const float* __restrict__ data, const unsigned char* __restrict__ mask,
    float* __restrict__ steps, int S, long npix, float start, float w) {
  typedef float tk_v4 __attribute__((ext_vector_type(4)));
  typedef float tk_v2 __attribute__((ext_vector_type(2)));
  __shared__ float red[4];
  const long n = blockIdx.x;
  const cf* __restrict__ F = farplane + n * S * npix;
  const float* __restrict__ I = intensity + n * npix;
  const float* __restrict__ d = data + n * npix;
  const long npair = npix / 2;
  float denom[MAXS], numer[MAXS], alpha[MAXS];
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    denom[s] = numer[s] = 0.f;
    alpha[s] = start;
  }
  for (int sweep = 0; sweep < 2; ++sweep) {
    for (long q0 = threadIdx.x; q0 < npair; q0 += (long)U * 256) {
      tk_v2 Ie[U], Im[U];
      tk_v4 f[U][MAXS];
      bool meas[U][2];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const long q = q0 + 256L * j;
        const bool in = q < npair;
        const long p = 2 * (in ? q : q0);
        Ie[j] = *reinterpret_cast<const tk_v2*>(I + p);
        Im[j] = *reinterpret_cast<const tk_v2*>(d + p);
        meas[j][0] = in && (mask ? mask[p] != 0 : true);
        meas[j][1] = in && (mask ? mask[p + 1] != 0 : true);
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
          if (s < S) f[j][s] = *reinterpret_cast<const tk_v4*>(F + s * npix + p);
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float ie = h ? Ie[j].y : Ie[j].x, im = h ? Im[j].y : Im[j].x;
          const float xi = 1.0f - im / (ie + 1e-9f);
#pragma unroll
          for (int s = 0; s < MAXS; ++s) {
            if (s < S) {
              const float a = h ? f[j][s].z * f[j][s].z + f[j][s].w * f[j][s].w
                                : f[j][s].x * f[j][s].x + f[j][s].y * f[j][s].y;
              const float xam1 = xi * alpha[s] - 1.0f;
              const float t = xi * a * (1.0f + im * xam1 / (a * xam1 * xam1 + ie - a));
              numer[s] += meas[j][h] ? t : 0.f;
              if (sweep == 0) denom[s] += meas[j][h] ? xi * xi * a : 0.f;
            }
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      if (s < S) {  // uniform
        if (sweep == 0) denom[s] = tk_block_sum256(denom[s], red);
        const float nm = tk_block_sum256(numer[s], red);
        alpha[s] = alpha[s] * (1.0f - w) + (nm / denom[s]) * w;
        numer[s] = 0.f;
      }
    }
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int s = 0; s < MAXS; ++s)
      if (s < S) steps[n * S + s] = alpha[s];
  }
}

// ---- the same step lengths WITHOUT a stored far plane (256^2 / 512^2): the
// column pass of fwd_gradient_scale_kernel with |F_s|^2 of all S modes kept in
// registers (S x RB floats), so that one read of the forward hand-off gives
//   FIRST: the poisson gradient factor and the costs (what
//          fwd_gradient_scale_kernel<N, 1, DT> stores) AND the first sweep of
//          exitwave.py:122-184 (denominator; numerator at alpha = start);
//   else : the second sweep (numerator at the alpha of the first).
// sums (nscan, S, 2) = { denominator, numerator } accumulate by atomics (one per
// wave, mode and sum); poisson_alpha_kernel turns them into alpha between and
// after the sweeps.  The far plane itself is never written: the inverse that
// follows (tike_grad_ifft2_pass1) re-forms it from the same hand-off.
template <int N, class DT, bool FIRST>
__global__ __launch_bounds__(256, 2) void poisson_colpass_kernel(
    const cf* __restrict__ colin, const DT* __restrict__ data,
    const unsigned char* __restrict__ mask, float* __restrict__ gscale,
    float* __restrict__ costs, const float* __restrict__ alpha, float start,
    float* __restrict__ sums, long nitem, int S, float scale, float unmeasured_scaling,
    float inv_nmeasured) {
  constexpr int RB = N / 16, NH = N / 256, MAXS = N == 256 ? 8 : 4;
  __shared__ float red[4];
  __shared__ float wsum[4][2 * MAXS];
  const float s2 = scale * scale;
  for (long v = blockIdx.x; v < nitem; v += gridDim.x) {
    const int hb = (int)(v % NH);
    const int k1 = (int)((v / NH) & 15);
    const long n = nitem / (16 * NH) - 1 - v / (16 * NH);  // descending, as its siblings
    const int t = hb * 256 + threadIdx.x;
    float a[MAXS][RB], I[RB];
#pragma unroll
    for (int k2 = 0; k2 < RB; ++k2) I[k2] = 0.f;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      if (s < S) {  // uniform
        const cf* __restrict__ src = colin + (n * S + s) * (long)N * N + k1 * N + t;
        cf u[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) u[r] = tk_ld_stream(src + (long)(16 * r) * N);
        Dft<RB, false>::run(u);
#pragma unroll
        for (int k2 = 0; k2 < RB; ++k2) {
          a[s][k2] = norm2(u[k2]) * s2;
          I[k2] += a[s][k2];
        }
      }
    }
    DT raw[RB];
    unsigned bits;
    tk_request_data<N, RB>(data, mask, n, k1, t, raw, bits);
    float den[MAXS], num[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) den[s] = num[s] = 0.f;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      if (s < S) {
        const float al = FIRST ? start : alpha[n * S + s];  // uniform
#pragma unroll
        for (int k2 = 0; k2 < RB; ++k2) {
          const bool meas = (bits >> k2) & 1u;
          const float dv = (float)raw[k2];
          const float xi = 1.0f - dv / (I[k2] + 1e-9f);
          const float xam1 = xi * al - 1.0f;
          const float av = a[s][k2];
          const float tn = xi * av * (1.0f + dv * xam1 / (av * xam1 * xam1 + I[k2] - av));
          num[s] += meas ? tn : 0.f;
          if (FIRST) den[s] += meas ? xi * xi * av : 0.f;
        }
      }
    }
    if (FIRST) {
      float cost = tk_gradient_factor<1, RB>(I, raw, bits, unmeasured_scaling, 1.0f);
      if (gscale != nullptr) {
#pragma unroll
        for (int k2 = 0; k2 < RB; ++k2)
          gscale[n * (long)N * N + (long)(k1 + 16 * k2) * N + t] = I[k2];
      }
      if (costs) {
        cost = tk_block_sum256(cost, red);
        if (threadIdx.x == 0) unsafeAtomicAdd(&costs[n], cost * inv_nmeasured);
      }
    }
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      if (s < S) {
        num[s] = tk_wave_sum(num[s]);
        if (FIRST) den[s] = tk_wave_sum(den[s]);
      }
    }
    __syncthreads();  // the previous item's sums have been read
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int s = 0; s < MAXS; ++s) {
        wsum[threadIdx.x >> 6][2 * s] = den[s];
        wsum[threadIdx.x >> 6][2 * s + 1] = num[s];
      }
    }
    __syncthreads();
    const int q = threadIdx.x;  // q = 2 s + {0: denominator, 1: numerator}
    if (q < 2 * S && (FIRST || (q & 1)))
      unsafeAtomicAdd(&sums[n * 2 * S + q], wsum[0][q] + wsum[1][q] + wsum[2][q] + wsum[3][q]);
  }
}

// ---- every pixel measured, 256^2: the SECOND sweep and the gradient pass in
// one launch.  With no unmeasured pixels the far-plane gradient of mode s is
// alpha_s x (F_s x poisson factor) -- linear in the step length -- so pass 1 of
// the inverse can be written BEFORE alpha_s of the second sweep is known and
// the factor applied by pass 2 (tike_ifft2_pass2_gradients_scaled).  The
// structure is fwd_grad_ifft2_pass1_kernel's two sweeps: sweep A re-forms F_s
// of every mode for the intensity; sweep B re-reads the rows, newest first,
// re-forms F_s -- whose |F_s|^2 gives the mode's numerator of the second sweep
// at the alpha of the first (a first version held |F_s|^2 of all modes across
// sweep A for them: 256 VGPRs + scratch, 2.3 ms; this one 2.0) -- applies the
// factor and runs the inverse's pass 1.  Replaces
// poisson_colpass_kernel<.., false> + tike_grad_ifft2_pass1 (the factor table
// written and read, the hand-off read once more from HBM).
template <class DT>
__global__ __launch_bounds__(256, 3) void poisson_sweep2_grad_ifft2_pass1_kernel(
    const cf* __restrict__ colin, const DT* __restrict__ data,
    const unsigned char* __restrict__ mask, const float* __restrict__ alpha,
    float* __restrict__ sums, cf* __restrict__ work, long nscan, int S, float fwd_scale,
    float unmeasured_scaling, const cf* __restrict__ twtab) {
  constexpr int N = 256;
  using G2 = Fft2Geom<N>;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int t = threadIdx.x;
  const float s2 = fwd_scale * fwd_scale;
  for (long v = blockIdx.x; v < nscan * 16; v += gridDim.x) {
    const int k1 = (int)(v & 15);
    const long n = nscan - 1 - (v >> 4);  // descending: see fwd_gradient_scale_kernel
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    // ---- sweep A: the intensity of rows k1 + 16 k2 (F_s discarded)
    float I[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) I[k2] = 0.f;
    for (int s = 0; s < S; ++s) {
      const cf* __restrict__ src = colin + (n * S + s) * (long)N * N + k1 * N;  // uniform
      cf u[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) u[r] = *tk_at_pinned(src + (16 * r) * N, t * 8u);
      Dft<16, false>::run(u);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) I[k2] += norm2(u[k2]) * s2;
    }
    DT raw[16];
    unsigned bits;
    tk_request_data16(data, mask, n, k1, t, raw, bits);
    // xi = 1 - d / (I + eps); the gradient factor is -xi (x the forward scale).
    // (the counts are not kept: d = (1 - xi)(I + eps) where sweep B needs them;
    // an unmeasured pixel -- its count may be NaN: selected, never used -- has
    // xi = 0 here: no term in the sums, factor 0, what
    // unmeasured_pixels_scaling = 1 asks for)
    float xi[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2)
      xi[k2] = ((bits >> k2) & 1u) ? 1.0f - (float)raw[k2] / (I[k2] + 1e-9f) : 0.f;
    // ---- sweep B: modes S - 1 .. 0.  F_s re-formed: |F_s|^2 gives the mode's
    // numerator of the second sweep (exitwave.py:160-172, one atomic per wave),
    // F_s x factor goes through the inverse's pass 1 without its step length
    for (int s = S - 1; s >= 0; --s) {
      const cf* __restrict__ src = colin + (n * S + s) * (long)N * N + k1 * N;  // uniform
      cf u[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) u[r] = tk_ld_stream(tk_at_pinned(src + (16 * r) * N, t * 8u));
      Dft<16, false>::run(u);
      const float al = alpha[n * S + s];  // uniform
      float num = 0.f;
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        const float av = norm2(u[k2]) * s2;
        const float xam1 = xi[k2] * al - 1.0f;
        const float dv = (1.0f - xi[k2]) * (I[k2] + 1e-9f);
        const float tn = xi[k2] * av * (1.0f + dv * xam1 / (av * xam1 * xam1 + I[k2] - av));
        num += xi[k2] != 0.f ? tn : 0.f;  // (0 x NaN of a dark unmeasured pixel)
        u[k2] = u[k2] * (-xi[k2] * fwd_scale);
      }
      num = tk_wave_sum(num);
      if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(&sums[n * 2 * S + 2 * s + 1], num);
      Dft<16, true>::run(u);
#pragma unroll
      for (int ya = 1; ya < 16; ++ya) u[ya] = mul_tw<true>(u[ya], twtab[N + k1 * ya]);
      cf* mid = work + (n * S + s) * (long)N * N;
      fft2_rows_from_columns<N, true, true>(lds, tw, line, j, u, mid + (long)(16 * k1) * N);
    }
  }
}

// alpha <- (1 - w) alpha + w numerator / denominator per (position, mode); the
// numerator is cleared for the next sweep.  first: alpha = start on entry.
__global__ __launch_bounds__(256) void poisson_alpha_kernel(float* __restrict__ sums,
                                                            float* __restrict__ alpha, long ntile,
                                                            float start, float w, int first) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < ntile; i += gridDim.x * 256L) {
    const float prev = first ? start : alpha[i];
    alpha[i] = prev * (1.0f - w) + (sums[2 * i + 1] / sums[2 * i]) * w;
    sums[2 * i + 1] = 0.f;
  }
}

extern "C" int tike_poisson_steps_handoff(const void* scratch, const void* data, int data_u16,
                                          const unsigned char* measured, float* gscale,
                                          float* costs, float* steps, float* sums, int nscan,
                                          int S, int det, float scale, float unmeasured_scaling,
                                          long num_measured, float step_start, float weight,
                                          void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && num_measured > 0);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(scratch && data && gscale && steps && sums);
  if (!((det == 256 && S <= 8) || (det == 512 && S <= 4))) return TK_ERR_UNSUPPORTED;
  const long ntile = (long)nscan * S;
  hipError_t e = hipMemsetAsync(sums, 0, sizeof(float) * 2 * (size_t)ntile, stream);
  if (e == hipSuccess && costs) e = hipMemsetAsync(costs, 0, sizeof(float) * (size_t)nscan, stream);
  if (e != hipSuccess) return (int)e;
  const long nitem = (long)nscan * 16 * (det / 256);
  const float inv = 1.0f / (float)num_measured;
  const dim3 grid(tk_grid(nitem, 32)), block(256);
  const dim3 agrid(tk_grid((ntile + 255) / 256, 4));
#define TK_PC(N, DT, FIRST)                                                                     \
  hipLaunchKernelGGL((poisson_colpass_kernel<N, DT, FIRST>), grid, block, 0, stream,               \
                     (const cf*)scratch, (const DT*)data, measured, gscale, costs, steps,          \
                     step_start, sums, nitem, S, scale, unmeasured_scaling, inv)
#define TK_PC_N(FIRST)                        \
  do {                                        \
    if (det == 256 && data_u16)               \
      TK_PC(256, unsigned short, FIRST);      \
    else if (det == 256)                      \
      TK_PC(256, float, FIRST);               \
    else if (data_u16)                        \
      TK_PC(512, unsigned short, FIRST);      \
    else                                      \
      TK_PC(512, float, FIRST);               \
  } while (0)
  TK_PC_N(true);
  hipLaunchKernelGGL(poisson_alpha_kernel, agrid, dim3(256), 0, stream, sums, steps, ntile,
                     step_start, weight, 1);
  TK_PC_N(false);
  hipLaunchKernelGGL(poisson_alpha_kernel, agrid, dim3(256), 0, stream, sums, steps, ntile,
                     step_start, weight, 0);
#undef TK_PC_N
#undef TK_PC
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// Every pixel measured, det = 256: the step lengths AND pass 1 of the inverse
// of F_s x factor (WITHOUT the step lengths: tike_ifft2_pass2_gradients_scaled
// applies `steps`) -- sweep 1, alpha, sweep 2 + gradient pass, alpha.
extern "C" int tike_poisson_steps_grad_ifft2_pass1(const void* scratch, const void* data,
                                                   int data_u16, const unsigned char* measured,
                                                   float* costs, float* steps, float* sums,
                                                   void* work, int nscan, int S, int det,
                                                   float scale, float unmeasured_scaling,
                                                   long num_measured, float step_start,
                                                   float weight, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && num_measured > 0);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(scratch && data && steps && sums && work && work != scratch);
  // (unmeasured pixels keep F x (unmeasured_scaling - 1), which no step length
  // multiplies: linear in the steps only when that is zero)
  if (det != 256 || S > 8 || (measured != nullptr && unmeasured_scaling != 1.0f))
    return TK_ERR_UNSUPPORTED;
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
  const long ntile = (long)nscan * S;
  hipError_t e = hipMemsetAsync(sums, 0, sizeof(float) * 2 * (size_t)ntile, stream);
  if (e == hipSuccess && costs) e = hipMemsetAsync(costs, 0, sizeof(float) * (size_t)nscan, stream);
  if (e != hipSuccess) return (int)e;
  const long nitem = (long)nscan * 16;
  const float inv = 1.0f / (float)num_measured;
  const dim3 grid(tk_grid(nitem, 32)), block(256);
  const dim3 agrid(tk_grid((ntile + 255) / 256, 4));
  if (S >= TK_FG_RESIDENT_MIN_MODES) {
    // F of all modes in registers (fwd_grad_ifft2_pass1_resident_kernel): each
    // sweep reads the hand-off once
    TkCostSink sink;
    int rc = tk_cost_sink(costs, nscan, 64, stream, &sink);
    if (rc) return rc;
    const TkCostSink none = {nullptr, nullptr, 0};
    const dim3 rgrid(tk_grid(nitem, 1)), rblock(512);
#define TK_PR_K(MH, DT, ST, SINK, AL, MK_)                                                    \
  hipLaunchKernelGGL((fwd_grad_ifft2_pass1_resident_kernel<MH, 1, DT, ST, MK_>), rgrid, rblock, \
                     0, stream, (const cf*)scratch, (const DT*)data, measured, SINK,          \
                     (cf*)work, (long)nscan, S, scale, unmeasured_scaling, inv, tw, AL,       \
                     step_start, sums)
#define TK_PR(MH, DT, ST, SINK, AL)        \
  do {                                     \
    if (measured != nullptr)               \
      TK_PR_K(MH, DT, ST, SINK, AL, true); \
    else                                   \
      TK_PR_K(MH, DT, ST, SINK, AL, false);\
  } while (0)
#define TK_PR_S(ST, SINK, AL)                      \
  do {                                             \
    if (S == 6 && data_u16)                        \
      TK_PR(3, unsigned short, ST, SINK, AL);      \
    else if (S == 6)                               \
      TK_PR(3, float, ST, SINK, AL);               \
    else if (data_u16)                             \
      TK_PR(4, unsigned short, ST, SINK, AL);      \
    else                                           \
      TK_PR(4, float, ST, SINK, AL);               \
  } while (0)
    TK_PR_S(1, sink, (const float*)nullptr);
    hipLaunchKernelGGL(poisson_alpha_kernel, agrid, dim3(256), 0, stream, sums, steps, ntile,
                       step_start, weight, 1);
    TK_PR_S(2, none, (const float*)steps);
    hipLaunchKernelGGL(poisson_alpha_kernel, agrid, dim3(256), 0, stream, sums, steps, ntile,
                       step_start, weight, 0);
#undef TK_PR_S
#undef TK_PR
#undef TK_PR_K
    TK_LAUNCH_CHECK();
    return tk_cost_finish(sink, nscan, stream);
  }
  if (data_u16)
    hipLaunchKernelGGL((poisson_colpass_kernel<256, unsigned short, true>), grid, block, 0, stream,
                       (const cf*)scratch, (const unsigned short*)data, measured,
                       (float*)nullptr, costs, steps, step_start, sums, nitem, S, scale,
                       unmeasured_scaling, inv);
  else
    hipLaunchKernelGGL((poisson_colpass_kernel<256, float, true>), grid, block, 0, stream,
                       (const cf*)scratch, (const float*)data, measured, (float*)nullptr, costs,
                       steps, step_start, sums, nitem, S, scale, unmeasured_scaling, inv);
  hipLaunchKernelGGL(poisson_alpha_kernel, agrid, dim3(256), 0, stream, sums, steps, ntile,
                     step_start, weight, 1);
  const dim3 ggrid(tk_grid(nitem, 8));
  if (data_u16)
    hipLaunchKernelGGL((poisson_sweep2_grad_ifft2_pass1_kernel<unsigned short>), ggrid, block, 0,
                       stream, (const cf*)scratch, (const unsigned short*)data, measured, steps,
                       sums, (cf*)work, (long)nscan, S, scale, unmeasured_scaling, tw);
  else
    hipLaunchKernelGGL((poisson_sweep2_grad_ifft2_pass1_kernel<float>), ggrid, block, 0, stream,
                       (const cf*)scratch, (const float*)data, measured, steps, sums, (cf*)work,
                       (long)nscan, S, scale, unmeasured_scaling, tw);
  hipLaunchKernelGGL(poisson_alpha_kernel, agrid, dim3(256), 0, stream, sums, steps, ntile,
                     step_start, weight, 0);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_poisson_steps(const void* farplane, const float* intensity,
                                  const float* data, const unsigned char* measured,
                                  float* steps, int nscan, int S, int det, float step_start,
                                  float weight, int dominant_mode, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && det >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(intensity && data && steps && (dominant_mode || farplane));
  const long npix = (long)det * det;
  if (dominant_mode)
    hipLaunchKernelGGL((poisson_steps_kernel<true>), dim3(nscan), dim3(256), 0, stream,
                       (const cf*)farplane, intensity, data, measured, steps, S, npix,
                       step_start, weight);
  else if (S <= 8 && npix % 2 == 0)
    hipLaunchKernelGGL((poisson_steps_allmodes_kernel<8, 2>), dim3(nscan), dim3(256), 0, stream,
                       (const cf*)farplane, intensity, data, measured, steps, S, npix,
                       step_start, weight);
  else
    hipLaunchKernelGGL((poisson_steps_kernel<false>), dim3((unsigned)nscan * S), dim3(256), 0,
                       stream, (const cf*)farplane, intensity, data, measured, steps, S, npix,
                       step_start, weight);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// farplane[n][s][p] *= mode_scale[n][s] on measured pixels (the generic-size
// poisson path applies it after tike_farplane_gradient).
__global__ __launch_bounds__(256) void scale_modes_kernel(cf* __restrict__ farplane,
                                                          const float* __restrict__ mode_scale,
                                                          const unsigned char* __restrict__ mask,
                                                          long npix) {
  const long tile = blockIdx.y;
  const float ms = mode_scale[tile];
  cf* __restrict__ F = farplane + tile * npix;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
       p += (long)gridDim.x * blockDim.x)
    if (!mask || mask[p]) F[p] = F[p] * ms;
}

extern "C" int tike_scale_modes(void* farplane, const float* mode_scale,
                                const unsigned char* measured, long ntile, int det,
                                void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(ntile >= 0 && det >= 1);
  if (ntile == 0) return TK_OK;
  TK_CHECK_ARG(farplane && mode_scale);
  const long npix = (long)det * det;
  const unsigned gx = (unsigned)((npix + 1023) / 1024);
  hipLaunchKernelGGL(scale_modes_kernel, dim3(gx, (unsigned)ntile), dim3(256), 0, stream,
                     (cf*)farplane, mode_scale, measured, npix);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------------ stand-alone objective ops
// The free functions tike.operators.{gaussian,poisson}{_each_pattern,_grad}
// (objective.py:18-124) and _intensity_from_farplane (ptycho.py:18-23).
__global__ __launch_bounds__(256) void intensity_kernel(const cf* __restrict__ F,
                                                        float* __restrict__ I, long nscan, int S,
                                                        long npix) {
  const long total = nscan * npix;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long n = i / npix, p = i % npix;
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += norm2(F[(n * S + s) * npix + p]);
    I[i] = a;
  }
}

template <int MODEL>
__global__ __launch_bounds__(256) void cost_each_kernel(const float* __restrict__ data,
                                                        const float* __restrict__ I,
                                                        float* __restrict__ costs, long nscan,
                                                        long npix) {
  __shared__ float red[4];
  for (long n = blockIdx.x; n < nscan; n += gridDim.x) {
    float c = 0.f;
    for (long p = threadIdx.x; p < npix; p += blockDim.x) {
      const float d = data[n * npix + p], iv = I[n * npix + p];
      if (MODEL == 0) {
        const float diff = sqrtf(iv) - sqrtf(d);
        c += diff * diff;
      } else {
        c += iv - d * logf(iv + 1e-9f);
      }
    }
    c = tk_block_sum256(c, red);
    if (threadIdx.x == 0) costs[n] = c / (float)npix;
  }
}

template <int MODEL>
__global__ __launch_bounds__(256) void objective_grad_kernel(const float* __restrict__ data,
                                                             const cf* __restrict__ F,
                                                             const float* __restrict__ I,
                                                             cf* __restrict__ out, long nscan,
                                                             int S, long npix) {
  const long total = nscan * npix;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long n = i / npix, p = i % npix;
    const float d = data[i], iv = I[i];
    const float g = MODEL == 0 ? 1.0f - sqrtf(d) / (sqrtf(iv) + 1e-9f) : 1.0f - d / (iv + 1e-9f);
    for (int s = 0; s < S; ++s) out[(n * S + s) * npix + p] = F[(n * S + s) * npix + p] * g;
  }
}

extern "C" int tike_intensity(const void* farplane, float* intensity, long nscan, int S,
                              long npix, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(farplane && intensity && nscan >= 0 && S >= 1 && npix >= 1);
  if (nscan == 0) return TK_OK;
  hipLaunchKernelGGL(intensity_kernel, dim3(tk_grid((nscan * npix + 255) / 256, 16)), dim3(256),
                     0, (hipStream_t)stream, (const cf*)farplane, intensity, nscan, S, npix);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_cost_each_pattern(const float* data, const float* intensity, float* costs,
                                      long nscan, long npix, int model, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(data && intensity && costs && nscan >= 0 && npix >= 1);
  TK_CHECK_ARG(model == 0 || model == 1);
  if (nscan == 0) return TK_OK;
  const dim3 grid(tk_grid(nscan, 16)), block(256);
  if (model == 0)
    hipLaunchKernelGGL((cost_each_kernel<0>), grid, block, 0, (hipStream_t)stream, data,
                       intensity, costs, nscan, npix);
  else
    hipLaunchKernelGGL((cost_each_kernel<1>), grid, block, 0, (hipStream_t)stream, data,
                       intensity, costs, nscan, npix);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_objective_grad(const float* data, const void* farplane,
                                   const float* intensity, void* out, long nscan, int S,
                                   long npix, int model, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(data && farplane && intensity && out && nscan >= 0 && S >= 1 && npix >= 1);
  TK_CHECK_ARG(model == 0 || model == 1);
  if (nscan == 0) return TK_OK;
  const dim3 grid(tk_grid((nscan * npix + 255) / 256, 16)), block(256);
  if (model == 0)
    hipLaunchKernelGGL((objective_grad_kernel<0>), grid, block, 0, (hipStream_t)stream, data,
                       (const cf*)farplane, intensity, (cf*)out, nscan, S, npix);
  else
    hipLaunchKernelGGL((objective_grad_kernel<1>), grid, block, 0, (hipStream_t)stream, data,
                       (const cf*)farplane, intensity, (cf*)out, nscan, S, npix);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// Patch gather / scatter-add and the unfused Convolution operator for gfx950.
//
// Replaces the reference's only CUDA source on the ptychography path,
// src/tike/operators/cupy/convolution.cu (fwd_patch :146-155, adj_patch
// :157-165, loop body :79-144), and the CuPy elementwise passes of
// convolution.py:58-154.  These are the general-shape kernels behind the
// Patch / Convolution operator API (any patch width, padding, nrepeat,
// broadcast K); the solver's hot loop uses the fused kernels in ptycho.hip.
#include "common.h"
#include "tike_amd.h"

// One workgroup iteration = one patch row (image ti, position ts, row py);
// threads run along px so image and patch accesses are contiguous.
template <bool ADJ>
__global__ __launch_bounds__(256) void patch_kernel(cf* __restrict__ images,
                                                    cf* __restrict__ patches,
                                                    const float* __restrict__ scan, int nimage,
                                                    int H, int W, int nscan, int nrepeat, int pw,
                                                    int padded, int npatch) {
  const int pad = (padded - pw) / 2;
  const long total = (long)nimage * H * W;
  const long nrow = (long)nimage * nscan * pw;
  float* __restrict__ imf = reinterpret_cast<float*>(images);
  for (long row = blockIdx.x; row < nrow; row += gridDim.x) {
    const int py = (int)(row % pw);
    const long ts_all = row / pw;  // ti * nscan + ts
    const int ts = (int)(ts_all % nscan);
    const int ti = (int)(ts_all / nscan);
    const TkCorner c = tk_corner(scan, ts_all);
    const int y = c.sy + py;
    if (y < 0 || y >= H) continue;
    // patches of image ti start at ti * npatch_per_image; forward has
    // npatch == nscan * nrepeat, adjoint may broadcast (convolution.cu:138).
    const long image_offset = (long)padded * padded * (ADJ ? (long)npatch : (long)nscan * nrepeat) * ti;
    for (int px = threadIdx.x; px < pw; px += blockDim.x) {
      const int x = c.sx + px;
      if (x < 0 || x >= W) continue;
      const long ii = ((long)ti * H + y) * W + x;
      const long pi = image_offset + (long)(pad + py) * padded + pad + px;
      if (!ADJ) {
        const cf v = tk_gather(images, ii, W, total, c);
        for (int r = 0; r < nrepeat; ++r)
          patches[pi + (long)padded * padded * (r + ((long)nrepeat * ts) % npatch)] = v;
      } else {
        for (int r = 0; r < nrepeat; ++r) {
          const cf v = patches[pi + (long)padded * padded * (r + ((long)nrepeat * ts) % npatch)];
          unsafeAtomicAdd(&imf[2 * ii], v.x * c.w00);
          unsafeAtomicAdd(&imf[2 * ii + 1], v.y * c.w00);
          if (c.w01 != 0.0f && ii + 1 < total) {
            unsafeAtomicAdd(&imf[2 * (ii + 1)], v.x * c.w01);
            unsafeAtomicAdd(&imf[2 * (ii + 1) + 1], v.y * c.w01);
          }
          if (c.w10 != 0.0f && ii + W < total) {
            unsafeAtomicAdd(&imf[2 * (ii + W)], v.x * c.w10);
            unsafeAtomicAdd(&imf[2 * (ii + W) + 1], v.y * c.w10);
          }
          if (c.w11 != 0.0f && ii + W + 1 < total) {
            unsafeAtomicAdd(&imf[2 * (ii + W + 1)], v.x * c.w11);
            unsafeAtomicAdd(&imf[2 * (ii + W + 1) + 1], v.y * c.w11);
          }
        }
      }
    }
  }
}

extern "C" int tike_patch_fwd(const void* images, void* patches, const float* positions,
                              int nimage, int H, int W, int nscan, int nrepeat, int patch_width,
                              int padded_width, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nimage >= 1 && H >= 1 && W >= 1 && nscan >= 0 && nrepeat >= 1);
  TK_CHECK_ARG(patch_width >= 1 && patch_width <= padded_width);
  const long nrow = (long)nimage * nscan * patch_width;
  if (nrow == 0) return TK_OK;
  TK_CHECK_ARG(images && patches && positions);
  hipLaunchKernelGGL((patch_kernel<false>), dim3(tk_grid(nrow, 16)), dim3(256), 0,
                     (hipStream_t)stream, (cf*)images, (cf*)patches, positions, nimage, H, W,
                     nscan, nrepeat, patch_width, padded_width, nscan * nrepeat);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_patch_adj(void* images, const void* patches, const float* positions,
                              int nimage, int H, int W, int nscan, int nrepeat, int patch_width,
                              int padded_width, int npatch, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nimage >= 1 && H >= 1 && W >= 1 && nscan >= 0 && nrepeat >= 1);
  TK_CHECK_ARG(patch_width >= 1 && patch_width <= padded_width);
  const long nrow = (long)nimage * nscan * patch_width;
  if (nrow == 0) return TK_OK;
  TK_CHECK_ARG(images && patches && positions);
  TK_CHECK_ARG(npatch >= nrepeat && ((long)nscan * nrepeat) % npatch == 0);
  hipLaunchKernelGGL((patch_kernel<true>), dim3(tk_grid(nrow, 16)), dim3(256), 0,
                     (hipStream_t)stream, (cf*)images, (cf*)patches, positions, nimage, H, W,
                     nscan, nrepeat, patch_width, padded_width, npatch);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------------------------ Convolution
// fwd: nearplane[n][s] = pad(patch_n(psi) * probe[n|0][s])      (convolution.py:58-101)
__global__ __launch_bounds__(256) void conv_fwd_kernel(const cf* __restrict__ psi,
                                                       const float* __restrict__ scan,
                                                       const TkProbe probe,
                                                       cf* __restrict__ nearplane, int nscan,
                                                       int S, int pw, int det, int H, int W) {
  const int pad = (det - pw) / 2;
  const long total = (long)H * W;
  const long nrow = (long)nscan * det;
  for (long row = blockIdx.x; row < nrow; row += gridDim.x) {
    const int dy = (int)(row % det);
    const long n = row / det;
    const int py = dy - pad;
    const TkCorner c = tk_corner(scan, n);
    const int y = c.sy + py;
    const bool row_ok = py >= 0 && py < pw && y >= 0 && y < H;
    for (int dx = threadIdx.x; dx < det; dx += blockDim.x) {
      const int px = dx - pad;
      const int x = c.sx + px;
      const bool ok = row_ok && px >= 0 && px < pw && x >= 0 && x < W;
      cf v = mk(0.f, 0.f);
      if (ok) v = tk_gather(psi, (long)y * W + x, W, total, c);
      for (int s = 0; s < S; ++s) {
        cf o = mk(0.f, 0.f);
        if (ok) o = v * probe.at(n, s, (long)py * pw + px);
        nearplane[((n * S + s) * det + dy) * (long)det + dx] = o;
      }
    }
  }
}

// adj: psi += scatter_n( sum_s conj(probe[n|0][s]) * crop(nearplane[n][s]) )   (:103-127)
__global__ __launch_bounds__(256) void conv_adj_kernel(const cf* __restrict__ nearplane,
                                                       const float* __restrict__ scan,
                                                       const TkProbe probe,
                                                       cf* __restrict__ psi, int nscan, int S,
                                                       int pw, int det, int H, int W) {
  const int pad = (det - pw) / 2;
  const long total = (long)H * W;
  float* __restrict__ imf = reinterpret_cast<float*>(psi);
  const long nrow = (long)nscan * pw;
  for (long row = blockIdx.x; row < nrow; row += gridDim.x) {
    const int py = (int)(row % pw);
    const long n = row / pw;
    const TkCorner c = tk_corner(scan, n);
    const int y = c.sy + py;
    if (y < 0 || y >= H) continue;
    for (int px = threadIdx.x; px < pw; px += blockDim.x) {
      const int x = c.sx + px;
      if (x < 0 || x >= W) continue;
      cf v = mk(0.f, 0.f);
      for (int s = 0; s < S; ++s) {
        const cf p = probe.at(n, s, (long)py * pw + px);
        const cf q = nearplane[((n * S + s) * det + pad + py) * (long)det + pad + px];
        v = v + conjf(p) * q;
      }
      const long ii = (long)y * W + x;
      unsafeAtomicAdd(&imf[2 * ii], v.x * c.w00);
      unsafeAtomicAdd(&imf[2 * ii + 1], v.y * c.w00);
      if (c.w01 != 0.0f && ii + 1 < total) {
        unsafeAtomicAdd(&imf[2 * (ii + 1)], v.x * c.w01);
        unsafeAtomicAdd(&imf[2 * (ii + 1) + 1], v.y * c.w01);
      }
      if (c.w10 != 0.0f && ii + W < total) {
        unsafeAtomicAdd(&imf[2 * (ii + W)], v.x * c.w10);
        unsafeAtomicAdd(&imf[2 * (ii + W) + 1], v.y * c.w10);
      }
      if (c.w11 != 0.0f && ii + W + 1 < total) {
        unsafeAtomicAdd(&imf[2 * (ii + W + 1)], v.x * c.w11);
        unsafeAtomicAdd(&imf[2 * (ii + W + 1) + 1], v.y * c.w11);
      }
    }
  }
}

// adj_probe: out[n][s] = conj(patch_n(psi)) * crop(nearplane[n][s])        (:129-154)
__global__ __launch_bounds__(256) void conv_adj_probe_kernel(const cf* __restrict__ nearplane,
                                                             const float* __restrict__ scan,
                                                             const cf* __restrict__ psi,
                                                             cf* __restrict__ out, int nscan,
                                                             int S, int pw, int det, int H,
                                                             int W) {
  const int pad = (det - pw) / 2;
  const long total = (long)H * W;
  const long nrow = (long)nscan * pw;
  for (long row = blockIdx.x; row < nrow; row += gridDim.x) {
    const int py = (int)(row % pw);
    const long n = row / pw;
    const TkCorner c = tk_corner(scan, n);
    const int y = c.sy + py;
    for (int px = threadIdx.x; px < pw; px += blockDim.x) {
      const int x = c.sx + px;
      const bool ok = y >= 0 && y < H && x >= 0 && x < W;
      cf v = mk(0.f, 0.f);
      if (ok) v = conjf(tk_gather(psi, (long)y * W + x, W, total, c));
      for (int s = 0; s < S; ++s) {
        const cf q = nearplane[((n * S + s) * det + pad + py) * (long)det + pad + px];
        out[((n * S + s) * pw + py) * (long)pw + px] = v * q;
      }
    }
  }
}

static int conv_check(const void* a, const void* b, const void* c, const void* d, int nscan,
                      int S, int pw, int det, int H, int W) {
  TK_CHECK_ARG(a && b && c && d);
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 1 && det >= pw && H >= 1 && W >= 1);
  return TK_OK;
}

int tk_conv_fwd(const cf* psi, const float* scan, const TkProbe& probe, cf* nearplane, int nscan,
                int S, int pw, int det, int H, int W, hipStream_t stream) {
  int rc = conv_check(psi, scan, probe.probe, nearplane, nscan, S, pw, det, H, W);
  if (rc) return rc;
  if (nscan == 0) return TK_OK;
  hipLaunchKernelGGL(conv_fwd_kernel, dim3(tk_grid((long)nscan * det, 16)), dim3(256), 0, stream,
                     psi, scan, probe, nearplane, nscan, S, pw, det, H, W);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

int tk_conv_adj(const cf* nearplane, const float* scan, const TkProbe& probe, cf* psi, int nscan,
                int S, int pw, int det, int H, int W, hipStream_t stream) {
  int rc = conv_check(psi, scan, probe.probe, nearplane, nscan, S, pw, det, H, W);
  if (rc) return rc;
  if (nscan == 0) return TK_OK;
  hipLaunchKernelGGL(conv_adj_kernel, dim3(tk_grid((long)nscan * pw, 16)), dim3(256), 0, stream,
                     nearplane, scan, probe, psi, nscan, S, pw, det, H, W);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

int tk_conv_adj_probe(const cf* nearplane, const float* scan, const cf* psi, cf* probe_adj,
                      int nscan, int S, int pw, int det, int H, int W, hipStream_t stream) {
  int rc = conv_check(psi, scan, probe_adj, nearplane, nscan, S, pw, det, H, W);
  if (rc) return rc;
  if (nscan == 0) return TK_OK;
  hipLaunchKernelGGL(conv_adj_probe_kernel, dim3(tk_grid((long)nscan * pw, 16)), dim3(256), 0,
                     stream, nearplane, scan, psi, probe_adj, nscan, S, pw, det, H, W);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_conv_fwd(const void* psi, const float* scan, const void* probe,
                             int probe_per_scan, void* nearplane, int nscan, int S, int pw,
                             int det, int H, int W, void* stream) {
  TK_ENTER();
  return tk_conv_fwd((const cf*)psi, scan,
                     tk_make_probe(probe, probe_per_scan, nullptr, nullptr, 0, 0, S, pw),
                     (cf*)nearplane, nscan, S, pw, det, H, W, (hipStream_t)stream);
}

extern "C" int tike_conv_adj(const void* nearplane, const float* scan, const void* probe,
                             int probe_per_scan, void* psi, int nscan, int S, int pw, int det,
                             int H, int W, void* stream) {
  TK_ENTER();
  return tk_conv_adj((const cf*)nearplane, scan,
                     tk_make_probe(probe, probe_per_scan, nullptr, nullptr, 0, 0, S, pw),
                     (cf*)psi, nscan, S, pw, det, H, W, (hipStream_t)stream);
}

extern "C" int tike_conv_adj_probe(const void* nearplane, const float* scan, const void* psi,
                                   void* probe_adj, int nscan, int S, int pw, int det, int H,
                                   int W, void* stream) {
  TK_ENTER();
  return tk_conv_adj_probe((const cf*)nearplane, scan, (const cf*)psi, (cf*)probe_adj, nscan, S,
                           pw, det, H, W, (hipStream_t)stream);
}

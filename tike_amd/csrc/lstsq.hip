// Kernels of the least-squares + gradient update loop (lstsq_grad) for gfx950.
//
// Reference: src/tike/ptycho/solvers/lstsq.py (one minibatch):
//   :506-520  object gradient  sum_s conj(P_n,s) chi_n,s scattered   -> tike_lstsq_gradients + tike_scatter_patches
//   :524-539  probe gradient   sum_n conj(O_n) chi_n,s               -> tike_lstsq_gradients (tike_probe_grad)
//   :619-718  step-size normal equations, per position               -> tike_lstsq_step_stats
//   :721-738  eigen-probe intensity coefficients                     -> (same kernel)
//   solvers/_preconditioner.py:116-167 probe preconditioner          -> tike_probe_preconditioner
// where chi is the exit-wave update (IFFT2 of the far-plane gradient cropped
// to the probe window), P_n,s the probe at position n (shared probe plus
// eigen probes synthesised on the fly) and O_n the bilinear object patch.
#include "fft_engine2.h"
#include <type_traits>

#include "internal.h"
#include "tike_amd.h"

#ifndef TK_STATS_PAIRS
#define TK_STATS_PAIRS 1  // build switch of the A/B (tools/build_variant.py)
#endif
static const bool g_stats_pairs = TK_STATS_PAIRS != 0;

// ------------------------------------------------- footprint scatter-add
// Adjoint of the bilinear patch gather with ONE atomic per object pixel and
// position instead of four per patch pixel: the patch value v[y][x] reaches
// the (pw+1)^2 object pixels (sy+y', sx+x') with
//   f[y'][x'] = (1-fy) u[y'][x'] + fy u[y'-1][x'],
//   u[y'][x'] = (1-fx) v[y'][x'] + fx v[y'][x'-1]          (v = 0 outside)
// which expands to the reference's four products w00..w11 (convolution.cu:
// 130-135).  A workgroup owns a strip of rows of one position; a thread owns
// a column, walks down the strip keeping u[y'-1] in registers and takes its
// left neighbour's v by wave shuffle.  Requires positions that keep the patch
// inside the image (check_allowed_positions, position.py:600-628); pixels
// falling outside are dropped.
constexpr int TK_STRIP = 32;
#define TK_ATOMIC_ADD(p, v) unsafeAtomicAdd(p, v)

// The accumulation image is PLANAR (all real parts, then all imaginary parts):
// one atomic wave-instruction then covers 256 contiguous bytes, the shape that
// runs at the full atomic rate (interleaved complex halves it).
// `sink(yp, xp, re, im)` receives the footprint value of row y' = yp, column
// x' = xp (0 <= yp, xp <= pw); rows y' in [r0, r1) are produced.
template <bool REAL_ONLY, class ValueFn, class Sink>
__device__ __forceinline__ void scatter_footprint_rows(ValueFn&& value, float fx, float fy,
                                                       int pw, int r0, int r1, Sink&& sink) {
  for (int x0 = 0; x0 < pw; x0 += blockDim.x) {
    const int xp = x0 + threadIdx.x;  // column x' (also the patch column)
    const bool active = xp < pw;
    cf uprev = mk(0.f, 0.f), uprev_last = mk(0.f, 0.f);
    constexpr int RG = 4;  // rows whose loads are issued together
    // Software pipeline: the loads of row group g+1 are issued BEFORE the
    // atomics of group g.  vmcnt retires in issue order, so a load issued
    // after an atomic would wait for that atomic's full round trip.
    cf vv[RG], ll[RG], nv[RG], nl[RG];
    auto load_group = [&](int yb, cf (&a)[RG], cf (&b)[RG]) {
      const int xpc = active ? xp : pw - 1;
#pragma unroll
      for (int k = 0; k < RG; ++k) {
        const int ypc = yb + k < pw ? yb + k : pw - 1;  // clamped, unconditional
        a[k] = value(ypc, xpc);
      }
      // left neighbour for lane 0 of each wave (the others take it by shuffle)
      if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < RG; ++k) {
          const int ypc = yb + k < pw ? yb + k : pw - 1;
          b[k] = value(ypc, xpc > 0 ? xpc - 1 : 0);
        }
      }
    };
    const int ystart = max(r0 - 1, 0);
    load_group(ystart, vv, ll);
    for (int yb = ystart; yb < r1; yb += RG) {
      if (yb + RG < r1) load_group(yb + RG, nv, nl);
#pragma unroll
      for (int k = 0; k < RG; ++k) {
        const int yp = yb + k;
        if (yp >= r1) break;
        cf v = (active && yp < pw) ? vv[k] : mk(0.f, 0.f);
        cf left = mk(__shfl_up(v.x, 1, 64), __shfl_up(v.y, 1, 64));
        if ((threadIdx.x & 63) == 0)
          left = (active && xp > 0 && yp < pw) ? ll[k] : mk(0.f, 0.f);
        const cf u = mk((1.0f - fx) * v.x + fx * left.x, (1.0f - fx) * v.y + fx * left.y);
        // the thread owning the last patch column also produces column x' = pw
        const cf ulast = mk(fx * v.x, fx * v.y);
        if (yp >= r0 && active) {
          sink(yp, xp, (1.0f - fy) * u.x + fy * uprev.x,
               REAL_ONLY ? 0.f : (1.0f - fy) * u.y + fy * uprev.y);
          if (xp == pw - 1)
            sink(yp, pw, (1.0f - fy) * ulast.x + fy * uprev_last.x,
                 REAL_ONLY ? 0.f : (1.0f - fy) * ulast.y + fy * uprev_last.y);
        }
        uprev = u;
        uprev_last = ulast;
      }
#pragma unroll
      for (int k = 0; k < RG; ++k) {
        vv[k] = nv[k];
        ll[k] = nl[k];
      }
    }
  }
}

// One position, one strip of TK_STRIP rows, straight to the image by atomics.
template <bool REAL_ONLY, class ValueFn>
__device__ __forceinline__ void scatter_footprint(ValueFn&& value, const TkCorner& c,
                                                  float fx, float fy, float* __restrict__ re,
                                                  float* __restrict__ im, int pw, int H, int W,
                                                  int strip) {
  const int r0 = strip * TK_STRIP;
  const int r1 = min(pw + 1, r0 + TK_STRIP);  // rows y' in [r0, r1)
  scatter_footprint_rows<REAL_ONLY>(value, fx, fy, pw, r0, r1,
                                    [&](int yp, int xp, float vr, float vi) {
                                      const int Y = c.sy + yp, X = c.sx + xp;
                                      if (Y >= 0 && Y < H && X >= 0 && X < W) {
                                        const long ii = (long)Y * W + X;
                                        TK_ATOMIC_ADD(&re[ii], vr);
                                        if (!REAL_ONLY) TK_ATOMIC_ADD(&im[ii], vi);
                                      }
                                    });
}

// ------------------------------------------- grouped footprint scatter-add
// Footprints of neighbouring scan positions overlap almost entirely (pw =
// 256 against a pitch of tens of pixels), so TK_GROUP CONSECUTIVE positions
// are summed on chip first -- over the bounding box of their footprints, one
// strip of TK_GROWS image rows per workgroup, one thread per box column with
// the row sums in registers (rounds 2-4: in LDS, a barrier per position) --
// and the image then takes ONE atomic per box pixel instead of one per
// position and pixel.  The caller orders positions so that consecutive ones
// are neighbours (the solver sorts every minibatch spatially); a group whose
// box is wider than TK_GSPREAD allows falls back to the per-position
// atomics, so any order gives the same sums.
constexpr int TK_GROUP = 8;
constexpr int TK_GROWS = 8;     // image rows per workgroup
constexpr int TK_GSPREAD = 112;  // extra box width and height beyond one footprint

struct TkGroupBox {
  int ymin, ymax, xmin, xmax;  // inclusive image bounds of the union footprint
};

__device__ __forceinline__ TkGroupBox tk_group_box(const float* __restrict__ scan, long n0,
                                                   long n1, int pw) {
  TkGroupBox b = {1 << 30, -(1 << 30), 1 << 30, -(1 << 30)};
  for (long n = n0; n < n1; ++n) {
    const int sy = (int)floorf(scan[2 * n]), sx = (int)floorf(scan[2 * n + 1]);
    b.ymin = min(b.ymin, sy);
    b.ymax = max(b.ymax, sy + pw);
    b.xmin = min(b.xmin, sx);
    b.xmax = max(b.xmax, sx + pw);
  }
  return b;
}

// lane i receives lane i - 1 (wave_shr:1); lane 0 receives 0
__device__ __forceinline__ float tk_lane_down(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}

// value(n, y, x): patch value of position n.  Round 5: a thread owns a COLUMN
// of the box and keeps its TK_GROWS row sums in registers -- no LDS, no
// barrier, and (inside a group) a fixed summation order.  For position n the
// thread's column is patch column x' = X - sx_n; it needs v[y'][x'] of the
// TK_GROWS + 1 patch rows that reach the strip and, for the tap to the left,
// its left neighbour's values: lane - 1 holds x' - 1 of the same position (DPP
// wave shift).  Lane 0 of every wave is a HALO lane: it repeats the last
// column of the wave before it, feeds lane 1 and writes nothing -- a wave
// covers 63 box columns and no lane ever needs a value from another wave
// (loading those at a wave-uniform address made scalar loads of them, each
// waited for on its own: 9 serial latencies per position).  Every load is
// unconditional (clamped address, value selected): the rows of the next
// position are requested before the sums of this one.
constexpr int TK_GCOLS = 63;  // box columns per wave

// rowptr(n, y): (uniform) pointer to row y of the patch of position n -- cf, or
// float when REAL_ONLY.  Columns outside the patch take weight 0 instead of a
// select per row (their clamped loads return finite values of the same row).
template <bool REAL_ONLY, class RowFn>
__device__ __forceinline__ void scatter_group(RowFn&& rowptr, const float* __restrict__ scan,
                                              long n0, long n1, int strip, int wmax,
                                              float* __restrict__ re, float* __restrict__ im,
                                              int pw, int H, int W) {
  using T = std::conditional_t<REAL_ONLY, float, cf>;
  auto ld = [](const T* p) {
    if constexpr (REAL_ONLY) return mk(*p, 0.f);
    else return *p;
  };
  const TkGroupBox b = tk_group_box(scan, n0, n1, pw);
  const int wb = b.xmax - b.xmin + 1;
  const int hb = b.ymax - b.ymin + 1;
  const int nstrip_direct = (pw + 1 + TK_STRIP - 1) / TK_STRIP;
  if (wb > wmax || hb > pw + 1 + TK_GSPREAD) {
    // positions too far apart for one box: per-position atomics; the
    // first workgroups of the group share the (position, strip) items
    const int nwg = (pw + 1 + TK_GSPREAD + TK_GROWS - 1) / TK_GROWS;
    for (long w = strip; w < (n1 - n0) * nstrip_direct; w += nwg) {
      const long n = n0 + w / nstrip_direct;
      const TkCorner c = tk_corner(scan, n);
      const float fy = scan[2 * n] - floorf(scan[2 * n]);
      const float fx = scan[2 * n + 1] - floorf(scan[2 * n + 1]);
      scatter_footprint<REAL_ONLY>([&](int y, int x) { return ld(rowptr(n, y) + x); }, c, fx,
                                   fy, re, im, pw, H, W, (int)(w % nstrip_direct));
    }
    return;
  }
  const int Y0 = b.ymin + strip * TK_GROWS;
  if (Y0 > b.ymax) return;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int nwave = (int)blockDim.x >> 6;
  const int lane = threadIdx.x & 63;
  constexpr int R = TK_GROWS + 1;  // patch rows y'_0 - 1 .. y'_0 + TK_GROWS - 1
  for (int c0 = wave * TK_GCOLS; c0 < wb; c0 += nwave * TK_GCOLS) {  // uniform
    const int X = b.xmin + c0 + lane - 1;  // lane 0: the column left of the wave's first
    // rows of position n at this thread's column; (wv, wl) = weights of the
    // thread's own value and of its left neighbour's in u = (1-fx) v + fx v_left
    auto load = [&](long n, cf (&v)[R], float& wv, float& wl) {
      const float py = scan[2 * n], px = scan[2 * n + 1];
      const int sy = (int)floorf(py), sx = (int)floorf(px);
      const float fx = px - floorf(px);
      const int xq = X - sx;
      const bool okx = (unsigned)xq < (unsigned)pw;
      wv = okx ? 1.0f - fx : 0.f;
      wl = (unsigned)(xq - 1) < (unsigned)pw ? fx : 0.f;
      const unsigned off = (unsigned)(okx ? xq : 0) * (unsigned)sizeof(T);
      const int y0 = Y0 - sy - 1;
      if (y0 >= 0 && y0 + R <= pw) {  // uniform: every row inside the patch
        const T* __restrict__ base = rowptr(n, y0);
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = ld(tk_at(base + (long)j * pw, off));
      } else {
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int y = y0 + j;
          const bool oky = y >= 0 && y < pw;
          const cf a = ld(tk_at(rowptr(n, oky ? y : 0), off));
          v[j] = oky ? a : mk(0.f, 0.f);
        }
      }
    };
    float ar[TK_GROWS], ai[TK_GROWS];
#pragma unroll
    for (int k = 0; k < TK_GROWS; ++k) ar[k] = ai[k] = 0.f;
    cf v[R], nv[R];
    float wv, wl, nwv = 0.f, nwl = 0.f;
    load(n0, v, wv, wl);
    for (long n = n0; n < n1; ++n) {
      if (n + 1 < n1) load(n + 1, nv, nwv, nwl);
      const float py = scan[2 * n];
      const float fy = py - floorf(py);
      cf u[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const cf left = mk(tk_lane_down(v[j].x), REAL_ONLY ? 0.f : tk_lane_down(v[j].y));
        u[j] = mk(wv * v[j].x + wl * left.x, REAL_ONLY ? 0.f : wv * v[j].y + wl * left.y);
      }
#pragma unroll
      for (int k = 0; k < TK_GROWS; ++k) {
        ar[k] += (1.0f - fy) * u[k + 1].x + fy * u[k].x;
        if (!REAL_ONLY) ai[k] += (1.0f - fy) * u[k + 1].y + fy * u[k].y;
      }
#pragma unroll
      for (int j = 0; j < R; ++j) v[j] = nv[j];
      wv = nwv;
      wl = nwl;
    }
    if (lane > 0 && X <= b.xmax && X >= 0 && X < W) {
#pragma unroll
      for (int k = 0; k < TK_GROWS; ++k) {
        const int Y = Y0 + k;
        if (Y > b.ymax || Y < 0 || Y >= H) continue;
        const long ii = (long)Y * W + X;
        if (REAL_ONLY) {
          if (ar[k] != 0.f) TK_ATOMIC_ADD(&re[ii], ar[k]);
        } else if (ar[k] != 0.f || ai[k] != 0.f) {
          TK_ATOMIC_ADD(&re[ii], ar[k]);
          TK_ATOMIC_ADD(&im[ii], ai[k]);
        }
      }
    }
  }
}

// Deterministic form of the same sum (tike_set_deterministic): a wave OWNS 63
// image columns of a strip of TK_GROWS rows and walks ALL positions in index
// order, adding the footprint values of those that reach its pixels in
// registers; the image is then updated by plain read-modify-writes -- every
// pixel has one owner, every sum one order.  Same arithmetic per position as
// scatter_group; no group boxes, so any position order costs the same.
template <bool REAL_ONLY, class RowFn>
__device__ __forceinline__ void scatter_ordered(RowFn&& rowptr, const float* __restrict__ scan,
                                                long nscan, float* __restrict__ re,
                                                float* __restrict__ im, int pw, int H, int W) {
  using T = std::conditional_t<REAL_ONLY, float, cf>;
  auto ld = [](const T* p) {
    if constexpr (REAL_ONLY) return mk(*p, 0.f);
    else return *p;
  };
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int nwave = (int)blockDim.x >> 6;
  const int lane = threadIdx.x & 63;
  constexpr int R = TK_GROWS + 1;
  const int Y0 = blockIdx.x * TK_GROWS;
  const int Xw = ((int)blockIdx.y * nwave + wave) * TK_GCOLS;  // first owned column (uniform)
  if (Y0 >= H || Xw >= W) return;
  const int X = Xw + lane - 1;  // lane 0: the halo column left of the first owned one
  float ar[TK_GROWS], ai[TK_GROWS];
#pragma unroll
  for (int k = 0; k < TK_GROWS; ++k) ar[k] = ai[k] = 0.f;
  for (long n = 0; n < nscan; ++n) {
    const float py = scan[2 * n], px = scan[2 * n + 1];
    const int sy = (int)floorf(py), sx = (int)floorf(px);
    // footprint rows [sy, sy + pw], columns [sx, sx + pw]: does it reach this
    // wave's pixels?  (uniform)
    if (sy > Y0 + TK_GROWS - 1 || sy + pw < Y0 || sx > Xw + TK_GCOLS - 1 || sx + pw < Xw) continue;
    const float fy = py - floorf(py), fx = px - floorf(px);
    const int xq = X - sx;
    const bool okx = (unsigned)xq < (unsigned)pw;
    const float wv = okx ? 1.0f - fx : 0.f;
    const float wl = (unsigned)(xq - 1) < (unsigned)pw ? fx : 0.f;
    const unsigned off = (unsigned)(okx ? xq : 0) * (unsigned)sizeof(T);
    const int y0 = Y0 - sy - 1;
    cf v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int y = y0 + j;
      const bool oky = y >= 0 && y < pw;
      const cf a = ld(tk_at(rowptr(n, oky ? y : 0), off));
      v[j] = oky ? a : mk(0.f, 0.f);
    }
    cf u[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const cf left = mk(tk_lane_down(v[j].x), REAL_ONLY ? 0.f : tk_lane_down(v[j].y));
      u[j] = mk(wv * v[j].x + wl * left.x, REAL_ONLY ? 0.f : wv * v[j].y + wl * left.y);
    }
#pragma unroll
    for (int k = 0; k < TK_GROWS; ++k) {
      ar[k] += (1.0f - fy) * u[k + 1].x + fy * u[k].x;
      if (!REAL_ONLY) ai[k] += (1.0f - fy) * u[k + 1].y + fy * u[k].y;
    }
  }
  if (lane > 0 && X < W) {
#pragma unroll
    for (int k = 0; k < TK_GROWS; ++k) {
      const int Y = Y0 + k;
      if (Y >= H) continue;
      const long ii = (long)Y * W + X;
      re[ii] += ar[k];
      if (!REAL_ONLY) im[ii] += ai[k];
    }
  }
}

// grid of the ordered form: (strips of the image, blocks of 4 x 63 columns)
static inline dim3 tk_ordered_grid(int H, int W) {
  return dim3((unsigned)((H + TK_GROWS - 1) / TK_GROWS),
              (unsigned)((W + 4 * TK_GCOLS - 1) / (4 * TK_GCOLS)));
}

__global__ __launch_bounds__(256) void scatter_patches_ordered_kernel(
    const cf* __restrict__ proj, const float* __restrict__ scan, float* __restrict__ acc,
    int nscan, int pw, int H, int W) {
  const long P = (long)pw * pw;
  scatter_ordered<false>([&](long n, int y) { return proj + n * P + (long)y * pw; }, scan, nscan,
                         acc, acc + (long)H * W, pw, H, W);
}

__global__ __launch_bounds__(256) void psi_precond_ordered_kernel(const float* __restrict__ amp,
                                                                  const float* __restrict__ scan,
                                                                  float* __restrict__ out,
                                                                  int nscan, int pw, int H, int W,
                                                                  long amp_stride) {
  scatter_ordered<true>([&](long n, int y) { return amp + n * amp_stride + (long)y * pw; }, scan,
                        nscan, out, out, pw, H, W);
}

// ----------------------------------------------------------- object gradient
// acc (2,H,W) planar f32 += scatter_n( objproj_n ),  objproj (nscan,pw,pw) c64 =
// sum_s conj(P_n,s) chi_n,s  computed by tike_lstsq_gradients
// (lstsq.py:510-520 = conj multiply + Patch.adj with nrepeat = S).
__global__ __launch_bounds__(1024) void scatter_patches_kernel(const cf* __restrict__ proj,
                                                               const float* __restrict__ scan,
                                                               float* __restrict__ acc, int nscan,
                                                               int pw, int H, int W, int wmax) {
  const long P = (long)pw * pw;
  float* __restrict__ re = acc;
  float* __restrict__ im = acc + (long)H * W;
  const long g = blockIdx.y;
  const long n0 = g * TK_GROUP, n1 = min((long)nscan, n0 + TK_GROUP);
  scatter_group<false>([&](long n, int y) { return proj + n * P + (long)y * pw; }, scan, n0, n1,
                       blockIdx.x, wmax, re, im, pw, H, W);
}

// (strips per group, widest box, threads per workgroup) of the grouped scatter
// for a probe width: one thread per box column, whole waves
static inline void tk_group_geometry(int pw, int* nstrip, int* wmax, int* threads) {
  *wmax = pw + 1 + TK_GSPREAD;
  *nstrip = (pw + 1 + TK_GSPREAD + TK_GROWS - 1) / TK_GROWS;
  const int t = (*wmax + TK_GCOLS - 1) / TK_GCOLS * 64;  // a wave covers TK_GCOLS columns
  *threads = t > 1024 ? 1024 : t;
}

extern "C" int tike_scatter_patches(const void* objproj, const float* scan, float* acc,
                                    int nscan, int pw, int H, int W, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && pw >= 1 && H >= 1 && W >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(objproj && scan && acc);
  if (tk_deterministic()) {
    hipLaunchKernelGGL(scatter_patches_ordered_kernel, tk_ordered_grid(H, W), dim3(256), 0,
                       (hipStream_t)stream, (const cf*)objproj, scan, acc, nscan, pw, H, W);
    TK_LAUNCH_CHECK();
    return TK_OK;
  }
  int nstrip, wmax, threads;
  tk_group_geometry(pw, &nstrip, &wmax, &threads);
  const dim3 grid(nstrip, (nscan + TK_GROUP - 1) / TK_GROUP);
  hipLaunchKernelGGL(scatter_patches_kernel, grid, dim3(threads), 0, (hipStream_t)stream,
                     (const cf*)objproj, scan, acc, nscan, pw, H, W, wmax);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------------------ psi preconditioner
// out (H,W) float32 += scatter_n( probe_amp ),  probe_amp = sum_s |probe_s|^2 (pw,pw) f32
// (solvers/_preconditioner.py:48-104: Patch.adj of one broadcast patch).
__global__ __launch_bounds__(1024) void psi_precond_kernel(const float* __restrict__ amp,
                                                           const float* __restrict__ scan,
                                                           float* __restrict__ out, int nscan,
                                                           int pw, int H, int W, int wmax) {
  const long g = blockIdx.y;
  const long n0 = g * TK_GROUP, n1 = min((long)nscan, n0 + TK_GROUP);
  scatter_group<true>([&](long, int y) { return amp + (long)y * pw; }, scan, n0, n1,
                      blockIdx.x, wmax, out, out, pw, H, W);
}

extern "C" int tike_psi_preconditioner(const float* probe_amp, const float* scan, void* out,
                                       int nscan, int pw, int H, int W, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && pw >= 1 && H >= 1 && W >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(probe_amp && scan && out);
  if (tk_deterministic()) {
    hipLaunchKernelGGL(psi_precond_ordered_kernel, tk_ordered_grid(H, W), dim3(256), 0,
                       (hipStream_t)stream, probe_amp, scan, (float*)out, nscan, pw, H, W, 0L);
    TK_LAUNCH_CHECK();
    return TK_OK;
  }
  int nstrip, wmax, threads;
  tk_group_geometry(pw, &nstrip, &wmax, &threads);
  const dim3 grid(nstrip, (nscan + TK_GROUP - 1) / TK_GROUP);
  hipLaunchKernelGGL(psi_precond_kernel, grid, dim3(threads), 0, (hipStream_t)stream, probe_amp,
                     scan, (float*)out, nscan, pw, H, W, wmax);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// out (H,W) float32 += scatter_n( amp_n ), amp (nscan,pw,pw) f32: one real
// patch PER POSITION (the illumination of a slice behind the first of a
// multislice object, _preconditioner.py:82-95).
__global__ __launch_bounds__(1024) void scatter_amplitudes_kernel(const float* __restrict__ amp,
                                                                  const float* __restrict__ scan,
                                                                  float* __restrict__ out,
                                                                  int nscan, int pw, int H, int W,
                                                                  int wmax) {
  const long P = (long)pw * pw;
  const long g = blockIdx.y;
  const long n0 = g * TK_GROUP, n1 = min((long)nscan, n0 + TK_GROUP);
  scatter_group<true>([&](long n, int y) { return amp + n * P + (long)y * pw; }, scan, n0, n1,
                      blockIdx.x, wmax, out, out, pw, H, W);
}

extern "C" int tike_scatter_amplitudes(const float* amp, const float* scan, float* out,
                                       int nscan, int pw, int H, int W, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && pw >= 1 && H >= 1 && W >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(amp && scan && out);
  if (tk_deterministic()) {
    hipLaunchKernelGGL(psi_precond_ordered_kernel, tk_ordered_grid(H, W), dim3(256), 0,
                       (hipStream_t)stream, amp, scan, out, nscan, pw, H, W, (long)pw * pw);
    TK_LAUNCH_CHECK();
    return TK_OK;
  }
  int nstrip, wmax, threads;
  tk_group_geometry(pw, &nstrip, &wmax, &threads);
  const dim3 grid(nstrip, (nscan + TK_GROUP - 1) / TK_GROUP);
  hipLaunchKernelGGL(scatter_amplitudes_kernel, grid, dim3(threads), 0, (hipStream_t)stream, amp,
                     scan, out, nscan, pw, H, W, wmax);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------------------------ probe gradient
// One thread per probe pixel, a workgroup walks a chunk of positions keeping S
// complex accumulators in registers; one atomic pair per (pixel, mode, chunk).
// Optionally stores the object patches (B, pw, pw) for later passes.
constexpr int TK_MAX_MODES = 16;

// SC = compile-time number of modes (0: runtime S <= TK_MAX_MODES).  With SC
// known the mode loop has no branches, so all S loads of a position are in
// flight together instead of one memory latency per mode.
template <bool WITH_CHI, int SC>
__global__ __launch_bounds__(256) void probe_grad_kernel(
    const cf* __restrict__ chi, const float* __restrict__ scan, const cf* __restrict__ psi,
    cf* __restrict__ patches, float* __restrict__ out, const TkProbe probe,
    cf* __restrict__ objproj, int nscan, int S_rt, int pw, int H, int W, int chunk,
    float* __restrict__ part) {
  constexpr int SM = SC > 0 ? SC : TK_MAX_MODES;
  const int S = SC > 0 ? SC : S_rt;
  const long P = (long)pw * pw;
  const long total = (long)H * W;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b0 = blockIdx.y * chunk;
  const int b1 = min(nscan, b0 + chunk);
  if (p >= P) return;
  const int py = (int)(p / pw), px = (int)(p % pw);
  cf acc[SM];
#pragma unroll
  for (int s = 0; s < SM; ++s) acc[s] = mk(0.f, 0.f);
  // The shared probe value of this pixel is the same for every position of
  // the chunk: keep it in registers instead of re-reading it per position
  // (only the modes that own eigen probes vary beyond a scalar weight).
  const bool hoist = WITH_CHI && objproj != nullptr && probe.pos_stride == 0;
  cf pr[SM];
  if (hoist) {
#pragma unroll
    for (int s = 0; s < SM; ++s)
      if (SC > 0 || s < S) pr[s] = probe.probe[s * P + p];
  }
  for (int b = b0; b < b1; ++b) {
    const TkCorner c = tk_corner(scan, b);
    const int y = c.sy + py, x = c.sx + px;
    const bool ok = y >= 0 && y < H && x >= 0 && x < W;
    const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
    const int xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
    cf xs[SM];
    if (WITH_CHI) {
#pragma unroll
      for (int s = 0; s < SM; ++s)
        if (SC > 0 || s < S) xs[s] = tk_ld_stream(chi + ((long)b * S + s) * P + p);
    }
    cf o;
    // interior position (uniform): the two taps of a row are adjacent complex
    // values, one 16-byte load each -- half the L1 requests of four 8-byte taps
    if (!WITH_CHI && c.sy >= 0 && c.sx >= 0 && c.sy + pw < H && c.sx + pw < W &&
        total < (1L << 28)) {
      typedef float tk_v4f __attribute__((ext_vector_type(4)));
      const unsigned off = (unsigned)(y * W + x) * (unsigned)sizeof(cf);
      tk_v4f u, l;
      __builtin_memcpy(&u, reinterpret_cast<const char*>(psi) + off, sizeof(u));
      __builtin_memcpy(&l, reinterpret_cast<const char*>(psi) + off + (unsigned)W * 8u, sizeof(l));
      o = mk(u.x * c.w00, u.y * c.w00);
      o.x += u.z * c.w01;
      o.y += u.w * c.w01;
      o.x += l.x * c.w10;
      o.y += l.y * c.w10;
      o.x += l.z * c.w11;
      o.y += l.w * c.w11;
    } else {
      o = tk_gather(psi, (long)yc * W + xc, W, total, c);
      if (!ok) o = mk(0.f, 0.f);
    }
    if (patches) patches[b * P + p] = o;
    if (WITH_CHI) {
      const cf oc = conjf(o);
      cf proj = mk(0.f, 0.f);
#pragma unroll
      for (int s = 0; s < SM; ++s)
        if (SC > 0 || s < S) {
          if (out) acc[s] = acc[s] + oc * xs[s];
          if (objproj) {
            cf ps;
            if (hoist && !(probe.weights != nullptr && probe.eigen != nullptr && s < probe.Sm)) {
              const float w0 =
                  probe.weights ? probe.weights[b * (long)(probe.C + 1) * probe.S + s] : 1.0f;
              ps = pr[s] * w0;
            } else {
              ps = probe.at(b, s, p);
            }
            proj = proj + conjf(ps) * xs[s];
          }
        }
      if (objproj) objproj[b * P + p] = proj;
    } else {
      acc[0].x += norm2(o);
    }
  }
  // deterministic mode: the chunk's sums go to its own row of `part`, added in
  // a fixed order after the launch (tk_ordered_sum)
  if (WITH_CHI) {
    if (out) {
#pragma unroll
      for (int s = 0; s < SM; ++s)
        if (SC > 0 || s < S) {
          if (part != nullptr) {
            float* o = part + 2 * (((long)blockIdx.y * S + s) * P + p);
            o[0] = acc[s].x;
            o[1] = acc[s].y;
          } else {
            unsafeAtomicAdd(&out[2 * (s * P + p)], acc[s].x);
            unsafeAtomicAdd(&out[2 * (s * P + p) + 1], acc[s].y);
          }
        }
    }
  } else if (part != nullptr) {
    part[(long)blockIdx.y * P + p] = acc[0].x;
  } else {
    unsafeAtomicAdd(&out[2 * p], acc[0].x);
  }
}

template <bool WITH_CHI>
static void launch_probe_grad(dim3 grid, hipStream_t stream, const cf* chi, const float* scan,
                              const cf* psi, cf* patches, float* out, const TkProbe& probe,
                              cf* objproj, int nscan, int S, int pw, int H, int W, int chunk,
                              float* part = nullptr) {
#define TK_PG(SC)                                                                              \
  hipLaunchKernelGGL((probe_grad_kernel<WITH_CHI, SC>), grid, dim3(256), 0, stream, chi, scan, \
                     psi, patches, out, probe, objproj, nscan, S, pw, H, W, chunk, part)
  switch (WITH_CHI ? S : 1) {
    case 1: TK_PG(1); break;
    case 2: TK_PG(2); break;
    case 3: TK_PG(3); break;
    case 4: TK_PG(4); break;
    case 5: TK_PG(5); break;
    case 6: TK_PG(6); break;
    case 8: TK_PG(8); break;
    default: TK_PG(0); break;
  }
#undef TK_PG
}

// Positions per chunk of the sums over positions, and -- deterministic mode --
// where the chunks leave their partial sums (`len` floats each; *part stays
// NULL otherwise).  When the caller's scratch buffer cannot hold them the
// launch falls back to ONE chunk, so that every sum has a single contributor
// per address (its one atomic then only adds to what earlier, stream-ordered
// launches left there).
static int probe_chunk(int nscan, long len = 0, float** part = nullptr) {
  // enough position chunks to fill the chip, at least 8 positions each
  int chunk = (nscan + 31) / 32;
  chunk = chunk < 8 ? 8 : chunk;
  if (!tk_deterministic()) return chunk;
  const int nchunk = (nscan + chunk - 1) / chunk;
  float* p = len > 0 && part ? tk_det_scratch(sizeof(float) * (size_t)len * nchunk) : nullptr;
  if (p != nullptr) {
    *part = p;
    return chunk;
  }
  return nscan > 8 ? nscan : 8;
}

extern "C" int tike_probe_grad(const void* chi, const float* scan, const void* psi,
                               void* patches, void* m_probe_update, int nscan, int S, int pw,
                               int H, int W, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && S <= TK_MAX_MODES && pw >= 1 && H >= 1 && W >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(chi && scan && psi && m_probe_update);
  const long P = (long)pw * pw;
  float* part = nullptr;
  const int chunk = probe_chunk(nscan, 2 * S * P, &part);
  dim3 grid((unsigned)((P + 255) / 256), (unsigned)((nscan + chunk - 1) / chunk));
  launch_probe_grad<true>(grid, (hipStream_t)stream, (const cf*)chi, scan, (const cf*)psi,
                          (cf*)patches, (float*)m_probe_update,
                          tk_make_probe(psi, 0, nullptr, nullptr, 0, 0, S, pw), (cf*)nullptr,
                          nscan, S, pw, H, W, chunk, part);
  TK_LAUNCH_CHECK();
  if (part != nullptr)
    return tk_ordered_sum((float*)m_probe_update, part, 2 * S * P, (int)grid.y, true,
                          (hipStream_t)stream);
  return TK_OK;
}

// One pass over chi for BOTH gradients (lstsq.py:506-539):
//   m_probe_update (S,pw,pw) += sum_n conj(O_n) chi_n,s          (may be NULL)
//   objproj (nscan,pw,pw)     = sum_s conj(P_n,s) chi_n,s        (may be NULL)
//   patches (nscan,pw,pw)     = O_n = patch_n(psi)               (may be NULL)
extern "C" int tike_lstsq_gradients(const void* chi, const float* scan, const void* psi,
                                    const void* probe, const void* eigen_probe,
                                    const float* eigen_weights, int num_eigen, int eigen_modes,
                                    const void* unique_probe, void* patches,
                                    void* m_probe_update, void* objproj, int nscan, int S,
                                    int pw, int H, int W, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && S <= TK_MAX_MODES && pw >= 1 && H >= 1 && W >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(chi && scan && psi && probe);
  const long P = (long)pw * pw;
  float* part = nullptr;
  const int chunk = probe_chunk(nscan, m_probe_update ? 2 * S * P : 0, &part);
  dim3 grid((unsigned)((P + 255) / 256), (unsigned)((nscan + chunk - 1) / chunk));
  launch_probe_grad<true>(grid, (hipStream_t)stream, (const cf*)chi, scan, (const cf*)psi,
                          (cf*)patches, (float*)m_probe_update,
                          tk_make_probe(probe, 0, eigen_probe, eigen_weights, num_eigen,
                                        eigen_modes, S, pw, unique_probe),
                          (cf*)objproj, nscan, S, pw, H, W, chunk, part);
  TK_LAUNCH_CHECK();
  if (part != nullptr)
    return tk_ordered_sum((float*)m_probe_update, part, 2 * S * P, (int)grid.y, true,
                          (hipStream_t)stream);
  return TK_OK;
}

// ------------------------------------------- inverse pass 2 + both gradients
// The second pass of the inverse 2-D FFT (fft_engine2.h: in-place radix-RB over
// rows {ya + 16 k}) run PIXEL-major and fused with everything that consumes
// the exit-wave update chi (lstsq.py:504-539), so chi is never stored:
//   objproj_n      = sum_s conj(P_n,s) chi_n,s      (one write per position)
//   m_probe_update += sum_n conj(O_n) chi_n,s       (register accumulators over
//                                                    a chunk of positions, one
//                                                    atomic per pixel/mode/chunk)
//   chi0_n         = chi_n,0                        (step sizes, eigen probes,
//                                                    position correction)
// Probe window = detector (pw == N).  A workgroup owns one slice of the tile:
// the RB rows {ya + 16 yb} -- exactly what one radix-RB butterfly per thread
// consumes and produces -- by 64 * (4 / MW) columns, and walks a chunk of
// positions.  Its four waves are MW mode-waves x (4 / MW) column-waves: wave
// (mw, cw) handles modes {mw, mw + MW, ...} (MPW of them) of column block cw,
// lane = column, so every global access is a 512-byte row segment at one of
// the RB offsets off0 + yb * 16 N (the same offsets for the intermediate,
// the patches, the probe and every output).  The per-position sum over modes
// crosses the mode-waves through LDS: every wave leaves its partial sum in a
// slot of its own, one barrier, then mode-wave mw adds the slots of rows
// yb = mw (mod MW) and writes them (slots double buffered where they fit).
struct TkModeProbe {  // probe of one (position, mode): uniform values
  const cf* base;     // shared probe of the mode, or its synthesised varying probe
  float w0;           // scale of `base`
  int nE;             // eigen probes to add on the fly (0 when `base` is final)
};

// EIG: eigen probes are applied on the fly (their loops cost the 512^2
// instantiation 19 spilled registers when compiled in and never taken).
// GRP (round 6: more modes than one launch holds in registers -- 9 .. 16 at
// 256^2): the launch serves S consecutive modes of a problem with Stot modes
// per position (`mid`, `probe`, the weights and `mpu` arrive offset to the
// first of them; tiles and mode_scale are Stot apart) and, `accumulate`, adds
// its projection to what the launch of the group in front left in objproj.
template <int N, int MW, int MPW, bool HAVE_PROJ, bool EIG = true, bool GRP = false>
__global__ __launch_bounds__(256, 2) void ifft2_pass2_gradients_kernel(
    const cf* __restrict__ mid, const cf* __restrict__ patches, const TkProbe probe,
    cf* __restrict__ objproj, cf* __restrict__ chi0, float* __restrict__ mpu, float mpu_scale,
    int nscan, int S, float inv_scale, int chunk, float* __restrict__ mpu_part,
    const float* __restrict__ mode_scale, int Stot_ = 0, int accumulate_ = 0) {
  const int Stot = GRP ? Stot_ : S;
  const bool accumulate = GRP && accumulate_ != 0;
  constexpr int RB = N / 16;
  constexpr int CW = 4 / MW;            // column-waves per workgroup
  constexpr int NCB = N / (64 * CW);    // column blocks
  constexpr int NSLICE = 16 * NCB;      // (ya, column block) slices
  constexpr bool REDUCE = HAVE_PROJ && MW > 1;
  // RB = 32 (N = 512): the accumulators and one butterfly already fill the
  // register file, so probe and patch values are re-read (L2) per use
  constexpr bool HOIST = RB <= 16;
  static_assert(NCB >= 1 && NSLICE % 8 == 0, "slice layout");
  constexpr int NBUF = RB <= 16 ? 2 : 1;  // slot sets (2: one barrier per position)
  __shared__ cf part[REDUCE ? NBUF * 4 * RB * 64 : 1];  // [buf][wave][yb][lane]
  extern __shared__ cf eigl[];  // conj(E_c,s) on this slice: [C][Sm][RB][64 CW]
  constexpr long P = (long)N * N;
  // XCD-aware slice order: workgroup v runs on XCD v % 8 (round-robin
  // dispatch); every XCD keeps NSLICE/8 slices, so its L2 holds 1/8 of the
  // probe and of the patches.  Placement affects speed only.
  const int v = blockIdx.x;
  constexpr int per = NSLICE / 8;
  const int slice = (v & 7) * per + (v >> 3) % per;
  // chunks in DESCENDING order: the inverse pass 1 wrote the intermediate in
  // ascending position order, its tail is still in the Infinity Cache
  const int nchunk_ = (nscan + chunk - 1) / chunk;
  const int b0 = (nchunk_ - 1 - ((v >> 3) / per)) * chunk;
  const int b1 = min(nscan, b0 + chunk);
  const int ya = slice / NCB, cb = slice % NCB;
  // the wave index is uniform: say so, so that everything derived from it
  // (mode, column block, base pointers, weights) lives in scalar registers
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int mw = w % MW, cw = w / MW;
  constexpr long ROW = 16 * N;  // elements between the rows of this slice
  // uniform element offset of the slice's first row and column block, and the
  // per-lane byte offset inside a row segment
  const long slice0 = (long)ya * N + (cb * CW + cw) * 64;
  const unsigned lb = (unsigned)lane * (unsigned)sizeof(cf);
  if (EIG && HAVE_PROJ && probe.weights != nullptr && probe.eigen != nullptr) {
    const int total = probe.C * probe.Sm * RB * 64 * CW;
    for (int i = threadIdx.x; i < total; i += 256) {
      const int x = i % (64 * CW), yb = (i / (64 * CW)) % RB, cs = i / (64 * CW * RB);
      eigl[i] = conjf(probe.eigen[(long)cs * P + ya * N + yb * ROW + cb * CW * 64 + x]);
    }
  }
  __syncthreads();
  static_assert(MW > 1 || MPW == 1, "a lone mode-wave writes objproj straight from one mode");
  cf acc[MPW][RB];
  // conj(shared probe) at this thread's pixels
  cf Pc[HAVE_PROJ && HOIST ? MPW : 1][HAVE_PROJ && HOIST ? RB : 1];
#pragma unroll
  for (int m = 0; m < MPW; ++m) {
    const int s = mw + MW * m;
    const int sc = s < S ? s : S - 1;  // idle (wave, m): any valid mode, result unused
#pragma unroll
    for (int yb = 0; yb < RB; ++yb) {
      acc[m][yb] = mk(0.f, 0.f);
      if (HAVE_PROJ && HOIST)
        Pc[m][yb] = conjf(*tk_at(probe.probe + (long)sc * P + slice0 + yb * ROW, lb));
    }
  }
  // eigen probes vary the probe of the first Sm modes per position
  // (probe.py:272-303); only the waves that own those modes meet them
  const bool vary = HAVE_PROJ && probe.weights != nullptr;
  const int nE = (EIG && vary && probe.eigen != nullptr) ? probe.C : 0;
  for (int n = b0; n < b1; ++n) {
    // keep the per-lane offset out of the loop's induction variables: bases
    // stay in scalar registers, one 32-bit VGPR offset serves every access
    unsigned lo = lb;
    asm volatile("" : "+v"(lo));
    const cf* __restrict__ On = patches + (long)n * P + slice0;
    // with two modes per wave the patch values are re-read for the second
    // one (an L1/L2 hit) rather than held across both: 32 registers
    constexpr bool O_ONCE = HOIST && MPW == 1;
    cf O[HOIST ? RB : 1];
    if (O_ONCE) {
#pragma unroll
      for (int yb = 0; yb < RB; ++yb) O[yb] = *tk_at(On + yb * ROW, lo);
    }
    // this wave's slot, and slot 0 of its column block, for this position
    cf* slot = part + ((((n - b0) & (NBUF - 1)) * 4 + w) * RB) * 64 + lane;
    const cf* slots = part + ((((n - b0) & (NBUF - 1)) * 4 + cw * MW) * RB) * 64 + lane;
    const float* __restrict__ wn =
        vary ? probe.weights + n * (long)(probe.C + 1) * probe.S : nullptr;
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
      const int s = mw + MW * m;
      if (s < S) {  // wave-uniform
        const cf* __restrict__ src = mid + ((long)n * Stot + s) * P + slice0;
        cf u[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) u[k] = tk_ld_stream(tk_at(src + k * ROW, lo));
        if (HOIST && !O_ONCE) {
#pragma unroll
          for (int yb = 0; yb < RB; ++yb) O[yb] = *tk_at(On + yb * ROW, lo);
        }
        const float w0 = vary ? wn[s] : 1.0f;
        // (poisson step lengths that became known after pass 1 was written:
        // a uniform factor per position and mode)
        const float sc = mode_scale ? inv_scale * mode_scale[(long)n * Stot + s] : inv_scale;
        Dft<RB, true>::run(u);
        if (HOIST) {
#pragma unroll
          for (int yb = 0; yb < RB; ++yb) {
            u[yb] = u[yb] * sc;  // chi of row ya + 16 yb
            acc[m][yb] = acc[m][yb] + conjf(O[yb]) * u[yb];
          }
        } else {
#pragma unroll
          for (int g = 0; g < RB; g += 8) {
            cf o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = *tk_at(On + (g + i) * ROW, lo);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              u[g + i] = u[g + i] * sc;
              acc[m][g + i] = acc[m][g + i] + conjf(o[i]) * u[g + i];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (s == 0 && chi0 != nullptr) {
#pragma unroll
          for (int yb = 0; yb < RB; ++yb)
            tk_st_stream(tk_at(chi0 + (long)n * P + slice0 + yb * ROW, lo), u[yb]);
        }
        if (HAVE_PROJ) {
          const cf* __restrict__ Ps = probe.probe + (long)s * P + slice0;
          const bool eig = EIG && nE > 0 && s < probe.Sm;  // wave-uniform, rare
#pragma unroll
          for (int g = 0; g < RB; g += 8) {
            cf pc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
              pc[i] = HOIST ? Pc[m][g + i] : conjf(*tk_at(Ps + (g + i) * ROW, lo));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              cf t = (pc[i] * u[g + i]) * w0;
              if (REDUCE) {
                // first mode of the wave fills its slot, later ones add to it
                // (a wave's LDS operations execute in order)
                slot[(g + i) * 64] = m == 0 ? t : slot[(g + i) * 64] + t;
              } else {
                // a lone mode-wave: the whole projection is this one product
                if (eig) {
#pragma unroll 1
                  for (int c = 0; c < nE; ++c)
                    t = t + (eigl[((c * probe.Sm + s) * RB + g + i) * (64 * CW) + cw * 64 + lane] *
                             u[g + i]) * wn[(c + 1) * probe.S + s];
                }
                tk_st_stream(tk_at(objproj + (long)n * P + slice0 + (g + i) * ROW, lo), t);
              }
            }
            if (!HOIST) __builtin_amdgcn_sched_barrier(0);
          }
          if (REDUCE && eig) {
            // + sum_c w_c conj(E_c,s) chi from the LDS-resident eigen slices
            // (only the waves owning the first Sm modes)
#pragma unroll 1
            for (int c = 0; c < nE; ++c) {
              const float wc = wn[(c + 1) * probe.S + s];
              const cf* __restrict__ el =
                  eigl + ((c * probe.Sm + s) * RB) * (64 * CW) + cw * 64 + lane;
#pragma unroll
              for (int yb = 0; yb < RB; ++yb)
                slot[yb * 64] = slot[yb * 64] + (el[yb * (64 * CW)] * u[yb]) * wc;
            }
          }
        }
      } else if (REDUCE && m == 0) {
        // idle mode-wave (fewer modes than waves): an empty partial sum
#pragma unroll
        for (int yb = 0; yb < RB; ++yb) slot[yb * 64] = mk(0.f, 0.f);
      }
    }
    if (REDUCE) {
      __syncthreads();
      // mode-wave mw finishes rows yb = mw, mw + MW, ... of its column block
#pragma unroll
      for (int q = 0; q < RB / MW; ++q) {
        const int yb = mw + MW * q;
        cf sum = slots[yb * 64];
#pragma unroll
        for (int k = 1; k < MW; ++k) sum = sum + slots[(k * RB + yb) * 64];
        if (accumulate) sum = sum + *tk_at(objproj + (long)n * P + slice0 + yb * ROW, lo);
        tk_st_stream(tk_at(objproj + (long)n * P + slice0 + yb * ROW, lo), sum);
      }
      if (NBUF == 1) __syncthreads();  // the single slot set is rewritten next
    }
  }
  if (mpu != nullptr) {
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
      const int s = mw + MW * m;
      if (s < S) {
#pragma unroll
        for (int yb = 0; yb < RB; ++yb) {
          if (mpu_part != nullptr) {
            // deterministic mode: this chunk's partial sum, added up in chunk
            // order by tk_ordered_sum after the launch
            const long ci = (nchunk_ - 1) - b0 / chunk;
            float* o = tk_at(mpu_part + 2 * ((ci * S + s) * P + slice0 + yb * ROW), lb);
            o[0] = acc[m][yb].x * mpu_scale;
            o[1] = acc[m][yb].y * mpu_scale;
          } else {
            float* o = tk_at(mpu + 2 * ((long)s * P + slice0 + yb * ROW), lb);
            unsafeAtomicAdd(o, acc[m][yb].x * mpu_scale);
            unsafeAtomicAdd(o + 1, acc[m][yb].y * mpu_scale);
          }
        }
      }
    }
  }
}

// work (nscan,S,det,det): output of tike_grad_ifft2_pass1 / tike_ifft2_pass1_scaled;
// patches (nscan,det,det): O_n from the forward kernel.  Outputs (each may be
// NULL): objproj (nscan,det,det), chi0 (nscan,det,det), m_probe_update
// (S,det,det, accumulated).  Probe window = detector; det in {128, 256, 512};
// S <= 8 (TIKE_ERR_UNSUPPORTED otherwise: use tike_ifft2_crop* +
// tike_lstsq_gradients).
static int launch_pass2_gradients(const void* work, const void* patches, const void* probe,
                                  const void* eigen_probe, const float* eigen_weights,
                                  int num_eigen, int eigen_modes, void* objproj, void* chi0,
                                  void* m_probe_update, float mpu_scale, int nscan, int S,
                                  int det, float inv_scale, const float* mode_scale,
                                  hipStream_t stream, int Stot = 0, int accumulate = 0) {
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && det >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(work && patches && (probe || !objproj));
  if (S > 8 || (det != 128 && det != 256 && det != 512)) return TK_ERR_UNSUPPORTED;
  // a group of S modes out of Stot (tike_ifft2_pass2_gradients_modes): the
  // weights are Stot apart, as the tiles
  const bool grp = Stot != 0;
  // (with objproj: through the mode-sum path, which needs two mode-waves)
  if (grp && objproj && S < 2) return TK_ERR_UNSUPPORTED;
  const TkProbe pr = tk_make_probe(probe, 0, eigen_probe, eigen_weights, num_eigen, eigen_modes,
                                   grp ? Stot : S, det);
  // mode-waves x column-waves of a workgroup and modes per wave
  int MW = S >= 3 ? 4 : S;
  if (det == 128 && MW == 1) MW = 2;  // a 128-wide tile has only two 64-column waves
  // eigen probes applied on the fly keep conj(E) of the workgroup's slice in LDS
  // (32 KiB at most: two workgroups per CU): a slice of 4 / MW column-waves --
  // with one or two modes at 512^2 (or several eigen probes) more mode-waves,
  // the spare ones idle, make it narrow enough
  auto eig_bytes = [&](int mw) {
    return sizeof(cf) * (size_t)num_eigen * eigen_modes * (det / 16) * 64 * (4 / mw);
  };
  if (objproj && eigen_weights && eigen_probe)
    while (MW < 4 && eig_bytes(MW) > 32 * 1024) MW *= 2;
  const int MPW = S > 4 ? 2 : 1;
  const int nslice = 16 * (det / (64 * (4 / MW)));
  // enough (slice, chunk) workgroups to fill the chip about twice -- and
  // in WHOLE rounds: the kernel holds two workgroups per CU (512 at a time) and
  // every workgroup walks the same number of positions, so 4.5 rounds cost 5
  int nchunk = (1024 + nslice - 1) / nslice;
  {
    int unit = 512, a = nslice;  // unit = 512 / gcd(512, nslice)
    while (a % 2 == 0 && unit > 1) { a /= 2; unit /= 2; }
    if (nchunk >= unit) nchunk = nchunk / unit * unit;
  }
  int chunk = (nscan + nchunk - 1) / nchunk;
  if (chunk < 8) chunk = 8;
  nchunk = (nscan + chunk - 1) / chunk;
  const dim3 grid((unsigned)(nslice * nchunk)), block(256);
  // deterministic mode: per-chunk partial sums of the probe gradient in the
  // caller's scratch buffer (one chunk when it is too small)
  float* mpu_part = nullptr;
  const long mpu_len = 2L * S * det * det;
  if (m_probe_update && tk_deterministic()) {
    mpu_part = tk_det_scratch(sizeof(float) * (size_t)mpu_len * nchunk);
    if (mpu_part == nullptr) return TK_ERR_ARG;
  }
  // LDS for the eigen-probe slices (only when they are applied on the fly)
  size_t eig_lds = 0;
  if (objproj && eigen_weights && eigen_probe) eig_lds = eig_bytes(MW);
  if (eig_lds > 32 * 1024) return TK_ERR_UNSUPPORTED;
#define TK_P2G(N, MW_, MPW_)                                                                 \
  do {                                                                                       \
    if (grp && !objproj)                                                                     \
      hipLaunchKernelGGL((ifft2_pass2_gradients_kernel<N, MW_, MPW_, false, true, true>),    \
                         grid, block, 0,                                                     \
                         stream, (const cf*)work, (const cf*)patches, pr, (cf*)objproj,      \
                         (cf*)chi0, (float*)m_probe_update, mpu_scale, nscan, S, inv_scale,  \
                         chunk, mpu_part, mode_scale, Stot, accumulate);                     \
    else if (grp && eig_lds > 0)                                                             \
      hipLaunchKernelGGL((ifft2_pass2_gradients_kernel<N, MW_, MPW_, true, true, true>),     \
                         grid, block, eig_lds,                                               \
                         stream, (const cf*)work, (const cf*)patches, pr, (cf*)objproj,      \
                         (cf*)chi0, (float*)m_probe_update, mpu_scale, nscan, S, inv_scale,  \
                         chunk, mpu_part, mode_scale, Stot, accumulate);                     \
    else if (grp)                                                                            \
      hipLaunchKernelGGL((ifft2_pass2_gradients_kernel<N, MW_, MPW_, true, false, true>),    \
                         grid, block, 0,                                                     \
                         stream, (const cf*)work, (const cf*)patches, pr, (cf*)objproj,      \
                         (cf*)chi0, (float*)m_probe_update, mpu_scale, nscan, S, inv_scale,  \
                         chunk, mpu_part, mode_scale, Stot, accumulate);                     \
    else if (objproj && eig_lds > 0)                                                         \
      hipLaunchKernelGGL((ifft2_pass2_gradients_kernel<N, MW_, MPW_, true>), grid, block,    \
                         eig_lds,                                                            \
                         stream, (const cf*)work, (const cf*)patches, pr, (cf*)objproj,      \
                         (cf*)chi0, (float*)m_probe_update, mpu_scale, nscan, S, inv_scale,  \
                         chunk, mpu_part, mode_scale);                                       \
    else if (objproj)                                                                        \
      hipLaunchKernelGGL((ifft2_pass2_gradients_kernel<N, MW_, MPW_, true, false>), grid,    \
                         block, 0,                                                           \
                         stream, (const cf*)work, (const cf*)patches, pr, (cf*)objproj,      \
                         (cf*)chi0, (float*)m_probe_update, mpu_scale, nscan, S, inv_scale,  \
                         chunk, mpu_part, mode_scale);                                       \
    else                                                                                     \
      hipLaunchKernelGGL((ifft2_pass2_gradients_kernel<N, MW_, MPW_, false>), grid, block,   \
                         0, stream, (const cf*)work, (const cf*)patches, pr, (cf*)objproj,   \
                         (cf*)chi0, (float*)m_probe_update, mpu_scale, nscan, S, inv_scale,  \
                         chunk, mpu_part, mode_scale);                                       \
  } while (0)
#define TK_P2G_N(N)                     \
  do {                                  \
    if (MW == 1)                        \
      TK_P2G(N < 256 ? 256 : N, 1, 1);  \
    else if (MW == 2)                   \
      TK_P2G(N, 2, 1);                  \
    else if (MPW == 1)                  \
      TK_P2G(N, 4, 1);                  \
    else                                \
      TK_P2G(N, 4, 2);                  \
  } while (0)
  switch (det) {
    case 128: TK_P2G_N(128); break;
    case 256: TK_P2G_N(256); break;
    default: TK_P2G_N(512); break;
  }
#undef TK_P2G_N
#undef TK_P2G
  TK_LAUNCH_CHECK();
  if (mpu_part != nullptr)
    return tk_ordered_sum((float*)m_probe_update, mpu_part, mpu_len, nchunk, true, stream);
  return TK_OK;
}

extern "C" int tike_ifft2_pass2_gradients(const void* work, const void* patches,
                                          const void* probe, const void* eigen_probe,
                                          const float* eigen_weights, int num_eigen,
                                          int eigen_modes, void* objproj, void* chi0,
                                          void* m_probe_update, float mpu_scale, int nscan,
                                          int S, int det, float inv_scale, void* stream) {
  TK_ENTER();
  return launch_pass2_gradients(work, patches, probe, eigen_probe, eigen_weights, num_eigen,
                                eigen_modes, objproj, chi0, m_probe_update, mpu_scale, nscan, S,
                                det, inv_scale, nullptr, (hipStream_t)stream);
}

// ... with chi_n,s also times mode_scale[n][s] (nscan,S): the poisson step
// lengths of tike_poisson_steps_grad_ifft2_pass1, known only after its pass 1
// was written.
extern "C" int tike_ifft2_pass2_gradients_scaled(const void* work, const void* patches,
                                                 const void* probe, const void* eigen_probe,
                                                 const float* eigen_weights, int num_eigen,
                                                 int eigen_modes, void* objproj, void* chi0,
                                                 void* m_probe_update, float mpu_scale,
                                                 int nscan, int S, int det, float inv_scale,
                                                 const float* mode_scale, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan == 0 || mode_scale != nullptr);
  return launch_pass2_gradients(work, patches, probe, eigen_probe, eigen_weights, num_eigen,
                                eigen_modes, objproj, chi0, m_probe_update, mpu_scale, nscan, S,
                                det, inv_scale, mode_scale, (hipStream_t)stream);
}

// 1 where the eigen probes' LDS slices of tike_ifft2_pass2_gradients fit (32 KiB
// per workgroup at the widest mode-wave split); 0: the caller keeps chi
// (tike_ifft2_crop* + tike_lstsq_gradients).  No device work.
extern "C" int tike_ifft2_pass2_eigen_fits(int det, int num_eigen, int eigen_modes) {
  if (det < 16 || num_eigen < 0 || eigen_modes < 0) return 0;
  return sizeof(cf) * (size_t)num_eigen * eigen_modes * (det / 16) * 64 <= 32 * 1024 ? 1 : 0;
}

// Modes [mode0, mode0 + nmodes) of an S-mode problem (2 <= nmodes <= 8): what
// tike_ifft2_pass2_gradients does for those modes alone -- their probe
// gradients, mode 0 of chi when mode0 == 0 -- with their share of objproj
// stored (accumulate == 0: the first group) or added to what is there.  The
// caller walks the groups in order; eigen probes must all belong to the modes
// of the first group (eigen_modes <= its nmodes).
extern "C" int tike_ifft2_pass2_gradients_modes(const void* work, const void* patches,
                                                const void* probe, const void* eigen_probe,
                                                const float* eigen_weights, int num_eigen,
                                                int eigen_modes, void* objproj, void* chi0,
                                                void* m_probe_update, float mpu_scale,
                                                int nscan, int S, int det, float inv_scale,
                                                int mode0, int nmodes, int accumulate,
                                                void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(S >= 1 && det >= 1 && mode0 >= 0 && nmodes >= 1 && mode0 + nmodes <= S);
  TK_CHECK_ARG(nscan == 0 || (work && (probe || !objproj)));
  const long P = (long)det * det;
  const bool first = mode0 == 0;
  if (first ? eigen_modes > nmodes : false) return TK_ERR_UNSUPPORTED;
  return launch_pass2_gradients(
      (const cf*)work + mode0 * P, patches, probe ? (const cf*)probe + mode0 * P : nullptr,
      first ? eigen_probe : nullptr, eigen_weights ? eigen_weights + mode0 : nullptr,
      first ? num_eigen : (eigen_weights ? num_eigen : 0), first ? eigen_modes : 0, objproj,
      first ? chi0 : nullptr,
      m_probe_update ? (void*)((float*)m_probe_update + 2 * mode0 * P) : nullptr, mpu_scale,
      nscan, nmodes, det, inv_scale, nullptr, (hipStream_t)stream, S, accumulate);
}

// The probe preconditioner with RW vertically adjacent pixels per thread: the
// RW + 1 tap rows of a position are loaded once (1.25 16-byte loads per pixel
// and position instead of 2; the sum is bound by its L1 requests).  Thread
// groups of cols = min(pw, 256) columns, 256 / cols groups stacked over the
// rows; grid.x = row blocks x column blocks, grid.y = position chunks.
template <int RW>
__global__ __launch_bounds__(256) void probe_precond_rows_kernel(
    const float* __restrict__ scan, const cf* __restrict__ psi, float* __restrict__ out,
    int nscan, int pw, int H, int W, int chunk, float* __restrict__ part) {
  typedef float tk_v4f __attribute__((ext_vector_type(4)));
  const long P = (long)pw * pw;
  const long total = (long)H * W;
  const int cols = pw < 256 ? pw : 256, ncb = pw / cols;
  const int x = ((int)blockIdx.x % ncb) * cols + (int)threadIdx.x % cols;
  const int y0 = (((int)blockIdx.x / ncb) * (256 / cols) + (int)threadIdx.x / cols) * RW;
  const int b0 = blockIdx.y * chunk;
  const int b1 = min(nscan, b0 + chunk);
  float acc[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) acc[r] = 0.f;
  bool inside = true;  // every position of the chunk interior (decided once)
  for (int b = b0; b < b1; ++b) {
    const TkCorner c = tk_corner(scan, b);
    inside = inside && c.sy >= 0 && c.sx >= 0 && c.sy + pw < H && c.sx + pw < W;
  }
  if (inside) {
    const unsigned row_bytes = (unsigned)W * (unsigned)sizeof(cf);
    const unsigned lane_off = (unsigned)y0 * row_bytes + (unsigned)x * (unsigned)sizeof(cf);
#pragma unroll 2
    for (int b = b0; b < b1; ++b) {
      const TkCorner c = tk_corner(scan, b);  // uniform
      const unsigned off = (unsigned)(c.sy * W + c.sx) * (unsigned)sizeof(cf) + lane_off;
      tk_v4f t[RW + 1];
#pragma unroll
      for (int r = 0; r <= RW; ++r)
        __builtin_memcpy(&t[r], reinterpret_cast<const char*>(psi) + off + (unsigned)r * row_bytes,
                         sizeof(tk_v4f));
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        cf o = mk(t[r].x * c.w00, t[r].y * c.w00);  // the order of tk_patch_pixel
        o.x += t[r].z * c.w01;
        o.y += t[r].w * c.w01;
        o.x += t[r + 1].x * c.w10;
        o.y += t[r + 1].y * c.w10;
        o.x += t[r + 1].z * c.w11;
        o.y += t[r + 1].w * c.w11;
        acc[r] += norm2(o);
      }
    }
  } else {
    for (int b = b0; b < b1; ++b) {
      const TkCorner c = tk_corner(scan, b);
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int y = c.sy + y0 + r, xx = c.sx + x;
        const bool ok = y >= 0 && y < H && xx >= 0 && xx < W;
        const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
        const int xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        const cf o = tk_gather(psi, (long)yc * W + xc, W, total, c);
        acc[r] += ok ? norm2(o) : 0.f;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const long p = (long)(y0 + r) * pw + x;
    if (part != nullptr)
      part[(long)blockIdx.y * P + p] = acc[r];
    else
      unsafeAtomicAdd(&out[2 * p], acc[r]);
  }
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
  float* part = nullptr;
  const int chunk = probe_chunk(nscan, P, &part);
  dim3 grid((unsigned)((P + 255) / 256), (unsigned)((nscan + chunk - 1) / chunk));
  constexpr int RW = 4;
  const int cols = pw < 256 ? pw : 256;
  if (g_stats_pairs && (pw % 256 == 0 || 256 % pw == 0) && pw % ((256 / cols) * RW) == 0 &&
      (long)H * W < (1L << 28)) {
    grid.x = (unsigned)(P / (256 * RW));
    hipLaunchKernelGGL(probe_precond_rows_kernel<RW>, grid, dim3(256), 0, (hipStream_t)stream,
                       scan, (const cf*)psi, (float*)out, nscan, pw, H, W, chunk, part);
  } else {
    launch_probe_grad<false>(grid, (hipStream_t)stream, (const cf*)nullptr, scan,
                             (const cf*)psi, (cf*)nullptr, (float*)out,
                             tk_make_probe(psi, 0, nullptr, nullptr, 0, 0, 1, pw),
                             (cf*)nullptr, nscan, 1, pw, H, W, chunk, part);
  }
  TK_LAUNCH_CHECK();
  if (part != nullptr)  // (the real parts of `out`)
    return tk_ordered_sum((float*)out, part, P, (int)grid.y, true, (hipStream_t)stream, 2);
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
// the sums of work item (position n, part `part` of nsplit) -- the body of
// step_stats_kernel, also the fall-back of step_stats_pair_kernel
template <bool HAVE_PATCHES, bool HAVE_GOBJ>
__device__ __forceinline__ void step_stats_item(
    const cf* __restrict__ chi, const float* __restrict__ scan, const cf* __restrict__ psi,
    const cf* __restrict__ gobj, const TkProbe& probe, const cf* __restrict__ mpu,
    const cf* __restrict__ patches, float* __restrict__ stats, int chi_modes, int pw, int H,
    int W, const cf* __restrict__ eigen0, float* __restrict__ eigen_proj, int nsplit, int n,
    int part, float* red) {
  const long P = (long)pw * pw;
  const long total = (long)H * W;
  const int plen = (int)(P / nsplit);
  {
    const TkCorner c = tk_corner(scan, n);
    // the varying probe of mode 0 (probe.py:272-303): weights and bases are
    // per position, hoisted here (TkProbe::at re-read them for every pixel)
    const cf* __restrict__ pbase = probe.probe + n * probe.pos_stride;
    float pw0 = 1.0f, pw1 = 0.f;
    int nE = 0;
    const float* __restrict__ wn = nullptr;
    if (probe.weights != nullptr) {
      if (probe.unique != nullptr && 0 < probe.Sm) {
        pbase = probe.unique + (long)n * probe.Sm * P;
      } else {
        wn = probe.weights + n * (long)(probe.C + 1) * probe.S;
        pw0 = wn[0];
        if (probe.eigen != nullptr && 0 < probe.Sm) {
          nE = probe.C;
          pw1 = wn[probe.S];
        }
      }
    }
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ep = 0.f;  // sum Re(conj(R_n) E_0), R_n = conj(O_n) chi_n,0 - mpu_0
    // branch-free body (clamped addresses, results zeroed by select) so that
    // the loads of two pixels are in flight together
    // (row, column) of the pixel advance with the stride instead of a
    // division per pixel
    const int qstep = (int)blockDim.x / pw, rstep = (int)blockDim.x % pw;
    const int pbeg = part * plen + (int)threadIdx.x;
    const int pend = part + 1 == nsplit ? (int)P : (part + 1) * plen;
    int py = pbeg / pw, px = pbeg % pw;
    // interior position (every tap of every pixel inside the image): the two
    // taps of a row are adjacent complex values, fetched with one 16-byte load
    const bool interior = HAVE_GOBJ && c.sy >= 0 && c.sx >= 0 && c.sy + pw < H && c.sx + pw < W &&
                          total < (1L << 28);
    // E_0 of the first eigen probe is the array the varying probe adds: one load
    const bool same_e = eigen_proj != nullptr && nE > 0 && eigen0 == probe.eigen;
    // ... and the shared probe P_0 is the base of the varying one unless a
    // synthesised array was given: one load serves both.
    const bool same_p = pbase == probe.probe;
    // The body is compiled for four cases.  A load behind a run-time
    // condition (`mpu ? mpu[p] : 0`, `same_e ? e1 : eigen0[p]`, the eigen
    // probes behind `nE > 0`) is a branch around the load, and the memory
    // latencies on either side of a branch add up: with the conditions of the
    // two common configurations settled at compile time every load of a pixel
    // is requested together (0.48 -> 0.36 ms per 1000 positions at 256^2).
    //   K = 0  any position, any configuration (clamped taps)
    //   K = 1  interior position, any configuration
    //   K = 2  interior, shared probe, no eigen probes, probe update given
    //   K = 3  interior, shared probe + ONE eigen probe whose projection is
    //          asked for (eigen0 is that eigen probe), probe update given
    auto body = [&](auto k_tag) {
      constexpr int K = decltype(k_tag)::value;
      constexpr bool FAST = K >= 1;
      constexpr bool HOT = K >= 2;
#pragma unroll 2
      for (int p = pbeg; p < pend; p += blockDim.x) {
        cf o, g;
        if (FAST) {
          typedef float tk_v4f __attribute__((ext_vector_type(4)));
          const unsigned off =
              (unsigned)((c.sy + py) * W + c.sx + px) * (unsigned)sizeof(cf);
          tk_v4f u, l;
          __builtin_memcpy(&u, reinterpret_cast<const char*>(gobj) + off, sizeof(u));
          __builtin_memcpy(&l, reinterpret_cast<const char*>(gobj) + off + (unsigned)W * 8u,
                           sizeof(l));
          if (HAVE_PATCHES) {
            o = patches[n * P + p];
          } else {
            // O_n from the object itself: the same two 16-byte tap loads at
            // the same offsets as the update's (L2) instead of 8 bytes of HBM
            tk_v4f uo, lo_;
            __builtin_memcpy(&uo, reinterpret_cast<const char*>(psi) + off, sizeof(uo));
            __builtin_memcpy(&lo_, reinterpret_cast<const char*>(psi) + off + (unsigned)W * 8u,
                             sizeof(lo_));
            o = mk(uo.x * c.w00, uo.y * c.w00);
            o.x += uo.z * c.w01;
            o.y += uo.w * c.w01;
            o.x += lo_.x * c.w10;
            o.y += lo_.y * c.w10;
            o.x += lo_.z * c.w11;
            o.y += lo_.w * c.w11;
          }
          g = mk(u.x * c.w00, u.y * c.w00);
          g.x += u.z * c.w01;
          g.y += u.w * c.w01;
          g.x += l.x * c.w10;
          g.y += l.y * c.w10;
          g.x += l.z * c.w11;
          g.y += l.w * c.w11;
        } else {
          const int y = c.sy + py, x = c.sx + px;
          const bool ok = y >= 0 && y < H && x >= 0 && x < W;
          const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
          const int xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
          const long ii = (long)yc * W + xc;
          // O_n: the patch stored by tike_lstsq_gradients when available
          o = HAVE_PATCHES ? patches[n * P + p] : tk_gather(psi, ii, W, total, c);
          g = HAVE_GOBJ ? tk_gather(gobj, ii, W, total, c) : mk(0.f, 0.f);
          if (!ok) {
            if (!HAVE_PATCHES) o = mk(0.f, 0.f);
            g = mk(0.f, 0.f);
          }
        }
        const cf x0 = chi[((long)n * chi_modes) * P + p];
        const cf p0 = probe.probe[p];
        cf pn = (HOT ? p0 : pbase[p]) * pw0;
        cf e1 = mk(0.f, 0.f);
        if (K == 3) {
          e1 = probe.eigen[p];
          pn.x += pw1 * e1.x;
          pn.y += pw1 * e1.y;
        } else if (K < 2 && nE > 0) {
          e1 = probe.eigen[p];
          pn.x += pw1 * e1.x;
          pn.y += pw1 * e1.y;
          for (int e = 1; e < nE; ++e) {  // uniform, rare
            const cf ee = probe.eigen[(long)e * probe.Sm * P + p];
            const float we = wn[(e + 1) * probe.S];
            pn.x += we * ee.x;
            pn.y += we * ee.y;
          }
        }
        const cf dOP = g * pn;
        const cf m0 = HOT ? mpu[p] : (mpu ? mpu[p] : mk(0.f, 0.f));
        const cf dPO = m0 * o;
        const cf OP = o * p0;
        a[0] += norm2(dOP);
        a[1] += norm2(dPO);
        const cf a2 = dOP * conjf(dPO);
        a[2] += a2.x;
        a[3] += a2.y;
        a[4] += dOP.x * x0.x + dOP.y * x0.y;
        a[5] += dPO.x * x0.x + dPO.y * x0.y;
        a[6] += OP.x * x0.x + OP.y * x0.y;
        a[7] += norm2(OP);
        if (K == 3) {
          const cf r = conjf(o) * x0 - m0;
          ep += r.x * e1.x + r.y * e1.y;
        } else if (K < 2 && eigen_proj) {
          const cf r = conjf(o) * x0 - m0;
          const cf e = same_e ? e1 : eigen0[p];
          ep += r.x * e.x + r.y * e.y;
        }
        py += qstep;
        px += rstep;
        if (px >= pw) {
          px -= pw;
          ++py;
        }
      }
    };
    const bool hot = interior && same_p && mpu != nullptr;
    if (hot && nE == 1 && same_e)
      body(std::integral_constant<int, 3>{});
    else if (hot && nE == 0 && eigen_proj == nullptr)
      body(std::integral_constant<int, 2>{});
    else if (interior)
      body(std::integral_constant<int, 1>{});
    else
      body(std::integral_constant<int, 0>{});
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float t = tk_block_sum256(a[k], red);
      if (threadIdx.x == 0) {
        if (nsplit > 1)
          unsafeAtomicAdd(&stats[(long)n * 8 + k], t);
        else
          stats[(long)n * 8 + k] = t;
      }
    }
    if (eigen_proj) {
      const float t = tk_block_sum256(ep, red);
      if (threadIdx.x == 0) {
        if (nsplit > 1)
          unsafeAtomicAdd(&eigen_proj[n], t);
        else
          eigen_proj[n] = t;
      }
    }
  }
}

template <bool HAVE_PATCHES, bool HAVE_GOBJ>
__global__ __launch_bounds__(256) void step_stats_kernel(
    const cf* __restrict__ chi, const float* __restrict__ scan, const cf* __restrict__ psi,
    const cf* __restrict__ gobj, const TkProbe probe, const cf* __restrict__ mpu,
    const cf* __restrict__ patches, float* __restrict__ stats, int nscan, int chi_modes, int pw,
    int H, int W, const cf* __restrict__ eigen0, float* __restrict__ eigen_proj, int nsplit) {
  __shared__ float red[4];
  // work item = (position, 1 / nsplit of its pixels): a minibatch of a few
  // hundred positions would otherwise leave most of the chip idle; with
  // nsplit > 1 the sums are accumulated into the (zeroed) tables by atomics
  for (int v = blockIdx.x; v < nscan * nsplit; v += gridDim.x)
    step_stats_item<HAVE_PATCHES, HAVE_GOBJ>(chi, scan, psi, gobj, probe, mpu, patches, stats,
                                             chi_modes, pw, H, W, eigen0, eigen_proj, nsplit,
                                             v / nsplit, v % nsplit, red);
}

// Two positions per work item (the common configurations K = 2 / K = 3 of
// step_stats_item, stored patches and the preconditioned update given): the
// shared operands of a pixel -- P_0, the probe update and the eigen probe --
// are loaded once for both positions, 11 loads per pixel pair instead of 14.
// A pair with a position on the border falls back to step_stats_item.
template <bool EIGEN>
__global__ __launch_bounds__(256) void step_stats_pair_kernel(
    const cf* __restrict__ chi, const float* __restrict__ scan, const cf* __restrict__ psi,
    const cf* __restrict__ gobj, const TkProbe probe, const cf* __restrict__ mpu,
    const cf* __restrict__ patches, float* __restrict__ stats, int nscan, int chi_modes, int pw,
    int H, int W, const cf* __restrict__ eigen0, float* __restrict__ eigen_proj, int nsplit) {
  __shared__ float red[4];
  typedef float tk_v4f __attribute__((ext_vector_type(4)));
  const long P = (long)pw * pw;
  const int plen = (int)(P / nsplit);
  const int npair = (nscan + 1) / 2;
  for (int v = blockIdx.x; v < npair * nsplit; v += gridDim.x) {
    const int n0 = 2 * (v / nsplit), part = v % nsplit;
    const TkCorner c0 = tk_corner(scan, n0);
    const bool two = n0 + 1 < nscan;
    const TkCorner c1 = tk_corner(scan, two ? n0 + 1 : n0);
    const bool in0 = c0.sy >= 0 && c0.sx >= 0 && c0.sy + pw < H && c0.sx + pw < W;
    const bool in1 = c1.sy >= 0 && c1.sx >= 0 && c1.sy + pw < H && c1.sx + pw < W;
    if (!(two && in0 && in1)) {  // uniform
      step_stats_item<true, true>(chi, scan, psi, gobj, probe, mpu, patches, stats, chi_modes,
                                  pw, H, W, eigen0, eigen_proj, nsplit, n0, part, red);
      if (two)
        step_stats_item<true, true>(chi, scan, psi, gobj, probe, mpu, patches, stats, chi_modes,
                                    pw, H, W, eigen0, eigen_proj, nsplit, n0 + 1, part, red);
      continue;
    }
    float s0 = 1.0f, s1 = 1.0f, t0 = 0.f, t1 = 0.f;  // weights of P_0 and of E_0
    if (probe.weights != nullptr) {
      const float* w = probe.weights + n0 * (long)(probe.C + 1) * probe.S;
      s0 = w[0];
      s1 = w[(long)(probe.C + 1) * probe.S];
      if (EIGEN) {
        t0 = w[probe.S];
        t1 = w[(long)(probe.C + 1) * probe.S + probe.S];
      }
    }
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ea = 0.f, eb = 0.f;
    const int qstep = (int)blockDim.x / pw, rstep = (int)blockDim.x % pw;
    const int pbeg = part * plen + (int)threadIdx.x;
    const int pend = part + 1 == nsplit ? (int)P : (part + 1) * plen;
    int py = pbeg / pw, px = pbeg % pw;
    const unsigned base0 = (unsigned)(c0.sy * W + c0.sx), base1 = (unsigned)(c1.sy * W + c1.sx);
    const cf* __restrict__ chi_a = chi + ((long)n0 * chi_modes) * P;
    const cf* __restrict__ chi_b = chi + ((long)(n0 + 1) * chi_modes) * P;
    const cf* __restrict__ pat_a = patches + (long)n0 * P;
    const cf* __restrict__ pat_b = pat_a + P;
    auto tap = [](const tk_v4f u, const tk_v4f l, const TkCorner& c) {
      cf g = mk(u.x * c.w00, u.y * c.w00);
      g.x += u.z * c.w01;
      g.y += u.w * c.w01;
      g.x += l.x * c.w10;
      g.y += l.y * c.w10;
      g.x += l.z * c.w11;
      g.y += l.w * c.w11;
      return g;
    };
    auto sums = [](float* acc, float& e_acc, const cf g, const cf o, const cf x0, const cf p0,
                   const cf pn, const cf m0, const cf e1) {
      const cf dOP = g * pn;
      const cf dPO = m0 * o;
      const cf OP = o * p0;
      acc[0] += norm2(dOP);
      acc[1] += norm2(dPO);
      const cf a2 = dOP * conjf(dPO);
      acc[2] += a2.x;
      acc[3] += a2.y;
      acc[4] += dOP.x * x0.x + dOP.y * x0.y;
      acc[5] += dPO.x * x0.x + dPO.y * x0.y;
      acc[6] += OP.x * x0.x + OP.y * x0.y;
      acc[7] += norm2(OP);
      if (EIGEN) {
        const cf r = conjf(o) * x0 - m0;
        e_acc += r.x * e1.x + r.y * e1.y;
      }
    };
    auto pixel = [&](const int p, const tk_v4f u0, const tk_v4f l0, const tk_v4f u1,
                     const tk_v4f l1) {
      const cf oa = pat_a[p], ob = pat_b[p];
      const cf xa = chi_a[p], xb = chi_b[p];
      const cf p0 = probe.probe[p];
      const cf m0 = mpu[p];
      cf e1 = mk(0.f, 0.f);
      cf pa = p0 * s0, pb = p0 * s1;
      if (EIGEN) {
        e1 = probe.eigen[p];
        pa.x += t0 * e1.x;
        pa.y += t0 * e1.y;
        pb.x += t1 * e1.x;
        pb.y += t1 * e1.y;
      }
      sums(a, ea, tap(u0, l0, c0), oa, xa, p0, pa, m0, e1);
      sums(b, eb, tap(u1, l1, c1), ob, xb, p0, pb, m0, e1);
    };
    auto taps16 = [&](const unsigned off) {
      tk_v4f t;
      __builtin_memcpy(&t, reinterpret_cast<const char*>(gobj) + off, sizeof(t));
      return t;
    };
    // Row walk: a thread keeps its column and goes down the rows of its
    // share, so the lower taps of one pixel are the upper taps of the next --
    // one 16-byte load per pixel and position instead of two.  Windows of
    // 256 k columns: the column blocks one after the other; narrower windows
    // that divide 256: 256 / pw thread groups stacked over the rows.
    const int cols = pw < 256 ? pw : 256, groups = 256 / cols;
    const bool walk = (pw % 256 == 0 || 256 % pw == 0) && pw % (nsplit * groups) == 0;
    if (walk) {
      const int rows = pw / (nsplit * groups);
      const int ybeg = (part * groups + (int)threadIdx.x / cols) * rows;
      for (int x = (int)threadIdx.x % cols; x < pw; x += 256) {
        unsigned rel = (unsigned)(ybeg * W + x);
        tk_v4f u0 = taps16((base0 + rel) * 8u), u1 = taps16((base1 + rel) * 8u);
        int p = ybeg * pw + x;
#pragma unroll 2
        for (int y = 0; y < rows; ++y) {
          rel += (unsigned)W;
          const tk_v4f l0 = taps16((base0 + rel) * 8u), l1 = taps16((base1 + rel) * 8u);
          pixel(p, u0, l0, u1, l1);
          u0 = l0;
          u1 = l1;
          p += pw;
        }
      }
    } else {
      for (int p = pbeg; p < pend; p += blockDim.x) {
        const unsigned rel = (unsigned)(py * W + px);
        const unsigned off0 = (base0 + rel) * 8u, off1 = (base1 + rel) * 8u;
        pixel(p, taps16(off0), taps16(off0 + (unsigned)W * 8u), taps16(off1),
              taps16(off1 + (unsigned)W * 8u));
        py += qstep;
        px += rstep;
        if (px >= pw) {
          px -= pw;
          ++py;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float ta = tk_block_sum256(a[k], red);
      const float tb = tk_block_sum256(b[k], red);
      if (threadIdx.x == 0) {
        if (nsplit > 1) {
          unsafeAtomicAdd(&stats[(long)n0 * 8 + k], ta);
          unsafeAtomicAdd(&stats[(long)n0 * 8 + 8 + k], tb);
        } else {
          stats[(long)n0 * 8 + k] = ta;
          stats[(long)n0 * 8 + 8 + k] = tb;
        }
      }
    }
    if (EIGEN) {
      const float ta = tk_block_sum256(ea, red);
      const float tb = tk_block_sum256(eb, red);
      if (threadIdx.x == 0) {
        if (nsplit > 1) {
          unsafeAtomicAdd(&eigen_proj[n0], ta);
          unsafeAtomicAdd(&eigen_proj[n0 + 1], tb);
        } else {
          eigen_proj[n0] = ta;
          eigen_proj[n0 + 1] = tb;
        }
      }
    }
  }
}

extern "C" int tike_lstsq_step_stats(const void* chi, const float* scan, const void* psi,
                                     const void* object_update_precond, const void* probe,
                                     const void* eigen_probe, const float* eigen_weights,
                                     int num_eigen, int eigen_modes, const void* unique_probe,
                                     const void* m_probe_update, const void* patches,
                                     float* stats, int nscan, int S, int chi_modes, int pw, int H,
                                     int W, const void* eigen0, float* eigen_proj, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && chi_modes >= 1 && pw >= 1 && H >= 1 && W >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(chi && scan && psi && probe && stats);
  TK_CHECK_ARG(!eigen_proj || (eigen0 && m_probe_update));
  const TkProbe pr = tk_make_probe(probe, 0, eigen_probe, eigen_weights, num_eigen, eigen_modes,
                                   S, pw, unique_probe);
  // split the pixels of a position over several workgroups until the launch
  // holds ~8192 of them (probe windows that are a multiple of 1024 pixels)
  // two positions per work item where the common configurations allow it
  const bool no_eigen = (eigen_probe == nullptr || eigen_modes == 0) && eigen_proj == nullptr;
  const bool one_eigen = eigen_probe != nullptr && eigen_weights != nullptr && eigen_modes > 0 &&
                         num_eigen == 1 && eigen_proj != nullptr && eigen0 == eigen_probe;
  const bool pairs = g_stats_pairs && patches && object_update_precond && m_probe_update &&
                     unique_probe == nullptr && (no_eigen || one_eigen) &&
                     (long)H * W < (1L << 28) && nscan > 1;
  const long nitem = pairs ? (nscan + 1) / 2 : nscan;
  int nsplit = 1;
  while (nsplit < 16 && nitem * nsplit * 2 <= 8192 && ((long)pw * pw) % (2048L * nsplit) == 0)
    nsplit *= 2;
  if (tk_deterministic()) nsplit = 1;  // one workgroup per position: no atomics
  if (nsplit > 1) {
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(float) * 8 * (size_t)nscan, (hipStream_t)stream);
    if (e == hipSuccess && eigen_proj)
      e = hipMemsetAsync(eigen_proj, 0, sizeof(float) * (size_t)nscan, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
#define TK_SS(HP, HG)                                                                         \
  hipLaunchKernelGGL((step_stats_kernel<HP, HG>), dim3(tk_grid((long)nscan * nsplit, 16)),    \
                     dim3(256), 0,                                                            \
                     (hipStream_t)stream, (const cf*)chi, scan, (const cf*)psi,               \
                     (const cf*)object_update_precond, pr, (const cf*)m_probe_update,         \
                     (const cf*)patches, stats, nscan, chi_modes, pw, H, W, (const cf*)eigen0, \
                     eigen_proj, nsplit)
#define TK_SP(EIG)                                                                            \
  hipLaunchKernelGGL((step_stats_pair_kernel<EIG>), dim3(tk_grid(nitem * nsplit, 16)),        \
                     dim3(256), 0,                                                            \
                     (hipStream_t)stream, (const cf*)chi, scan, (const cf*)psi,               \
                     (const cf*)object_update_precond, pr, (const cf*)m_probe_update,         \
                     (const cf*)patches, stats, nscan, chi_modes, pw, H, W, (const cf*)eigen0, \
                     eigen_proj, nsplit)
  if (pairs && one_eigen) TK_SP(true);
  else if (pairs) TK_SP(false);
  else if (patches && object_update_precond) TK_SS(true, true);
  else if (patches) TK_SS(true, false);
  else if (object_update_precond) TK_SS(false, true);
  else TK_SS(false, false);
#undef TK_SS
#undef TK_SP
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------------------- eigen-probe update
// Residual probe update of position n for eigen probe index c (mode 0):
//   R_n = conj(O_n) chi_n,0 - mpu_0 - sum_{c' < c} coef[n][c'] E_c'
// (lstsq.py:740-761 _get_residuals/_update_residuals; probe.py:362-476).
// O_n = patches[n], chi_n,0 = chi0[n]; eigen (C, Sm, pw, pw) holds the CURRENT
// eigen probes (mode 0 slice used), coefs (nscan, C) complex the projections
// already removed.  Never materialised: every pass recomputes it.
struct TkResidual {
  const cf* patches;
  const cf* chi0;
  const cf* mpu0;
  const cf* eigen;
  const cf* coefs;
  int C, Sm, c;
  long P;
  long XS;  // elements between chi0 of consecutive positions (chi_modes * P)
  // C0: the residual of the FIRST eigen probe (c == 0) has no projections to
  // remove -- without that run-time loop in the body the callers' loops unroll
  // and the loads of several pixels / positions are in flight together
  template <bool C0>
  __device__ __forceinline__ cf at(long n, long p) const {
    cf r = conjf(patches[n * P + p]) * chi0[n * XS + p] - mpu0[p];
    if (!C0)
      for (int k = 0; k < c; ++k) r = r - coefs[n * C + k] * eigen[((long)k * Sm) * P + p];
    return r;
  }
};

// sums[n] = { sum Re(conj(R) E_c), sum Re(chi0 conj(O E_c)), sum |O E_c|^2,
//             Re sum R conj(E_c), Im sum R conj(E_c) }
template <bool C0>
__global__ __launch_bounds__(256) void eigen_position_sums_kernel(const TkResidual R,
                                                                  float* __restrict__ sums,
                                                                  int nscan) {
  __shared__ float red[4];
  const cf* __restrict__ E = R.eigen + ((long)R.c * R.Sm) * R.P;
  for (int n = blockIdx.x; n < nscan; n += gridDim.x) {
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (long p = threadIdx.x; p < R.P; p += blockDim.x) {  // (unrolled by 4: 0.217 -> 0.234 ms)
      const cf e = E[p];
      const cf r = R.at<C0>(n, p);
      const cf phi = R.patches[n * R.P + p] * e;
      const cf x = R.chi0[n * R.XS + p];
      a[0] += r.x * e.x + r.y * e.y;
      a[1] += x.x * phi.x + x.y * phi.y;
      a[2] += norm2(phi);
      const cf re = r * conjf(e);
      a[3] += re.x;
      a[4] += re.y;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float v = tk_block_sum256(a[k], red);
      if (threadIdx.x == 0) sums[(long)n * 5 + k] = v;
    }
  }
}

// update[p] += sum_n R_n[p] * pm[n]      (probe.py:432-436 before the mean)
template <bool C0>
__global__ __launch_bounds__(256) void eigen_pixel_update_kernel(const TkResidual R,
                                                                 const float* __restrict__ pm,
                                                                 float* __restrict__ update,
                                                                 int nscan, int chunk,
                                                                 float* __restrict__ part) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= R.P) return;
  const int b0 = blockIdx.y * chunk;
  const int b1 = min(nscan, b0 + chunk);
  cf acc = mk(0.f, 0.f);
#pragma unroll 4
  for (int n = b0; n < b1; ++n) {
    const cf r = R.at<C0>(n, p);
    const float w = pm[n];
    acc.x += r.x * w;
    acc.y += r.y * w;
  }
  if (part != nullptr) {  // deterministic mode: see probe_grad_kernel
    part[2 * ((long)blockIdx.y * R.P + p)] = acc.x;
    part[2 * ((long)blockIdx.y * R.P + p) + 1] = acc.y;
    return;
  }
  unsafeAtomicAdd(&update[2 * p], acc.x);
  unsafeAtomicAdd(&update[2 * p + 1], acc.y);
}

// O_n = patch_n(psi) at one pixel of an INTERIOR position, recomputed from the
// object (L2-resident: neighbouring positions share most of their footprint)
// exactly as forward pass 1 formed the stored patch -- same taps, same order
// of operations -- instead of streaming the stored patch from HBM.
// off: byte offset of the pixel's upper-left tap inside psi.
__device__ __forceinline__ cf tk_patch_pixel(const cf* __restrict__ psi, unsigned off,
                                             unsigned row_bytes, const TkCorner& c) {
  typedef float tk_v4f __attribute__((ext_vector_type(4)));
  tk_v4f u, l;
  __builtin_memcpy(&u, reinterpret_cast<const char*>(psi) + off, sizeof(u));
  __builtin_memcpy(&l, reinterpret_cast<const char*>(psi) + off + row_bytes, sizeof(l));
  cf o = mk(u.x * c.w00, u.y * c.w00);
  o.x += u.z * c.w01;
  o.y += u.w * c.w01;
  o.x += l.x * c.w10;
  o.y += l.y * c.w10;
  o.x += l.z * c.w11;
  o.y += l.w * c.w11;
  return o;
}
__device__ __forceinline__ bool tk_interior(const TkCorner& c, int pw, int H, int W) {
  return c.sy >= 0 && c.sx >= 0 && c.sy + pw < H && c.sx + pw < W && (long)H * W < (1L << 28);
}

// The packed tail (one eigen probe per mode, c = 0): the per-position factor
// pm[n] = (eproj[n] / P + w[n]) / norm (probe.py:429-433) is formed on the fly
// from the projection the step statistics left and the batch norm.
// The one workgroup past the pixel blocks (blockIdx.x == gridDim.x - 1,
// blockIdx.y == 0) forms sums3 = { sum(A1 + eps), sum(A4 + eps), sum(costs) }
// of the step-statistics table (tike_lstsq_step_sums) -- the two go into one
// all-reduce.
__global__ __launch_bounds__(256) void eigen_pixel_update1_kernel(
    const TkResidual R, const float* __restrict__ eproj, const float* __restrict__ weights_c,
    long row, const float* __restrict__ norm, float inv_P, float* __restrict__ update, int nscan,
    int chunk, const float* __restrict__ stats, const float* __restrict__ costs, float eps,
    float* __restrict__ sums3, const cf* __restrict__ psi, const float* __restrict__ scan, int pw,
    int H, int W, float* __restrict__ part) {
  if (blockIdx.x + 1 == gridDim.x) {
    if (blockIdx.y != 0 || sums3 == nullptr) return;
    __shared__ float red[4];
    float a1 = 0.f, a4 = 0.f, c = 0.f;
    for (int n = threadIdx.x; n < nscan; n += 256) {
      a1 += stats[8 * n] + eps;
      a4 += stats[8 * n + 1] + eps;
      c += costs[n];
    }
    a1 = tk_block_sum256(a1, red);
    a4 = tk_block_sum256(a4, red);
    c = tk_block_sum256(c, red);
    if (threadIdx.x == 0) {
      sums3[0] = a1;
      sums3[1] = a4;
      sums3[2] = c;
    }
    return;
  }
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= R.P) return;
  const int b0 = blockIdx.y * chunk;
  const int b1 = min(nscan, b0 + chunk);
  const float inv_norm = 1.0f / norm[0];
  cf acc = mk(0.f, 0.f);
  // O_n recomputed from the object when every position of the chunk is
  // interior (decided once: no load sits behind a per-position branch)
  bool gather = psi != nullptr;
  if (gather) {
    for (int n = b0; n < b1; ++n) gather = gather && tk_interior(tk_corner(scan, n), pw, H, W);
  }
  if (gather) {
    const int py = (int)(p / pw), px = (int)(p % pw);
    const unsigned row_bytes = (unsigned)W * (unsigned)sizeof(cf);
    const unsigned lane_off = (unsigned)py * row_bytes + (unsigned)px * (unsigned)sizeof(cf);
    const cf m0 = R.mpu0[p];
#pragma unroll 4
    for (int n = b0; n < b1; ++n) {
      const TkCorner c = tk_corner(scan, n);  // uniform
      const cf x = R.chi0[n * R.XS + p];
      const cf o = tk_patch_pixel(
          psi, (unsigned)(c.sy * W + c.sx) * (unsigned)sizeof(cf) + lane_off, row_bytes, c);
      const cf r = conjf(o) * x - m0;
      const float w = (eproj[n] * inv_P + weights_c[n * row]) * inv_norm;
      acc.x += r.x * w;
      acc.y += r.y * w;
    }
  } else {
#pragma unroll 4
    for (int n = b0; n < b1; ++n) {
      const cf r = R.at<true>(n, p);
      const float w = (eproj[n] * inv_P + weights_c[n * row]) * inv_norm;
      acc.x += r.x * w;
      acc.y += r.y * w;
    }
  }
  if (part != nullptr) {  // deterministic mode: see probe_grad_kernel
    part[2 * ((long)blockIdx.y * R.P + p)] = acc.x;
    part[2 * ((long)blockIdx.y * R.P + p) + 1] = acc.y;
    return;
  }
  unsafeAtomicAdd(&update[2 * p], acc.x);
  unsafeAtomicAdd(&update[2 * p + 1], acc.y);
}

// out[0] += scale * sum_n table[n * stride + col]: one workgroup, thread t takes
// rows t, t + 256, ..., then the fixed tree of tk_block_sum256
__global__ __launch_bounds__(256) void column_sum_ordered_kernel(const float* __restrict__ table,
                                                                 int stride, int col, int n,
                                                                 float scale,
                                                                 float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += table[(long)i * stride + col] * scale;
  a = tk_block_sum256(a, red);
  if (threadIdx.x == 0) out[0] += a;
}

// Position sums against the (already updated) first eigen probe, plus
// dsum[0] += sum_n sums[n][2] / P (the denominator mean, probe.py:463-469).
__global__ __launch_bounds__(256) void eigen_position_sums1_kernel(
    const TkResidual R, float* __restrict__ sums, float* __restrict__ dsum, int nscan,
    const cf* __restrict__ psi, const float* __restrict__ scan, int pw, int H, int W) {
  __shared__ float red[4];
  const cf* __restrict__ E = R.eigen;
  const unsigned row_bytes = (unsigned)W * (unsigned)sizeof(cf);
  const int qstep = (int)blockDim.x / pw, rstep = (int)blockDim.x % pw;
  for (int n = blockIdx.x; n < nscan; n += gridDim.x) {
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    TkCorner c = {0, 0, 0.f, 0.f, 0.f, 0.f};
    if (psi != nullptr) c = tk_corner(scan, n);  // uniform
    const bool gather = psi != nullptr && tk_interior(c, pw, H, W);
    const unsigned off0 = gather ? (unsigned)(c.sy * W + c.sx) * (unsigned)sizeof(cf) : 0u;
    // (the choice is settled outside the pixel loop: a load behind a run-time
    // condition would wait for the loads in front of the branch)
    auto body = [&](auto g_tag) {
      constexpr bool G = decltype(g_tag)::value;
      int py = (int)threadIdx.x / pw, px = (int)threadIdx.x % pw;
      for (long p = threadIdx.x; p < R.P; p += blockDim.x) {
        const cf e = E[p];
        const cf x = R.chi0[n * R.XS + p];
        const cf m0 = R.mpu0[p];
        cf o;
        if constexpr (G)
          o = tk_patch_pixel(psi, off0 + (unsigned)py * row_bytes + (unsigned)px * 8u, row_bytes, c);
        else
          o = R.patches[n * R.P + p];
        const cf r = conjf(o) * x - m0;
        const cf phi = o * e;
        py += qstep;
        px += rstep;
        if (px >= pw) {
          px -= pw;
          ++py;
        }
        a[0] += r.x * e.x + r.y * e.y;
        a[1] += x.x * phi.x + x.y * phi.y;
        a[2] += norm2(phi);
        const cf re = r * conjf(e);
        a[3] += re.x;
        a[4] += re.y;
      }
    };
    if (gather)
      body(std::true_type{});
    else
      body(std::false_type{});
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float v = tk_block_sum256(a[k], red);
      if (threadIdx.x == 0) {
        sums[(long)n * 5 + k] = v;
        if (k == 2 && dsum != nullptr) unsafeAtomicAdd(dsum, v / (float)R.P);
      }
    }
  }
}

// Two positions per workgroup of 512 threads, row walk (see
// step_stats_pair_kernel): E_0 and the probe update are loaded once for both
// positions and every pixel costs one 16-byte tap load per position -- 6 loads
// per pixel pair instead of 10.  cols = min(pw, 256) columns per thread group,
// 512 / cols groups stacked over the rows; a pair with a position on the
// border (or the odd last position) takes the strided per-position loop.
__global__ __launch_bounds__(512) void eigen_position_sums1_pair_kernel(
    const TkResidual R, float* __restrict__ sums, float* __restrict__ dsum, int nscan,
    const cf* __restrict__ psi, const float* __restrict__ scan, int pw, int H, int W) {
  __shared__ float red[8];
  typedef float tk_v4f __attribute__((ext_vector_type(4)));
  const cf* __restrict__ E = R.eigen;
  const unsigned row_bytes = (unsigned)W * (unsigned)sizeof(cf);
  const int cols = pw < 256 ? pw : 256, rows = pw / (512 / cols);
  const int npair = (nscan + 1) / 2;
  auto add = [](float* a, const cf o, const cf x, const cf e, const cf m0) {
    const cf r = conjf(o) * x - m0;
    const cf phi = o * e;
    a[0] += r.x * e.x + r.y * e.y;
    a[1] += x.x * phi.x + x.y * phi.y;
    a[2] += norm2(phi);
    const cf re = r * conjf(e);
    a[3] += re.x;
    a[4] += re.y;
  };
  auto finish = [&](const float* a, const int n) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float v = tk_block_sum512(a[k], red);
      if (threadIdx.x == 0) {
        sums[(long)n * 5 + k] = v;
        if (k == 2 && dsum != nullptr) unsafeAtomicAdd(dsum, v / (float)R.P);
      }
    }
  };
  auto one = [&](const int n, const TkCorner& c, const bool gather) {
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const unsigned off0 = gather ? (unsigned)(c.sy * W + c.sx) * 8u : 0u;
    for (long p = threadIdx.x; p < R.P; p += blockDim.x) {
      const cf o = gather ? tk_patch_pixel(psi, off0 + (unsigned)(p / pw) * row_bytes +
                                                    (unsigned)(p % pw) * 8u, row_bytes, c)
                          : R.patches[n * R.P + p];
      add(a, o, R.chi0[n * R.XS + p], E[p], R.mpu0[p]);
    }
    finish(a, n);
  };
  auto taps16 = [&](const unsigned off) {
    tk_v4f t;
    __builtin_memcpy(&t, reinterpret_cast<const char*>(psi) + off, sizeof(t));
    return t;
  };
  auto tap = [](const tk_v4f u, const tk_v4f l, const TkCorner& c) {
    cf o = mk(u.x * c.w00, u.y * c.w00);  // the order of tk_patch_pixel
    o.x += u.z * c.w01;
    o.y += u.w * c.w01;
    o.x += l.x * c.w10;
    o.y += l.y * c.w10;
    o.x += l.z * c.w11;
    o.y += l.w * c.w11;
    return o;
  };
  for (int pair = blockIdx.x; pair < npair; pair += gridDim.x) {
    const int n0 = 2 * pair;
    const bool two = n0 + 1 < nscan;
    const TkCorner c0 = tk_corner(scan, n0);
    const TkCorner c1 = tk_corner(scan, two ? n0 + 1 : n0);
    const bool in0 = tk_interior(c0, pw, H, W), in1 = tk_interior(c1, pw, H, W);
    if (!(two && in0 && in1)) {  // uniform
      one(n0, c0, in0);
      if (two) one(n0 + 1, c1, in1);
      continue;
    }
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, b[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const unsigned base0 = (unsigned)(c0.sy * W + c0.sx), base1 = (unsigned)(c1.sy * W + c1.sx);
    const cf* __restrict__ chi_a = R.chi0 + (long)n0 * R.XS;
    const cf* __restrict__ chi_b = chi_a + R.XS;
    const int ybeg = ((int)threadIdx.x / cols) * rows;
    for (int x = (int)threadIdx.x % cols; x < pw; x += 256) {
      unsigned rel = (unsigned)(ybeg * W + x);
      tk_v4f u0 = taps16((base0 + rel) * 8u), u1 = taps16((base1 + rel) * 8u);
      int p = ybeg * pw + x;
#pragma unroll 2
      for (int y = 0; y < rows; ++y) {
        rel += (unsigned)W;
        const tk_v4f l0 = taps16((base0 + rel) * 8u), l1 = taps16((base1 + rel) * 8u);
        const cf e = E[p], m0 = R.mpu0[p];
        const cf xa = chi_a[p], xb = chi_b[p];
        add(a, tap(u0, l0, c0), xa, e, m0);
        add(b, tap(u1, l1, c1), xb, e, m0);
        u0 = l0;
        u1 = l1;
        p += pw;
      }
    }
    finish(a, n0);
    finish(b, n0 + 1);
  }
}

static TkResidual make_residual(const void* patches, const void* chi0, const void* mpu0,
                                const void* eigen, const void* coefs, int C, int Sm, int c,
                                int pw, int chi_modes) {
  TkResidual R;
  R.patches = (const cf*)patches;
  R.chi0 = (const cf*)chi0;
  R.mpu0 = (const cf*)mpu0;
  R.eigen = (const cf*)eigen;
  R.coefs = (const cf*)coefs;
  R.C = C;
  R.Sm = Sm;
  R.c = c;
  R.P = (long)pw * pw;
  R.XS = R.P * chi_modes;
  return R;
}

extern "C" int tike_eigen_position_sums(const void* patches, const void* chi0, const void* mpu0,
                                        const void* eigen_probe, const void* coefs,
                                        int num_eigen, int eigen_modes, int c, float* sums,
                                        int nscan, int pw, int chi_modes, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && pw >= 1 && num_eigen >= 1 && c >= 0 && c < num_eigen &&
               chi_modes >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(patches && chi0 && mpu0 && eigen_probe && sums && (c == 0 || coefs));
  const TkResidual R = make_residual(patches, chi0, mpu0, eigen_probe, coefs, num_eigen,
                                     eigen_modes, c, pw, chi_modes);
  if (c == 0)
    hipLaunchKernelGGL(eigen_position_sums_kernel<true>, dim3(tk_grid(nscan, 16)), dim3(256), 0,
                       (hipStream_t)stream, R, sums, nscan);
  else
    hipLaunchKernelGGL(eigen_position_sums_kernel<false>, dim3(tk_grid(nscan, 16)), dim3(256),
                       0, (hipStream_t)stream, R, sums, nscan);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_eigen_pixel_update(const void* patches, const void* chi0, const void* mpu0,
                                       const void* eigen_probe, const void* coefs,
                                       int num_eigen, int eigen_modes, int c, const float* pm,
                                       void* update, int nscan, int pw, int chi_modes,
                                       void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && pw >= 1 && num_eigen >= 1 && c >= 0 && c < num_eigen &&
               chi_modes >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(patches && chi0 && mpu0 && eigen_probe && pm && update && (c == 0 || coefs));
  const long P = (long)pw * pw;
  float* part = nullptr;
  const int chunk = probe_chunk(nscan, 2 * P, &part);
  dim3 grid((unsigned)((P + 255) / 256), (unsigned)((nscan + chunk - 1) / chunk));
  const TkResidual R = make_residual(patches, chi0, mpu0, eigen_probe, coefs, num_eigen,
                                     eigen_modes, c, pw, chi_modes);
  if (c == 0)
    hipLaunchKernelGGL(eigen_pixel_update_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream,
                       R, pm, (float*)update, nscan, chunk, part);
  else
    hipLaunchKernelGGL(eigen_pixel_update_kernel<false>, grid, dim3(256), 0,
                       (hipStream_t)stream, R, pm, (float*)update, nscan, chunk, part);
  TK_LAUNCH_CHECK();
  if (part != nullptr)
    return tk_ordered_sum((float*)update, part, 2 * P, (int)grid.y, true, (hipStream_t)stream);
  return TK_OK;
}

extern "C" int tike_eigen_pixel_update1(const void* patches, const void* chi0,
                                        const void* mpu0, const void* eigen0,
                                        const float* eigen_proj, const float* weights_c,
                                        long weights_row, const float* norm, void* update,
                                        int nscan, int pw, int chi_modes, const float* stats,
                                        const float* costs, float eps, float* sums3,
                                        const void* psi, const float* scan, int H, int W,
                                        void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && pw >= 1 && chi_modes >= 1 && weights_row >= 1);
  // (an EMPTY share of a minibatch -- more ranks than positions -- comes with
  // null per-position arrays: nothing to check, the sums are zero)
  if (nscan == 0) {
    if (sums3) return (int)hipMemsetAsync(sums3, 0, 3 * sizeof(float), (hipStream_t)stream);
    return TK_OK;
  }
  TK_CHECK_ARG(!sums3 || (stats && costs));
  TK_CHECK_ARG(!psi || (scan && H >= 1 && W >= 1));
  TK_CHECK_ARG(patches && chi0 && mpu0 && eigen0 && eigen_proj && weights_c && norm && update);
  const long P = (long)pw * pw;
  float* part = nullptr;
  const int chunk = probe_chunk(nscan, 2 * P, &part);
  dim3 grid((unsigned)((P + 255) / 256) + 1, (unsigned)((nscan + chunk - 1) / chunk));
  const TkResidual R = make_residual(patches, chi0, mpu0, eigen0, nullptr, 1, 1, 0, pw, chi_modes);
  hipLaunchKernelGGL(eigen_pixel_update1_kernel, grid, dim3(256), 0, (hipStream_t)stream, R,
                     eigen_proj, weights_c, weights_row, norm, 1.0f / (float)P, (float*)update,
                     nscan, chunk, stats, costs, eps, sums3, (const cf*)psi, scan, pw, H, W,
                     part);
  TK_LAUNCH_CHECK();
  if (part != nullptr)
    return tk_ordered_sum((float*)update, part, 2 * P, (int)grid.y, true, (hipStream_t)stream);
  return TK_OK;
}

extern "C" int tike_eigen_position_sums1(const void* patches, const void* chi0,
                                         const void* mpu0, const void* eigen0, float* sums,
                                         float* dsum, int nscan, int pw, int chi_modes,
                                         const void* psi, const float* scan, int H, int W,
                                         void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && pw >= 1 && chi_modes >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(!psi || (scan && H >= 1 && W >= 1));
  TK_CHECK_ARG(patches && chi0 && mpu0 && eigen0 && sums && dsum);
  const TkResidual R = make_residual(patches, chi0, mpu0, eigen0, nullptr, 1, 1, 0, pw, chi_modes);
  const bool det = tk_deterministic();
  const int cols = pw < 256 ? pw : 256;
  const bool pairs = g_stats_pairs && psi != nullptr && nscan > 1 &&
                     (pw % 256 == 0 || 256 % pw == 0) && pw % (512 / cols) == 0 &&
                     (long)H * W < (1L << 28);
  if (pairs)
    hipLaunchKernelGGL(eigen_position_sums1_pair_kernel, dim3(tk_grid((nscan + 1) / 2, 16)),
                       dim3(512), 0, (hipStream_t)stream, R, sums, det ? nullptr : dsum, nscan,
                       (const cf*)psi, scan, pw, H, W);
  else
    hipLaunchKernelGGL(eigen_position_sums1_kernel, dim3(tk_grid(nscan, 16)), dim3(256), 0,
                       (hipStream_t)stream, R, sums, det ? nullptr : dsum, nscan, (const cf*)psi,
                       scan, pw, H, W);
  if (det)  // dsum += sum_n sums[n][2] / P, one workgroup, a fixed order
    hipLaunchKernelGGL(column_sum_ordered_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums,
                       5, 2, nscan, 1.0f / (float)((long)pw * pw), dsum);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------------------- varying probe
// out[n][s] = weights[n][0][s] * probe[s] + sum_c weights[n][c+1][s] * eigen[c][s]
// for the first Sm modes (probe.py:272-303 get_varying_probe); the modes
// without eigen probes only need the scalar weights[n][0][s].
__global__ __launch_bounds__(256) void varying_probe_kernel(const TkProbe probe,
                                                            cf* __restrict__ out, long total,
                                                            long PP) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long pix = i % PP;
    const long ns = i / PP;
    out[i] = probe.at(ns / probe.Sm, (int)(ns % probe.Sm), pix);
  }
}

extern "C" int tike_varying_probe(const void* probe, const void* eigen_probe,
                                  const float* eigen_weights, int num_eigen, int eigen_modes,
                                  void* out, int nscan, int S, int pw, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 1 && eigen_modes >= 1 && eigen_modes <= S);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(probe && eigen_weights && out && (num_eigen == 0 || eigen_probe));
  const long PP = (long)pw * pw;
  const long total = (long)nscan * eigen_modes * PP;
  hipLaunchKernelGGL(varying_probe_kernel, dim3(tk_grid((total + 255) / 256, 16)), dim3(256), 0,
                     (hipStream_t)stream,
                     tk_make_probe(probe, 0, eigen_probe, eigen_weights, num_eigen, eigen_modes,
                                   S, pw),
                     (cf*)out, total, PP);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------------------- position correction
// lstsq.py:545-579.  Per position n (mode m = 0, central window [crop, pw-crop)):
//   gx = gaussian derivative of the object patch along rows, gy along columns
//        (position.py:779-810: scipy gaussian_filter1d(-x, order=1, mode
//        'nearest'); the taps come precomputed from the host),
//   num[n] = ( sum Re(conj(gx P) chi), sum Re(conj(gy P) chi) ),
//   den[n] = ( sum |gx P|^2,           sum |gy P|^2 ),    P = probe_n mode 0.
// One workgroup per position.
struct TkTaps {
  float t[9];
  int r;
};

template <int RT>  // tap radius at compile time (-1: taps.r), so that the 2 (2 r + 1) tap
                   // loads of a pixel are requested together with its probe and chi values
__global__ __launch_bounds__(256) void position_sums_kernel(
    const cf* __restrict__ patches, const cf* __restrict__ chi, int chi_modes,
    const TkProbe probe, const TkTaps taps, float* __restrict__ num, float* __restrict__ den,
    int pw, int nsplit) {
  __shared__ float red[4];
  // work item = (position, 1 / nsplit of the window): see step_stats_kernel
  const long n = blockIdx.x / nsplit;
  const int part = blockIdx.x % nsplit;
  const long P = (long)pw * pw;
  const cf* __restrict__ O = patches + n * P;
  const cf* __restrict__ X = chi + n * chi_modes * P;
  const int crop = pw / 4;
  const int w = pw - 2 * crop;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  const int ilen = (w * w + nsplit - 1) / nsplit;
  const int iend = min(w * w, (part + 1) * ilen);
  for (int i = part * ilen + threadIdx.x; i < iend; i += blockDim.x) {
    const int y = crop + i / w, x = crop + i % w;
    cf gx = mk(0.f, 0.f), gy = mk(0.f, 0.f);
    const long pix = (long)y * pw + x;
    const cf Pm = probe.at(n, 0, pix);
    const cf c = X[pix];
    auto tap = [&](int d, int r) {
      const float t = taps.t[d + r];
      int yy = y + d, xx = x + d;
      yy = yy < 0 ? 0 : (yy >= pw ? pw - 1 : yy);
      xx = xx < 0 ? 0 : (xx >= pw ? pw - 1 : xx);
      const cf oy = O[yy * pw + x], ox = O[y * pw + xx];
      gx.x += t * oy.x;
      gx.y += t * oy.y;
      gy.x += t * ox.x;
      gy.y += t * ox.y;
    };
    if (RT >= 0) {
#pragma unroll
      for (int d = -RT; d <= RT; ++d) tap(d, RT);
    } else {
      for (int d = -taps.r; d <= taps.r; ++d) tap(d, taps.r);
    }
    const cf px = gx * Pm, py = gy * Pm;
    a[0] += px.x * c.x + px.y * c.y;
    a[1] += py.x * c.x + py.y * c.y;
    a[2] += norm2(px);
    a[3] += norm2(py);
  }
  for (int k = 0; k < 4; ++k) a[k] = tk_block_sum256(a[k], red);
  if (threadIdx.x == 0) {
    if (nsplit > 1) {
      unsafeAtomicAdd(&num[2 * n], a[0]);
      unsafeAtomicAdd(&num[2 * n + 1], a[1]);
      unsafeAtomicAdd(&den[2 * n], a[2]);
      unsafeAtomicAdd(&den[2 * n + 1], a[3]);
    } else {
      num[2 * n] = a[0];
      num[2 * n + 1] = a[1];
      den[2 * n] = a[2];
      den[2 * n + 1] = a[3];
    }
  }
}

// Radius 2, two positions per work item, row walk (see step_stats_pair_kernel):
// a thread keeps its column of the central window and goes down the rows of
// its share with the five vertical taps of each position in registers -- one
// new 8-byte load per pixel instead of five -- the four horizontal neighbours
// come as two 16-byte loads, and a shared probe is loaded once for the pair:
// 9 loads per pixel pair instead of 24.  cols = min(w, 256) columns per thread
// group, 256 / cols groups stacked over the rows.
__global__ __launch_bounds__(256) void position_sums_pair_kernel(
    const cf* __restrict__ patches, const cf* __restrict__ chi, int chi_modes,
    const TkProbe probe, const TkTaps taps, float* __restrict__ num, float* __restrict__ den,
    int pw, int nscan, int nsplit) {
  __shared__ float red[4];
  typedef float tk_v4f __attribute__((ext_vector_type(4)));
  const long P = (long)pw * pw;
  const int crop = pw / 4;
  const int w = pw - 2 * crop;
  const int cols = w < 256 ? w : 256, groups = 256 / cols, rows = w / (nsplit * groups);
  const int npair = (nscan + 1) / 2;
  const bool shared = probe.weights == nullptr && probe.pos_stride == 0;
  const float t0 = taps.t[0], t1 = taps.t[1], t2 = taps.t[2], t3 = taps.t[3], t4 = taps.t[4];
  auto ld16 = [](const cf* p) {
    tk_v4f v;
    __builtin_memcpy(&v, p, sizeof(v));
    return v;
  };
  auto add = [&](float* a, const cf (&v)[5], const tk_v4f hl, const tk_v4f hr, const cf Pm,
                 const cf c) {
    cf gx = mk(t0 * v[0].x, t0 * v[0].y), gy = mk(t0 * hl.x, t0 * hl.y);
    gx.x += t1 * v[1].x;
    gx.y += t1 * v[1].y;
    gy.x += t1 * hl.z;
    gy.y += t1 * hl.w;
    gx.x += t2 * v[2].x;
    gx.y += t2 * v[2].y;
    gy.x += t2 * v[2].x;
    gy.y += t2 * v[2].y;
    gx.x += t3 * v[3].x;
    gx.y += t3 * v[3].y;
    gy.x += t3 * hr.x;
    gy.y += t3 * hr.y;
    gx.x += t4 * v[4].x;
    gx.y += t4 * v[4].y;
    gy.x += t4 * hr.z;
    gy.y += t4 * hr.w;
    const cf px = gx * Pm, py = gy * Pm;
    a[0] += px.x * c.x + px.y * c.y;
    a[1] += py.x * c.x + py.y * c.y;
    a[2] += norm2(px);
    a[3] += norm2(py);
  };
  for (int item = blockIdx.x; item < npair * nsplit; item += gridDim.x) {
    const long n0 = 2 * (item / nsplit);
    const int part = item % nsplit;
    const bool two = n0 + 1 < nscan;
    const long n1 = two ? n0 + 1 : n0;
    const cf* __restrict__ O0 = patches + n0 * P;
    const cf* __restrict__ O1 = patches + n1 * P;
    const cf* __restrict__ X0 = chi + n0 * chi_modes * P;
    const cf* __restrict__ X1 = chi + n1 * chi_modes * P;
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    const int ybeg = crop + (part * groups + (int)threadIdx.x / cols) * rows;
    for (int x = crop + (int)threadIdx.x % cols; x < crop + w; x += 256) {
      cf u[5], v[5];  // rows y - 2 .. y + 2 of column x, positions n0 / n1
#pragma unroll
      for (int d = 1; d < 5; ++d) {
        u[d] = O0[(ybeg - 3 + d) * pw + x];
        v[d] = O1[(ybeg - 3 + d) * pw + x];
      }
      for (int y = ybeg; y < ybeg + rows; ++y) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          u[d] = u[d + 1];
          v[d] = v[d + 1];
        }
        const int pix = y * pw + x;
        u[4] = O0[pix + 2 * pw];
        v[4] = O1[pix + 2 * pw];
        const tk_v4f ul = ld16(O0 + pix - 2), ur = ld16(O0 + pix + 1);
        const tk_v4f vl = ld16(O1 + pix - 2), vr = ld16(O1 + pix + 1);
        const cf c0 = X0[pix], c1 = X1[pix];
        const cf P0 = probe.at(n0, 0, pix);
        const cf P1 = shared ? P0 : probe.at(n1, 0, pix);
        add(a, u, ul, ur, P0, c0);
        add(b, v, vl, vr, P1, c1);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] = tk_block_sum256(a[k], red);
      b[k] = tk_block_sum256(b[k], red);
    }
    if (threadIdx.x == 0) {
      if (nsplit > 1) {
        unsafeAtomicAdd(&num[2 * n0], a[0]);
        unsafeAtomicAdd(&num[2 * n0 + 1], a[1]);
        unsafeAtomicAdd(&den[2 * n0], a[2]);
        unsafeAtomicAdd(&den[2 * n0 + 1], a[3]);
        if (two) {
          unsafeAtomicAdd(&num[2 * n1], b[0]);
          unsafeAtomicAdd(&num[2 * n1 + 1], b[1]);
          unsafeAtomicAdd(&den[2 * n1], b[2]);
          unsafeAtomicAdd(&den[2 * n1 + 1], b[3]);
        }
      } else {
        num[2 * n0] = a[0];
        num[2 * n0 + 1] = a[1];
        den[2 * n0] = a[2];
        den[2 * n0 + 1] = a[3];
        if (two) {
          num[2 * n1] = b[0];
          num[2 * n1 + 1] = b[1];
          den[2 * n1] = b[2];
          den[2 * n1 + 1] = b[3];
        }
      }
    }
  }
}

extern "C" int tike_position_sums(const void* patches, const void* chi, int chi_modes,
                                  const void* probe, const void* eigen_probe,
                                  const float* eigen_weights, int num_eigen, int eigen_modes,
                                  const float* taps_host, int radius, float* numerator,
                                  float* denominator, int nscan, int S, int pw, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 4 && chi_modes >= 1 && radius >= 0 && radius <= 4);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(patches && chi && probe && taps_host && numerator && denominator);
  TkTaps taps;
  taps.r = radius;
  for (int k = 0; k < 9; ++k) taps.t[k] = k <= 2 * radius ? taps_host[k] : 0.f;
  const TkProbe pr =
      tk_make_probe(probe, 0, eigen_probe, eigen_weights, num_eigen, eigen_modes, S, pw);
  // the pair kernel: radius 2, a window whose columns tile 256 threads
  const int win = pw - 2 * (pw / 4), wcols = win < 256 ? win : 256;
  const bool pairs = g_stats_pairs && radius == 2 && pw >= 16 && nscan > 1 &&
                     (win % 256 == 0 || 256 % win == 0) && win % (256 / wcols) == 0;
  const long nitem = pairs ? (nscan + 1) / 2 : nscan;
  int nsplit = 1;
  while (nsplit < 16 && nitem * nsplit * 2 <= 8192 && pw >= 64 &&
         (!pairs || win % (2 * nsplit * (256 / wcols)) == 0))
    nsplit *= 2;
  if (tk_deterministic()) nsplit = 1;
  if (nsplit > 1) {
    hipError_t e = hipMemsetAsync(numerator, 0, sizeof(float) * 2 * (size_t)nscan,
                                  (hipStream_t)stream);
    if (e == hipSuccess)
      e = hipMemsetAsync(denominator, 0, sizeof(float) * 2 * (size_t)nscan, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  if (pairs)
    hipLaunchKernelGGL(position_sums_pair_kernel, dim3(tk_grid(nitem * nsplit, 16)), dim3(256), 0,
                       (hipStream_t)stream, (const cf*)patches, (const cf*)chi, chi_modes, pr,
                       taps, numerator, denominator, pw, nscan, nsplit);
  else if (radius == 2)  // position.py:779-810: sigma = 0.333, truncate 4 -> radius 2
    hipLaunchKernelGGL(position_sums_kernel<2>, dim3((unsigned)nscan * nsplit), dim3(256), 0,
                       (hipStream_t)stream, (const cf*)patches, (const cf*)chi, chi_modes, pr,
                       taps, numerator, denominator, pw, nsplit);
  else
    hipLaunchKernelGGL(position_sums_kernel<-1>, dim3((unsigned)nscan * nsplit), dim3(256), 0,
                       (hipStream_t)stream, (const cf*)patches, (const cf*)chi, chi_modes, pr,
                       taps, numerator, denominator, pw, nsplit);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

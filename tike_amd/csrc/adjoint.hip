// Ptycho.adj fused for gfx950 (reference operators/cupy/ptycho.py:148-176 over
// propagation.py:59-73 and convolution.py:103-154):
//
//   chi_n,s      = scale * IFFT2( farplane_n,s )            (probe window = detector)
//   probe_adj_n,s = conj(O_n) * chi_n,s                     one array PER POSITION
//   psi_adj      = sum_n scatter_n( sum_s conj(P_n,s) chi_n,s )
//
// with O_n the bilinear object patch at scan position n and P_n,s the probe
// (shared, or one per position).  Three launches per sub-batch of positions:
//   1. adj_ifft2_pass1_kernel   far plane -> the probe_adj ARRAY (it is the
//      workspace of the two-pass inverse transform: fft_engine2.h pass 1),
//   2. ifft2_pass2_adjoint_kernel   pass 2 IN PLACE on that array, pixel-major
//      and fused with both products: the thread that finishes a chi value
//      gathers its O_n from psi (two 8-byte loads per pixel, the tap to the
//      right comes from the neighbouring lane), overwrites the value with
//      conj(O) chi and leaves conj(P) chi -- summed over the modes through LDS
//      -- in `objproj` (nscan,pw,pw),
//   3. the grouped footprint scatter of lstsq.hip (tike_scatter_patches) into a
//      planar accumulator; one psi-sized kernel interleaves it at the end.
// chi is never stored; HBM sees the far plane once (read), probe_adj once
// (write) when a sub-batch's intermediate stays in the Infinity Cache.
#include <type_traits>

#include "fft_engine2.h"
#include "internal.h"
#include "tike_amd.h"

// ------------------------------------------- pass 1 alone, on plain tiles
// (inverse: Ptycho.adj; forward and inverse: the Fresnel steps of a
// multislice object).  KEEP: plain stores (retained by the Infinity Cache,
// where pass 2 of the same sub-batch finds them); otherwise non-temporal.
template <int N, bool INV, bool KEEP>
__global__ __launch_bounds__(N, (N <= 256 ? 4 : 2)) void plain_pass1_kernel(
    const cf* __restrict__ farplane, cf* __restrict__ work, long ntile,
    const cf* __restrict__ twtab) {
  using G2 = Fft2Geom<N>;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const cf* __restrict__ src = farplane + tile * (long)N * N;
    cf* mid = work + tile * (long)N * N;
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    for (int r = 0; r < G2::RB; ++r)
      fft2_pass1<N, INV, !KEEP>(
          lds, twtab, tw, line, j, r,
          [&](int y, int e, auto) { return tk_ld_stream(src + y * N + e); }, mid);
  }
}

// ------------------------------------- inverse pass 2 + both adjoint products
// Same decomposition as ifft2_pass2_gradients_kernel (lstsq.hip): a workgroup
// owns the RB rows {ya + 16 yb} x 64 * CW columns of the tile and walks a chunk
// of positions; its four waves are MW mode-waves x CW column-waves, lane =
// column; wave (mw, cw) handles modes {mw, mw + MW, ...}.  `work` holds the
// pass-1 output on entry and probe_adj on exit: a thread reads and writes the
// same RB elements of a (position, mode) tile.
//
// O_n on the fly.  Interior positions (footprint + 1 inside the image): per
// row two 8-byte loads (row y and y + 1 of psi at column sx + x); the taps at
// x + 1 are the neighbouring lane's values (DPP wave shift), lane 63 takes
// them from a load at a wave-uniform address.  Other positions: the reference's
// tap rules (convolution.cu:101-134: pixels outside the image are skipped,
// trailing taps addressed linearly), through tk_gather.
__device__ __forceinline__ float tk_lane_up(float v) {
  // lane i receives lane i + 1 (wave_shl:1); lane 63 receives 0
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}

// OUT selects what becomes of chi and of conj(O) chi:
//   0  Ptycho.adj: the tile is overwritten with conj(O_n) chi (probe_adj, one
//      array per position);
//   1  a slice of a multislice object that has slices in front of it
//      (rpie.py:444-472): sum_n conj(O_n) chi accumulates in registers over the
//      chunk (one atomic per pixel, mode and chunk into `pnum`, times
//      `pnum_scale`) and the tile is overwritten with chi itself, which the
//      Fresnel step back to the slice in front transforms next;
//   2  the first slice: the same sums, and only mode 0 of chi is kept (`chi0`,
//      for the eigen-probe weights).
// PW: position-waves.  A 128-wide tile has only two 64-column waves; with ONE
// mode the other two waves of the workgroup take every second position of the
// chunk instead of idling as a second mode-wave (no sum over modes: the waves
// stay independent).
template <int N, int MW, int MPW, bool PER_POS, int OUT, int PW = 1>
__global__ __launch_bounds__(256, (N == 512 && OUT > 0) ? 1 : (N == 512 || MW > 1 || OUT > 0) ? 2 : 3)
void ifft2_pass2_adjoint_kernel(
    cf* work, const cf* __restrict__ psi, const float* __restrict__ scan,
    const cf* __restrict__ probe, cf* __restrict__ objproj, float* __restrict__ pnum,
    float pnum_scale, cf* __restrict__ chi0, int nscan, int S, int H, int W, float inv_scale,
    int chunk, float* __restrict__ pnum_part) {
  constexpr int RB = N / 16;
  constexpr int CW = 4 / (MW * PW);
  constexpr int NCB = N / (64 * CW);
  constexpr int NSLICE = 16 * NCB;
  constexpr bool REDUCE = MW > 1;
  constexpr int G = 8;  // rows whose operands are requested together
  static_assert(NCB >= 1 && NSLICE % 8 == 0, "slice layout");
  static_assert(PW == 1 || MW == 1, "position-waves only without a sum over modes");
  static_assert(MW > 1 || MPW == 1, "a lone mode-wave writes objproj straight from one mode");
  // shared probe hoisted in registers when it fits beside one butterfly
  // (two modes per wave, or the numerator's accumulators: 64 more registers
  // would leave one wave per SIMD or spill)
  constexpr bool HOIST = !PER_POS && RB <= 16 && MPW == 1 && OUT == 0;
  // (two slot sets = one barrier per position; with two column-waves the
  // shared patch takes their place: LDS must hold two workgroups per CU)
  constexpr int NBUF = RB <= 16 && MW == 4 ? 2 : 1;
  __shared__ cf part[REDUCE ? NBUF * 4 * RB * 64 : 1];  // [buf][wave][yb][lane]
  constexpr long P = (long)N * N;
  const int v = blockIdx.x;
  constexpr int per = NSLICE / 8;
  const int slice = (v & 7) * per + (v >> 3) % per;
  // chunks in DESCENDING order: pass 1 wrote ascending, its tail is the
  // likeliest to be cached still
  const int nchunk_ = (nscan + chunk - 1) / chunk;
  const int b0 = (nchunk_ - 1 - ((v >> 3) / per)) * chunk;
  const int b1 = min(nscan, b0 + chunk);
  const int ya = slice / NCB, cb = slice % NCB;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int mw = w % MW, cw = (w / MW) % CW, pwi = w / (MW * CW);
  constexpr long ROW = 16 * N;
  const int x0 = (cb * CW + cw) * 64;  // first column of this wave
  const long slice0 = (long)ya * N + x0;
  const unsigned lb = (unsigned)lane * (unsigned)sizeof(cf);
  const long total = (long)H * W;
  cf acc[OUT > 0 ? MPW : 1][OUT > 0 ? RB : 1];
  if (OUT > 0) {
#pragma unroll
    for (int m = 0; m < MPW; ++m)
#pragma unroll
      for (int yb = 0; yb < RB; ++yb) acc[m][yb] = mk(0.f, 0.f);
  }
  cf Pc[HOIST ? MPW : 1][HOIST ? RB : 1];
  if (HOIST) {
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
      const int s = mw + MW * m;
      const int sc = s < S ? s : S - 1;
#pragma unroll
      for (int yb = 0; yb < RB; ++yb)
        Pc[m][yb] = conjf(*tk_at(probe + (long)sc * P + slice0 + yb * ROW, lb));
    }
  }
  // conj(O_n) of CNT rows yb = yb0 + i * step of this thread's column
  auto patch_rows = [&](int n, auto cnt_tag, int yb0, int step, auto& o) {
    constexpr int CNT = decltype(cnt_tag)::value;
    const TkCorner c = tk_corner(scan, n);
    const bool interior = c.sy >= 0 && c.sx >= 0 && c.sy + N < H && c.sx + N < W &&
                          total < (1L << 28);  // wave-uniform
    if (interior) {
      const cf* __restrict__ top = psi + (long)(c.sy + ya + 16 * yb0) * W + c.sx + x0;
      const long rstep = (long)(16 * step) * W;
      cf a[CNT], d[CNT], ea[CNT], ed[CNT];
#pragma unroll
      for (int i = 0; i < CNT; ++i) {
        a[i] = *tk_at(top + i * rstep, lb);
        d[i] = *tk_at(top + i * rstep + W, lb);
        // the column right of the wave's last one: a uniform address
        ea[i] = top[i * rstep + 64];
        ed[i] = top[i * rstep + W + 64];
      }
#pragma unroll
      for (int i = 0; i < CNT; ++i) {
        cf b = mk(tk_lane_up(a[i].x), tk_lane_up(a[i].y));
        cf e = mk(tk_lane_up(d[i].x), tk_lane_up(d[i].y));
        if (lane == 63) {
          b = ea[i];
          e = ed[i];
        }
        cf r = mk(a[i].x * c.w00, a[i].y * c.w00);
        r.x += b.x * c.w01;
        r.y += b.y * c.w01;
        r.x += d[i].x * c.w10;
        r.y += d[i].y * c.w10;
        r.x += e.x * c.w11;
        r.y += e.y * c.w11;
        o[i] = conjf(r);
      }
    } else {
#pragma unroll
      for (int i = 0; i < CNT; ++i) {
        const int y = c.sy + ya + 16 * (yb0 + i * step), x = c.sx + x0 + lane;
        const bool ok = y >= 0 && y < H && x >= 0 && x < W;
        const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
        const int xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
        const cf r = tk_gather(psi, (long)yc * W + xc, W, total, c);
        o[i] = ok ? conjf(r) : mk(0.f, 0.f);
      }
    }
  };
  // Several mode-waves work on the SAME pixels: each gathers 1 / MW of the
  // rows of conj(O_n) and leaves them in LDS for the others -- for position
  // n + 1 while position n is being processed, so that the barrier that
  // closes the sum over the modes also publishes the next patch.
  // (512^2: the slots of the mode sum alone take 64 KiB; a shared patch beside
  // them would leave one workgroup per CU)
  constexpr bool SHARE = MW > 1 && N <= 256;
  constexpr int QR = RB / MW;  // rows gathered per wave
  __shared__ cf osh[SHARE ? 2 * CW * RB * 64 : 1];  // [buf][column-wave][yb][lane]
  auto publish = [&](int n, int buf) {
    cf o[QR];
    patch_rows(n, std::integral_constant<int, QR>{}, mw, MW, o);
#pragma unroll
    for (int i = 0; i < QR; ++i) osh[((buf * CW + cw) * RB + mw + MW * i) * 64 + lane] = o[i];
  };
  if (SHARE) {
    if (b0 < b1) publish(b0, 0);
    __syncthreads();
  }
  for (int n = b0 + pwi; n < b1; n += PW) {
    unsigned lo = lb;
    asm volatile("" : "+v"(lo));
    const int obuf = (n - b0) & 1;
    if (SHARE && n + 1 < b1) publish(n + 1, obuf ^ 1);
    const cf* __restrict__ on = osh + ((obuf * CW + cw) * RB) * 64 + lane;
    cf* slot = part + ((((n - b0) & (NBUF - 1)) * 4 + w) * RB) * 64 + lane;
    const cf* slots = part + ((((n - b0) & (NBUF - 1)) * 4 + cw * MW) * RB) * 64 + lane;
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
      const int s = mw + MW * m;
      if (s < S) {  // wave-uniform
        cf* tile = work + ((long)n * S + s) * P + slice0;
        const cf* __restrict__ Ps =
            probe + (PER_POS ? ((long)n * S + s) * P : (long)s * P) + slice0;
        cf u[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) u[k] = *tk_at(tile + k * ROW, lo);
        Dft<RB, true>::run(u);
#pragma unroll
        for (int g = 0; g < RB; g += G) {
          cf oc[G], pc[G];
          if (SHARE) {
#pragma unroll
            for (int i = 0; i < G; ++i) oc[i] = on[(g + i) * 64];
          } else {
            patch_rows(n, std::integral_constant<int, G>{}, g, 1, oc);
          }
#pragma unroll
          for (int i = 0; i < G; ++i)
            pc[i] = HOIST ? Pc[m][g + i] : conjf(tk_ld_stream(tk_at(Ps + (g + i) * ROW, lo)));
#pragma unroll
          for (int i = 0; i < G; ++i) {
            const cf chi = u[g + i] * inv_scale;  // chi of row ya + 16 (g + i)
            if (OUT == 0) {
              tk_st_stream(tk_at(tile + (g + i) * ROW, lo), oc[i] * chi);
            } else {
              acc[m][g + i] = acc[m][g + i] + oc[i] * chi;
              if (OUT == 1)
                tk_st_stream(tk_at(tile + (g + i) * ROW, lo), chi);
              else if (s == 0 && chi0 != nullptr)
                tk_st_stream(tk_at(chi0 + (long)n * P + slice0 + (g + i) * ROW, lo), chi);
            }
            const cf t = pc[i] * chi;
            if (REDUCE)
              slot[(g + i) * 64] = m == 0 ? t : slot[(g + i) * 64] + t;
            else
              tk_st_stream(tk_at(objproj + (long)n * P + slice0 + (g + i) * ROW, lo), t);
          }
        }
      } else if (REDUCE && m == 0) {
#pragma unroll
        for (int yb = 0; yb < RB; ++yb) slot[yb * 64] = mk(0.f, 0.f);
      }
    }
    if (REDUCE) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < RB / MW; ++q) {
        const int yb = mw + MW * q;
        cf sum = slots[yb * 64];
#pragma unroll
        for (int k = 1; k < MW; ++k) sum = sum + slots[(k * RB + yb) * 64];
        tk_st_stream(tk_at(objproj + (long)n * P + slice0 + yb * ROW, lo), sum);
      }
      if (NBUF == 1) __syncthreads();
    }
  }
  if (OUT > 0 && pnum != nullptr) {
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
      const int s = mw + MW * m;
      if (s < S) {
#pragma unroll
        for (int yb = 0; yb < RB; ++yb) {
          if (pnum_part != nullptr) {
            // deterministic mode: the chunk's own row, summed in chunk order
            // by tk_ordered_sum after the launch
            // (position-waves keep separate sums: a row each)
            const long ci = (long)((v >> 3) / per) * PW + pwi;
            float* q = tk_at(pnum_part + 2 * ((ci * S + s) * P + slice0 + yb * ROW), lb);
            q[0] = acc[m][yb].x * pnum_scale;
            q[1] = acc[m][yb].y * pnum_scale;
            continue;
          }
          float* o = tk_at(pnum + 2 * ((long)s * P + slice0 + yb * ROW), lb);
          unsafeAtomicAdd(o, acc[m][yb].x * pnum_scale);
          unsafeAtomicAdd(o + 1, acc[m][yb].y * pnum_scale);
        }
      }
    }
  }
}

// psi_adj (H,W) c64 = planar accumulator (2,H,W) f32
__global__ __launch_bounds__(256) void adj_interleave_kernel(const float* __restrict__ acc,
                                                             cf* __restrict__ out, long npix) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < npix; i += gridDim.x * 256L)
    out[i] = mk(acc[i], acc[npix + i]);
}

template <int N>
static int launch_plain_pass1(const cf* in, cf* out, long ntile, bool inverse, bool keep,
                              hipStream_t stream) {
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
#define TK_PP1(INV, KEEP)                                                                     \
  hipLaunchKernelGGL((plain_pass1_kernel<N, INV, KEEP>), dim3(tk_grid(ntile, 4)), dim3(N), 0, \
                     stream, in, out, ntile, tw)
  if (inverse && keep)
    TK_PP1(true, true);
  else if (inverse)
    TK_PP1(true, false);
  else if (keep)
    TK_PP1(false, true);
  else
    TK_PP1(false, false);
#undef TK_PP1
  TK_LAUNCH_CHECK();
  return TK_OK;
}

int tk_fft2_pass1(const cf* in, cf* out, long ntile, int det, bool inverse, bool keep,
                  hipStream_t stream) {
  switch (det) {
    case 128: return launch_plain_pass1<128>(in, out, ntile, inverse, keep, stream);
    case 256: return launch_plain_pass1<256>(in, out, ntile, inverse, keep, stream);
    case 512: return launch_plain_pass1<512>(in, out, ntile, inverse, keep, stream);
    default: return TK_ERR_UNSUPPORTED;
  }
}

// out: 0 probe_adj per position, 1 probe numerator + chi in place, 2 probe
// numerator + chi0 (see the kernel)
int tk_ifft2_pass2_products(cf* work, const cf* psi, const float* scan, const cf* probe,
                            int probe_per_scan, cf* objproj, float* pnum, float pnum_scale,
                            cf* chi0, int out, int nscan, int S, int det, int H, int W,
                            float inv_scale, hipStream_t stream) {
  int MW = S >= 3 ? 4 : S;
  const int MPW = S > 4 ? 2 : 1;
  // 128^2 x one mode: two column-waves x two position-waves (see the kernel)
  const int PW = det == 128 && MW == 1 ? 2 : 1;
  const int nslice = 16 * (det / (64 * (4 / (MW * PW))));
  // (slice, chunk) workgroups: about two rounds of the chip, but chunks of at
  // least 8 positions (a workgroup loads its slice of a shared probe once)
  int nchunk = (1536 + nslice - 1) / nslice;
  int chunk = (nscan + nchunk - 1) / nchunk;
  if (chunk < 8) chunk = 8;
  nchunk = (nscan + chunk - 1) / chunk;
  // deterministic mode: per-chunk partial sums of the probe numerator in the
  // caller's scratch buffer; ONE chunk (a single contributor per address)
  // when it is too small
  float* pnum_part = nullptr;
  const long pnum_len = 2L * S * det * det;
  if (out != 0 && pnum != nullptr && tk_deterministic()) {
    pnum_part = tk_det_scratch(sizeof(float) * (size_t)pnum_len * nchunk * PW);
    if (pnum_part == nullptr) {
      chunk = nscan > 8 ? nscan : 8;
      nchunk = (nscan + chunk - 1) / chunk;
    }
  }
  const dim3 grid((unsigned)(nslice * nchunk)), block(256);
#define TK_ADJ_O(N, MW_, MPW_, PP, OUT_)                                                       \
  hipLaunchKernelGGL(                                                                          \
      (ifft2_pass2_adjoint_kernel<N, MW_, MPW_, PP, OUT_, (N == 128 && MW_ == 1) ? 2 : 1>),    \
      grid, block, 0, stream, work, psi, scan, probe, objproj, pnum, pnum_scale, chi0, nscan,  \
      S, H, W, inv_scale, chunk, pnum_part)
#define TK_ADJ(N, MW_, MPW_)                        \
  do {                                              \
    if (probe_per_scan && out == 0)                 \
      TK_ADJ_O(N, MW_, MPW_, true, 0);              \
    else if (probe_per_scan && out == 1)            \
      TK_ADJ_O(N, MW_, MPW_, true, 1);              \
    else if (probe_per_scan)                        \
      TK_ADJ_O(N, MW_, MPW_, true, 2);              \
    else if (out == 0)                              \
      TK_ADJ_O(N, MW_, MPW_, false, 0);             \
    else if (out == 1)                              \
      TK_ADJ_O(N, MW_, MPW_, false, 1);             \
    else                                            \
      TK_ADJ_O(N, MW_, MPW_, false, 2);             \
  } while (0)
#define TK_ADJ_N(N, A)               \
  do {                               \
    if (MW == 1)                     \
      A(N, 1, 1);                    \
    else if (MW == 2)                \
      A(N, 2, 1);                    \
    else if (MPW == 1)               \
      A(N, 4, 1);                    \
    else                             \
      A(N, 4, 2);                    \
  } while (0)
  // (512^2 with the numerator's accumulators beside a radix-32 butterfly: one
  // workgroup of 256 per SIMD pair -- at two it would spill 70 registers)
  switch (det) {
    case 128: TK_ADJ_N(128, TK_ADJ); break;
    case 256: TK_ADJ_N(256, TK_ADJ); break;
    default: TK_ADJ_N(512, TK_ADJ); break;
  }
#undef TK_ADJ_N
#undef TK_ADJ
#undef TK_ADJ_O
  TK_LAUNCH_CHECK();
  if (pnum_part != nullptr)
    return tk_ordered_sum(pnum, pnum_part, pnum_len, nchunk * PW, true, stream);
  return TK_OK;
}

extern "C" int tike_ptycho_adj(const void* farplane, const void* probe, int probe_per_scan,
                               const float* scan, const void* psi, void* psi_adj,
                               void* probe_adj, void* objproj_work, float* acc_work, int nscan,
                               int S, int pw, int det, int H, int W, float scale, int sub_batch,
                               void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 1 && det >= pw && H >= 1 && W >= 1);
  TK_CHECK_ARG(psi_adj && acc_work);
  if (pw != det || (det != 128 && det != 256 && det != 512) || S > 8) return TK_ERR_UNSUPPORTED;
  const long npix = (long)H * W;
  hipError_t e = hipMemsetAsync(acc_work, 0, sizeof(float) * 2 * npix, stream);
  if (e != hipSuccess) return (int)e;
  if (nscan > 0) {
    TK_CHECK_ARG(farplane && probe && scan && psi && probe_adj && objproj_work);
    TK_CHECK_ARG(probe_adj != farplane);
    const size_t tile_bytes = sizeof(cf) * (size_t)det * det;
    // sub_batch > 0: sub-batches of that many positions -- pass 1 leaves its
    // output in the Infinity Cache (plain stores), pass 2 finds it there and
    // overwrites it in place.  The default (0, as -1) is ONE batch since late
    // round 6: measured, the launches of a sub-batch cost more than the cache
    // gives back (256^2 x 1 mode, 4096 positions: 0.391 of the roofline at 256
    // MiB per sub-batch, 0.419 at 512, 0.428 at 1024, 0.438 in one batch; 256^2
    // x 8 0.314 -> 0.319, 128^2 x 1 0.373 -> 0.383)
    long sub = sub_batch > 0 ? sub_batch : nscan;
    if (sub < 1) sub = 1;
    const bool keep = sub < nscan;
    for (long lo = 0; lo < nscan; lo += sub) {
      const int m = (int)(nscan - lo < sub ? nscan - lo : sub);
      const cf* far = (const cf*)farplane + lo * S * det * det;
      cf* work = (cf*)probe_adj + lo * S * det * det;
      const cf* pr = (const cf*)probe + (probe_per_scan ? lo * S * det * det : 0L);
      int rc;
      rc = tk_fft2_pass1(far, work, (long)m * S, det, true, keep, stream);
      if (rc) return rc;
      rc = tk_ifft2_pass2_products(work, (const cf*)psi, scan + 2 * lo, pr, probe_per_scan,
                                   (cf*)objproj_work, nullptr, 0.f, nullptr, 0, m, S, det, H, W,
                                   scale, stream);
      if (rc) return rc;
      rc = tike_scatter_patches(objproj_work, scan + 2 * lo, acc_work, m, pw, H, W, stream_);
      if (rc) return rc;
    }
  }
  hipLaunchKernelGGL(adj_interleave_kernel, dim3(tk_grid((npix + 255) / 256, 8)), dim3(256), 0,
                     stream, acc_work, (cf*)psi_adj, npix);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

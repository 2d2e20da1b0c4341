// Fused stages of a multislice object (reference operators/cupy/multislice.py:
// 69-92,144-194; fresnelspectprop.py:52-113; ptycho/solvers/rpie.py:367-495).
//
// The free-space step between two slices is FFT2 -> x propagator -> IFFT2
// (FresnelSpectProp).  As generic transforms that is four trips of every wave
// through memory plus one for the multiply; here it is three launches with
// two hand-offs, built from the two-pass engine (fft_engine2.h):
//   1. forward pass 1          rows + radix-16 column stage.  The first one of
//                              a slice is tike_fwd_pass1 (it forms patch x
//                              incident probe on the fly); on a stored wave
//                              (the way back) it is tike_fft2_pass1;
//   2. tike_fresnel_colpass    forward column pass -> x H (or conj H, the
//                              adjoint) -> inverse radix-16 over k2, twiddle,
//                              LDS transpose, inverse row transforms: the
//                              spectrum never exists in memory;
//   3. inverse pass 2          in place: tike_fft2_pass2_inplace on the way
//                              forward (the result is the next slice's
//                              incident probe), tike_ifft2_pass2_products on
//                              the way back (fused with both numerators of the
//                              slice in front).
#include "fft_engine2.h"
#include "internal.h"
#include "tike_amd.h"

// Work item = (tile, k1): a thread owns column t of the 16 rows {16 r + k1} of
// the hand-off, software pipelined over k1 (the rows of k1 + 1 are requested
// before the butterflies of k1).  H (N,N): the propagator in FFT order, shared
// by every tile (L2).  N = 256 (128 and 512: fresnel_colpass_n_kernel below).
template <bool CONJ>
__global__ __launch_bounds__(256, 4) void fresnel_colpass_kernel(
    const cf* __restrict__ colin, cf* __restrict__ work, long ntile, float scale,
    const cf* __restrict__ twtab, const cf* __restrict__ Hprop, int ksplit) {
  constexpr int N = 256;
  using G2 = Fft2Geom<N>;
  static_assert(G2::RB == 16, "one radix-16 per direction");
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int t = threadIdx.x;
  const int kn = 16 / ksplit;
  for (long w = blockIdx.x; w < ntile * ksplit; w += gridDim.x) {
    // tiles in DESCENDING order: pass 1 wrote them ascending
    const long tile = ntile - 1 - w / ksplit;
    const int kbeg = (int)(w % ksplit) * kn;
    const cf* __restrict__ src = colin + tile * (long)N * N;
    cf* mid = work + tile * (long)N * N;
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
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
      cf h[16];
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) h[k2] = Hprop[(k1 + 16 * k2) * N + t];
      Dft<16, false>::run(u);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2)
        u[k2] = (u[k2] * (CONJ ? conjf(h[k2]) : h[k2])) * scale;
      Dft<16, true>::run(u);
#pragma unroll
      for (int ya = 1; ya < 16; ++ya) u[ya] = mul_tw<true>(u[ya], twtab[N + k1 * ya]);
      fft2_rows_from_columns<N, true, true>(lds, tw, line, j, u, mid + (long)(16 * k1) * N);
    }
  }
}

// The last pass of a Fresnel step and the first pass of the next slice's
// transform in one launch (multislice.py:86-91 followed by :79-85 of the next
// slice): a work item = (position, k1) finishes the inverse on the rows
// {16 r + k1} of every mode -- the probe incident on the slice, rows
// {k1 + 16 k2}, written back in place for the way back -- multiplies by the
// slice's object patch (gathered once per work item, 16 pixels per thread) and,
// these being exactly the 16 rows of group r = k1 of a forward pass 1, starts
// the next transform on the same registers: radix-16 over k2, twiddle, LDS
// transpose, forward row transforms (the two directions commute).
// tike_fft2_pass2_inplace + tike_fwd_pass1 with per-position probes read the
// incident wave a second time; here it never leaves the registers.  N = 256.
__global__ __launch_bounds__(256, 2) void slice_step_kernel(
    cf* wave, const cf* __restrict__ psi, const float* __restrict__ scan, cf* __restrict__ far,
    long nscan, int S, int H, int W, float scale, const cf* __restrict__ twtab) {
  constexpr int N = 256;
  using G2 = Fft2Geom<N>;
  static_assert(G2::RB == 16, "one radix-16 per direction");
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int t = threadIdx.x;
  const long total = (long)H * W;
  for (long v = blockIdx.x; v < nscan * 16; v += gridDim.x) {
    const int k1 = (int)(v & 15);
    const long n = nscan - 1 - (v >> 4);  // descending: the column pass wrote ascending
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    const TkCorner c = tk_corner(scan, n);
    // object patch at rows k1 + 16 k2, column t
    cf O[16];
    const bool interior = c.sy >= 0 && c.sx >= 0 && c.sy + N < H && c.sx + N < W &&
                          total < (1L << 28);
    if (interior) {  // uniform
      typedef float tk_v4f __attribute__((ext_vector_type(4)));
      const unsigned row_bytes = (unsigned)W * 8u;
      const unsigned off0 = (unsigned)((c.sy + k1) * W + c.sx + t) * 8u;
      tk_v4f u4[16], l4[16];
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        const unsigned off = off0 + (unsigned)(16 * k2) * row_bytes;
        __builtin_memcpy(&u4[k2], reinterpret_cast<const char*>(psi) + off, 16);
        __builtin_memcpy(&l4[k2], reinterpret_cast<const char*>(psi) + off + row_bytes, 16);
      }
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        cf o = mk(u4[k2].x * c.w00, u4[k2].y * c.w00);
        o.x += u4[k2].z * c.w01;
        o.y += u4[k2].w * c.w01;
        o.x += l4[k2].x * c.w10;
        o.y += l4[k2].y * c.w10;
        o.x += l4[k2].z * c.w11;
        o.y += l4[k2].w * c.w11;
        O[k2] = o;
      }
    } else {
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        const int y = c.sy + k1 + 16 * k2, x = c.sx + t;
        const bool ok = y >= 0 && y < H && x >= 0 && x < W;
        const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
        const int xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
        const cf o = tk_gather(psi, (long)yc * W + xc, W, total, c);
        O[k2] = ok ? o : mk(0.f, 0.f);
      }
    }
    for (int s = 0; s < S; ++s) {
      cf* p = wave + (n * S + s) * (long)N * N + (long)k1 * N + t;
      cf u[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) u[r] = p[(long)(16 * r) * N];
      Dft<16, true>::run(u);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        u[k2] = u[k2] * scale;
        p[(long)(16 * k2) * N] = u[k2];  // the incident probe, kept for the way back
        u[k2] = u[k2] * O[k2];
      }
      Dft<16, false>::run(u);
#pragma unroll
      for (int ya = 1; ya < 16; ++ya) u[ya] = mul_tw<false>(u[ya], twtab[N + k1 * ya]);
      fft2_rows_from_columns<N, false, true>(lds, tw, line, j, u,
                                             far + (n * S + s) * (long)N * N + (long)(16 * k1) * N);
    }
  }
}

// ---- the same two kernels at 128^2 and 512^2 ---------------------------------
// The column stage of pass 2 has radix RB = N / 16 and the one of pass 1 radix
// 16: only at 256^2 is a work item of the one a work item of the other.
//   512^2 (RB = 32): item q = k1 < 16.  The radix-32 over the rows {16 r + k1}
//       yields the rows k1 + 16 k2, k2 < 32 -- the rows {r' + 32 y2} of TWO
//       16-row groups: r' = k1 (k2 = 2 y2) and r' = k1 + 16 (k2 = 2 y2 + 1).
//   128^2 (RB = 8): item q = r' < 8.  The 16 rows {r' + 8 y2} of the group
//       are the outputs of TWO radix-8 stages: k1 = r' (y2 = 2 k2) and
//       k1 = r' + 8 (y2 = 2 k2 + 1).
// Either way a thread owns column t of NG groups of 16 rows after NF column
// stages, all in registers.  No request ahead of the next item: at 512^2 one
// item holds 64 + 64 registers of values and propagator already.
template <int N>
struct TkSliceItem {
  static constexpr int RB = N / 16;
  static constexpr int NI = RB >= 16 ? 16 : RB;      // items per tile
  static constexpr int NG = RB > 16 ? RB / 16 : 1;   // 16-row groups per item
  static constexpr int NF = RB < 16 ? 16 / RB : 1;   // radix-RB stages per item
  static_assert(NG * 16 == NF * RB, "an item is NG groups = NF stages");
  // value (stage f, output k2) sits in group g at y2
  static __device__ __forceinline__ constexpr int group(int f, int k2) {
    return NG > 1 ? k2 % NG : 0;
  }
  static __device__ __forceinline__ constexpr int y2(int f, int k2) {
    return NG > 1 ? k2 / NG : (NF > 1 ? NF * k2 + f : k2);
  }
};

// tike_fresnel_colpass at N in {128, 512}: work item = (tile, q).
template <int N, bool CONJ>
__global__ __launch_bounds__(N, N == 512 ? 1 : 4) void fresnel_colpass_n_kernel(
    const cf* __restrict__ colin, cf* __restrict__ work, long ntile, float scale,
    const cf* __restrict__ twtab, const cf* __restrict__ Hprop) {
  using G2 = Fft2Geom<N>;
  using It = TkSliceItem<N>;
  constexpr int RB = It::RB;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int t = threadIdx.x;
  for (long w = blockIdx.x; w < ntile * It::NI; w += gridDim.x) {
    const long tile = ntile - 1 - w / It::NI;  // descending: pass 1 wrote ascending
    const int q = (int)(w % It::NI);
    const cf* __restrict__ src = colin + tile * (long)N * N;
    cf* mid = work + tile * (long)N * N;
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    cf v[It::NG][16];
#pragma unroll
    for (int f = 0; f < It::NF; ++f) {
      const int k1 = q + It::NI * f;
      cf u[RB], h[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) u[r] = tk_ld_stream(src + (long)(16 * r + k1) * N + t);
#pragma unroll
      for (int k2 = 0; k2 < RB; ++k2) h[k2] = Hprop[(long)(k1 + 16 * k2) * N + t];
      Dft<RB, false>::run(u);
#pragma unroll
      for (int k2 = 0; k2 < RB; ++k2)
        v[It::group(f, k2)][It::y2(f, k2)] = (u[k2] * (CONJ ? conjf(h[k2]) : h[k2])) * scale;
    }
#pragma unroll
    for (int g = 0; g < It::NG; ++g) {
      const int rp = q + 16 * g;  // the group's rows: rp + RB y2
      Dft<16, true>::run(v[g]);
#pragma unroll
      for (int ya = 1; ya < 16; ++ya) v[g][ya] = mul_tw<true>(v[g][ya], twtab[N + rp * ya]);
      fft2_rows_from_columns<N, true, true>(lds, tw, line, j, v[g], mid + (long)(16 * rp) * N);
    }
  }
}

// tike_slice_step at N in {128, 512}: work item = (position, q).
template <int N>
__global__ __launch_bounds__(N, N == 512 ? 1 : 3) void slice_step_n_kernel(
    cf* wave, const cf* __restrict__ psi, const float* __restrict__ scan, cf* __restrict__ far,
    long nscan, int S, int H, int W, float scale, const cf* __restrict__ twtab) {
  using G2 = Fft2Geom<N>;
  using It = TkSliceItem<N>;
  constexpr int RB = It::RB;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int t = threadIdx.x;
  const long total = (long)H * W;
  for (long w = blockIdx.x; w < nscan * It::NI; w += gridDim.x) {
    const int q = (int)(w % It::NI);
    const long n = nscan - 1 - w / It::NI;  // descending: the column pass wrote ascending
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    const TkCorner c = tk_corner(scan, n);
    // object patch at the rows rp + RB y2 of every group, column t
    cf O[It::NG][16];
    const bool interior = c.sy >= 0 && c.sx >= 0 && c.sy + N < H && c.sx + N < W &&
                          total < (1L << 28);
#pragma unroll
    for (int g = 0; g < It::NG; ++g) {
      const int rp = q + 16 * g;
      if (interior) {  // uniform
        typedef float tk_v4f __attribute__((ext_vector_type(4)));
        const unsigned row_bytes = (unsigned)W * 8u;
        const unsigned off0 = (unsigned)((c.sy + rp) * W + c.sx + t) * 8u;
        // (eight rows requested together: sixteen would hold 128 registers)
#pragma unroll
        for (int yh = 0; yh < 16; yh += 8) {
          tk_v4f u4[8], l4[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const unsigned off = off0 + (unsigned)(RB * (yh + i)) * row_bytes;
            __builtin_memcpy(&u4[i], reinterpret_cast<const char*>(psi) + off, 16);
            __builtin_memcpy(&l4[i], reinterpret_cast<const char*>(psi) + off + row_bytes, 16);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            cf o = mk(u4[i].x * c.w00, u4[i].y * c.w00);
            o.x += u4[i].z * c.w01;
            o.y += u4[i].w * c.w01;
            o.x += l4[i].x * c.w10;
            o.y += l4[i].y * c.w10;
            o.x += l4[i].z * c.w11;
            o.y += l4[i].w * c.w11;
            O[g][yh + i] = o;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int y2 = 0; y2 < 16; ++y2) {
          const int y = c.sy + rp + RB * y2, x = c.sx + t;
          const bool ok = y >= 0 && y < H && x >= 0 && x < W;
          const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
          const int xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
          const cf o = tk_gather(psi, (long)yc * W + xc, W, total, c);
          O[g][y2] = ok ? o : mk(0.f, 0.f);
        }
      }
    }
    for (int s = 0; s < S; ++s) {
      cf v[It::NG][16];
#pragma unroll
      for (int f = 0; f < It::NF; ++f) {
        const int k1 = q + It::NI * f;
        cf* p = wave + (n * S + s) * (long)N * N + (long)k1 * N + t;
        cf u[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) u[r] = p[(long)(16 * r) * N];
        Dft<RB, true>::run(u);
#pragma unroll
        for (int k2 = 0; k2 < RB; ++k2) {
          u[k2] = u[k2] * scale;
          p[(long)(16 * k2) * N] = u[k2];  // the incident probe, kept for the way back
          v[It::group(f, k2)][It::y2(f, k2)] = u[k2] * O[It::group(f, k2)][It::y2(f, k2)];
        }
      }
#pragma unroll
      for (int g = 0; g < It::NG; ++g) {
        const int rp = q + 16 * g;
        Dft<16, false>::run(v[g]);
#pragma unroll
        for (int ya = 1; ya < 16; ++ya) v[g][ya] = mul_tw<false>(v[g][ya], twtab[N + rp * ya]);
        fft2_rows_from_columns<N, false, true>(
            lds, tw, line, j, v[g], far + (n * S + s) * (long)N * N + (long)(16 * rp) * N);
      }
    }
  }
}

// Pass 2 alone, in place: rows {k1 + 16 r} of a tile in, the same rows out.
// Work item = (tile, k1, 256-column block), tiles in descending order.
template <int N, bool INV>
__global__ __launch_bounds__(256, N == 512 ? 2 : 4) void colpass_inplace_kernel(cf* tiles,
                                                                               long ntile,
                                                                               float scale) {
  constexpr int RB = N / 16, NH = N >= 256 ? N / 256 : 1, NT = N >= 256 ? 256 : N;
  const long nitem = ntile * 16 * NH;
  if ((int)threadIdx.x >= NT) return;
  for (long v = blockIdx.x; v < nitem; v += gridDim.x) {
    const int hb = (int)(v % NH);
    const int k1 = (int)((v / NH) & 15);
    const long tile = ntile - 1 - v / (16 * NH);
    cf* p = tiles + tile * (long)N * N + (long)k1 * N + hb * 256 + threadIdx.x;
    cf u[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) u[r] = p[(long)(16 * r) * N];
    Dft<RB, INV>::run(u);
#pragma unroll
    for (int k2 = 0; k2 < RB; ++k2) tk_st_stream(p + (long)(16 * k2) * N, u[k2] * scale);
  }
}

// Pass 2 of the S modes of a position with the illumination sum_s |wave_s|^2
// formed from the registers (_preconditioner.py:40-45,86-95): the wave is read
// once and, unless KEEP, never written -- tike_fft2_pass2_inplace followed by
// tike_intensity moves 3 T + T / S per position, this T + T / S.
// Work item = (position, k1, 256-column block), positions descending.
template <int N, bool INV, bool KEEP>
__global__ __launch_bounds__(256, N == 512 ? 2 : 4) void colpass_intensity_kernel(
    cf* tiles, float* __restrict__ amp, long nscan, int S, float scale) {
  constexpr int RB = N / 16, NH = N >= 256 ? N / 256 : 1, NT = N >= 256 ? 256 : N;
  const long nitem = nscan * 16 * NH;
  if ((int)threadIdx.x >= NT) return;
  for (long v = blockIdx.x; v < nitem; v += gridDim.x) {
    const int hb = (int)(v % NH);
    const int k1 = (int)((v / NH) & 15);
    const long n = nscan - 1 - v / (16 * NH);
    const long lane = (long)k1 * N + hb * 256 + threadIdx.x;
    float acc[RB];
#pragma unroll
    for (int k2 = 0; k2 < RB; ++k2) acc[k2] = 0.f;
    for (int s = 0; s < S; ++s) {
      cf* p = tiles + (n * S + s) * (long)N * N + lane;
      cf u[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) u[r] = KEEP ? p[(long)(16 * r) * N] : tk_ld_stream(p + (long)(16 * r) * N);
      Dft<RB, INV>::run(u);
#pragma unroll
      for (int k2 = 0; k2 < RB; ++k2) {
        const cf w = u[k2] * scale;
        acc[k2] += w.x * w.x + w.y * w.y;
        if (KEEP) tk_st_stream(p + (long)(16 * k2) * N, w);
      }
    }
    float* __restrict__ a = amp + n * (long)N * N + lane;
#pragma unroll
    for (int k2 = 0; k2 < RB; ++k2) a[(long)(16 * k2) * N] = acc[k2];
  }
}

extern "C" int tike_fft2_pass2_intensity(void* tiles, float* amplitude, long nscan, int S,
                                         int det, int inverse, float scale, int keep,
                                         void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && det >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(tiles != nullptr && amplitude != nullptr);
  const long nitem = nscan * 16 * (det >= 256 ? det / 256 : 1);
  const dim3 grid(tk_grid(nitem, 32)), block(256);
#define TK_CI(N, INV, KEEP)                                                                  \
  hipLaunchKernelGGL((colpass_intensity_kernel<N, INV, KEEP>), grid, block, 0, stream,       \
                     (cf*)tiles, amplitude, nscan, S, scale)
#define TK_CI_N(N)                          \
  do {                                      \
    if (inverse && keep) TK_CI(N, true, true);         \
    else if (inverse) TK_CI(N, true, false);           \
    else if (keep) TK_CI(N, false, true);              \
    else TK_CI(N, false, false);                       \
  } while (0)
  switch (det) {
    case 128: TK_CI_N(128); break;
    case 256: TK_CI_N(256); break;
    case 512: TK_CI_N(512); break;
    default: return TK_ERR_UNSUPPORTED;
  }
#undef TK_CI_N
#undef TK_CI
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_fft2_pass1(const void* in, void* out, long ntile, int det, int inverse,
                               void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(ntile >= 0 && det >= 1);
  if (ntile == 0) return TK_OK;
  TK_CHECK_ARG(in && out && in != out);
  return tk_fft2_pass1((const cf*)in, (cf*)out, ntile, det, inverse != 0, false,
                       (hipStream_t)stream);
}

extern "C" int tike_fft2_pass2_inplace(void* tiles, long ntile, int det, int inverse, float scale,
                                       void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(ntile >= 0 && det >= 1);
  if (ntile == 0) return TK_OK;
  TK_CHECK_ARG(tiles != nullptr);
  const long nitem = ntile * 16 * (det >= 256 ? det / 256 : 1);
  const dim3 grid(tk_grid(nitem, 32)), block(256);
#define TK_CP(N)                                                                              \
  do {                                                                                        \
    if (inverse)                                                                              \
      hipLaunchKernelGGL((colpass_inplace_kernel<N, true>), grid, block, 0, stream, (cf*)tiles, \
                         ntile, scale);                                                       \
    else                                                                                      \
      hipLaunchKernelGGL((colpass_inplace_kernel<N, false>), grid, block, 0, stream,          \
                         (cf*)tiles, ntile, scale);                                           \
  } while (0)
  switch (det) {
    case 128: TK_CP(128); break;
    case 256: TK_CP(256); break;
    case 512: TK_CP(512); break;
    default: return TK_ERR_UNSUPPORTED;
  }
#undef TK_CP
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_slice_step(void* wave, const void* psi, const float* scan, void* farplane1,
                               int nscan, int S, int det, int H, int W, float scale,
                               void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && det >= 1 && H >= 1 && W >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(wave && psi && scan && farplane1 && wave != farplane1);
  if (det != 128 && det != 256 && det != 512) return TK_ERR_UNSUPPORTED;
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
  if (det == 128 || det == 512) {
    if (det == 128)
      hipLaunchKernelGGL(slice_step_n_kernel<128>, dim3(tk_grid((long)nscan * 8, 16)), dim3(128),
                         0, stream, (cf*)wave, (const cf*)psi, scan, (cf*)farplane1, (long)nscan,
                         S, H, W, scale, tw);
    else
      hipLaunchKernelGGL(slice_step_n_kernel<512>, dim3(tk_grid((long)nscan * 16, 1)), dim3(512),
                         0, stream, (cf*)wave, (const cf*)psi, scan, (cf*)farplane1, (long)nscan,
                         S, H, W, scale, tw);
    TK_LAUNCH_CHECK();
    return TK_OK;
  }
  hipLaunchKernelGGL(slice_step_kernel, dim3(tk_grid((long)nscan * 16, 8)), dim3(256), 0, stream,
                     (cf*)wave, (const cf*)psi, scan, (cf*)farplane1, (long)nscan, S, H, W, scale,
                     tw);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_fresnel_colpass(const void* colin, const void* propagator, int adjoint,
                                    void* work, long ntile, int det, float scale,
                                    void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(ntile >= 0 && det >= 1);
  if (ntile == 0) return TK_OK;
  TK_CHECK_ARG(colin && propagator && work && work != colin);
  if (det != 128 && det != 256 && det != 512) return TK_ERR_UNSUPPORTED;
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
  if (det == 128 || det == 512) {
#define TK_FCN(N, CJ, PER)                                                                     \
  hipLaunchKernelGGL((fresnel_colpass_n_kernel<N, CJ>),                                        \
                     dim3(tk_grid(ntile * TkSliceItem<N>::NI, PER)), dim3(N), 0, stream,       \
                     (const cf*)colin, (cf*)work, ntile, scale, tw, (const cf*)propagator)
    if (det == 128 && adjoint)
      TK_FCN(128, true, 16);
    else if (det == 128)
      TK_FCN(128, false, 16);
    else if (adjoint)
      TK_FCN(512, true, 1);
    else
      TK_FCN(512, false, 1);
#undef TK_FCN
    TK_LAUNCH_CHECK();
    return TK_OK;
  }
  // few tiles: the 16 values of k1 of a tile are independent work items
  int ksplit = 1;
  while (ksplit < 16 && ntile * ksplit < 2048) ksplit *= 2;
  const dim3 grid(tk_grid(ntile * ksplit, 4)), block(256);
  if (adjoint)
    hipLaunchKernelGGL((fresnel_colpass_kernel<true>), grid, block, 0, stream, (const cf*)colin,
                       (cf*)work, ntile, scale, tw, (const cf*)propagator, ksplit);
  else
    hipLaunchKernelGGL((fresnel_colpass_kernel<false>), grid, block, 0, stream, (const cf*)colin,
                       (cf*)work, ntile, scale, tw, (const cf*)propagator, ksplit);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_ifft2_pass2_products(void* work, const void* psi, const float* scan,
                                         const void* probe, int probe_per_scan, void* objproj,
                                         void* probe_numerator, float numerator_scale,
                                         void* chi0, int keep_chi, int nscan, int S, int det,
                                         int H, int W, float inv_scale, void* stream) {
  TK_ENTER();
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && det >= 1 && H >= 1 && W >= 1);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(work && psi && scan && probe && objproj);
  if ((det != 128 && det != 256 && det != 512) || S > 8) return TK_ERR_UNSUPPORTED;
  return tk_ifft2_pass2_products((cf*)work, (const cf*)psi, scan, (const cf*)probe,
                                 probe_per_scan, (cf*)objproj, (float*)probe_numerator,
                                 numerator_scale, (cf*)chi0, keep_chi ? 1 : 2, nscan, S, det, H,
                                 W, inv_scale, (hipStream_t)stream);
}

// Tile-level 2-D FFT, version 2: strided two-pass decomposition of the
// column transform (N = 16 * RB, N in {128, 256, 512}).
//
// With y = r + RB*y2 (r < RB, y2 < 16) and ky = k1 + 16*k2 (k1 < 16, k2 < RB)
//   w_N^(y*ky) = w_16^(y2*k1) * w_N^(r*k1) * w_RB^(r*k2),
// so the column DFT splits into
//   pass 1 (fixed r):  the 16 rows {r + RB*y2} get their row FFTs (length N,
//       radix-16 Stockham in registers, the inter-stage exchange stays inside
//       a wave), are transposed through LDS so that one thread owns one column
//       of those 16 rows, take a radix-16 DFT over y2 and the twiddle
//       w_N^(r*k1) (uniform per workgroup iteration -> scalar registers), and
//       are stored as the 16 CONSECUTIVE rows 16*r + k1 of the intermediate;
//   pass 2 (fixed k1): one thread per column loads rows {16*r + k1}, does a
//       radix-RB DFT over r in registers and stores rows {k1 + 16*k2} -- the
//       same set of rows, so pass 2 runs in place; no LDS, no barriers.
// Every global access of both passes is a whole contiguous row (N*8 bytes
// per workgroup), one workgroup barrier pair per 16 rows instead of four.
// Pass 1 is NOT in place (it reads rows r + RB*y2 and writes rows 16*r + k1).
#pragma once

#include <type_traits>
#include <utility>

#include "fft_engine.h"

// Compile-time loop: f(std::integral_constant<int, I>{}) for I in [0, NN).
// Used where a caller-side register array must be indexed by a constant
// (runtime-indexed arrays are placed in scratch memory).
template <class F, int... I>
__device__ __forceinline__ void tk_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int NN, class F>
__device__ __forceinline__ void tk_static_for(F&& f) {
  tk_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, NN>{});
}

// minimum waves per SIMD requested from the register allocator for the v2
// kernels (N threads per workgroup): tunable at build time.
#ifndef TK_V2_WAVES
#define TK_V2_WAVES 3
#endif
#define TK_V2_MINW(N) ((N) <= 256 ? TK_V2_WAVES : 2)

template <int N>
struct Fft2Geom {
  using G = FftGeom<N>;
  static constexpr int RA = 16;      // rows per pass-1 iteration
  static constexpr int RB = N / 16;  // radix of pass 2
  static constexpr int NT = N;       // one thread per column
  static constexpr int T = N / 16;   // threads per row in the row FFT
  static constexpr int LS = N + N / 16 + 1;
  static constexpr int LDS_ELEMS = RA * LS;
  static_assert(N == 128 || N == 256 || N == 512, "unsupported size");
  static_assert(FftPlan<N>::E == 16, "row plan must hold 16 elements per thread");
};

// Row-FFT stage with wave-level synchronisation: a line's T <= 32 threads
// always sit in one wave (lanes j = tid % T), LDS instructions of a wave
// execute in order, so no workgroup barrier is needed between the stages.
// Inter-stage twiddles kept in LDS instead of registers: they depend only on
// (stage, slot, j), i.e. are the same for every tile and every line, so one
// table per workgroup ((NST-1)*16*T entries) filled once per kernel replaces
// 30-46 VGPRs per thread.  Reads are conflict-free (consecutive j) and
// broadcast across the lines of a wave.
template <int N>
struct FftTwLds {
  static constexpr int T = N / 16;
  static constexpr int ELEMS = (FftPlan<N>::NST - 1) * 16 * T;
  const cf* tab;  // LDS
  int j;
  __device__ __forceinline__ cf get(int s, int i) const { return tab[((s - 1) * 16 + i) * T + j]; }
  // fill (all threads of the workgroup), caller barriers afterwards
  static __device__ __forceinline__ void fill(cf* tab, const cf* __restrict__ g_tw) {
    using G = FftGeom<N>;
    using P = FftPlan<N>;
    for (int idx = threadIdx.x; idx < ELEMS; idx += blockDim.x) {
      const int jj = idx % T;
      const int i = (idx / T) % 16;
      const int s = idx / (16 * T) + 1;
      const int R = P::R[s], B = 16 / R, Ns = G::ns(s);
      const int b = i % B, r = i / B;
      const int k = (jj + b * T) & (Ns - 1);
      tab[idx] = g_tw[N + k * r * (N / (Ns * R))];
    }
  }
};

template <int N>
struct FftTwReg {
  const FftTw<N>& tw;
  __device__ __forceinline__ cf get(int s, int i) const { return tw.w[s][i]; }
};

template <int N, bool INV, int S>
struct FftStageWave {
  using G = FftGeom<N>;
  using P = FftPlan<N>;
  static constexpr int T = N / 16;
  // `at(e)`: LDS offset (from lbase) of element e of the line the T threads
  // share -- tk_pad16 for a contiguous row, a strided map for a tile column
  template <class Tw, class At>
  static __device__ __forceinline__ void run_at(cf (&v)[16], cf* __restrict__ lbase, int j,
                                                const Tw& tw, const At& at) {
    constexpr int R = P::R[S];
    constexpr int B = 16 / R;
    constexpr int Ns = G::ns(S);
    constexpr bool LAST = (S == P::NST - 1);
#pragma unroll
    for (int b = 0; b < B; ++b) {
      cf u[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        u[r] = v[b + r * B];
        if (S > 0 && r > 0) u[r] = mul_tw<INV>(u[r], tw.get(S, b + r * B));
      }
      Dft<R, INV>::run(u);
      if (LAST) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[b + r * B] = u[r];
      } else {
        const int jj = j + b * T;
        const int k = jj & (Ns - 1);
        const int j0 = (jj - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) lbase[at(j0 + r * Ns)] = u[r];
      }
    }
    if constexpr (!LAST) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = lbase[at(j + i * T)];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      FftStageWave<N, INV, S + 1>::run_at(v, lbase, j, tw, at);
    }
  }
  template <class Tw>
  static __device__ __forceinline__ void run(cf (&v)[16], cf* __restrict__ lbase, int j,
                                             const Tw& tw) {
    run_at(v, lbase, j, tw, [](int e) { return tk_pad16(e); });
  }
};

// Pass 1 for the 16 rows {r + RB*y2}.  `load(y, e, i)` returns input element
// e = j + i*T of row y (i = the caller's register slot, a compile-time index); results go to rows 16*r + k1 of `mid` (row-major N x N tile).
// Contains two workgroup barriers; every thread of the NT = N workgroup calls.
// STREAM: the intermediate is consumed by a LATER kernel (not by pass 2 of this
// one), so it is stored with the non-temporal hint and does not displace the
// probe / object lines the next tiles re-read from L2.
template <int N, bool INV, bool STREAM = false, class Tw, class Load>
__device__ __forceinline__ void fft2_pass1(cf* __restrict__ lds, const cf* __restrict__ twtab,
                                           const Tw& tw, int line, int j, int r, Load&& load,
                                           cf* __restrict__ mid) {
  using G2 = Fft2Geom<N>;
  cf v[16];
  const int y = r + G2::RB * line;
  tk_static_for<16>([&](auto I) {
    constexpr int i = decltype(I)::value;  // register slot, a constant expression
    v[i] = load(y, j + i * G2::T, I);
  });
  cf* lbase = lds + line * G2::LS;
  FftStageWave<N, INV, 0>::run(v, lbase, j, tw);
  // natural-order row spectrum -> LDS [line][e]
#pragma unroll
  for (int i = 0; i < 16; ++i) lbase[tk_pad16(j + i * G2::T)] = v[i];
  __syncthreads();
  // thread t owns column t of the 16 rows
  const int t = threadIdx.x;
#pragma unroll
  for (int y2 = 0; y2 < 16; ++y2) v[y2] = lds[y2 * G2::LS + tk_pad16(t)];
  __syncthreads();
  Dft<16, INV>::run(v);
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) {
    cf o = v[k1];
    if (k1 > 0) o = mul_tw<INV>(o, twtab[N + r * k1]);  // uniform address -> scalar load
    if (STREAM)
      tk_st_stream(mid + (16 * r + k1) * N + t, o);
    else
      mid[(16 * r + k1) * N + t] = o;
  }
}

// Pass 2 for one k1: in-place radix-RB over rows {16*r + k1} -> {k1 + 16*k2}.
// `store(ky, t, value)` receives the final element of row ky, column t.
template <int N, bool INV, class Store>
__device__ __forceinline__ void fft2_pass2(const cf* __restrict__ mid, int k1, Store&& store) {
  using G2 = Fft2Geom<N>;
  const int t = threadIdx.x;
  cf u[G2::RB];
#pragma unroll
  for (int r = 0; r < G2::RB; ++r) u[r] = mid[(16 * r + k1) * N + t];
  Dft<G2::RB, INV>::run(u);
#pragma unroll
  for (int k2 = 0; k2 < G2::RB; ++k2) store(k1 + 16 * k2, t, u[k2]);
}

// Column-first counterpart of fft2_pass1 (16 rows of T = N / 16 threads, the
// N threads of the workgroup): every thread t holds a[ya], ya < 16 -- column t of 16 rows whose
// COLUMN stage is already done -- and the rows still need their length-N
// transform.  The values are transposed through LDS into the row layout
// (16 threads per row, element e = j + i*T), transformed with the in-wave
// Stockham stages and stored as the 16 consecutive rows starting at `rows`.
// Contains two workgroup barriers; every thread of the workgroup calls.
template <int N, bool INV, bool STREAM = false, class Tw>
__device__ __forceinline__ void fft2_rows_from_columns(cf* __restrict__ lds, const Tw& tw, int line,
                                                       int j, cf (&a)[16], cf* __restrict__ rows) {
  using G2 = Fft2Geom<N>;
  const int t = threadIdx.x;
#pragma unroll
  for (int ya = 0; ya < 16; ++ya) lds[ya * G2::LS + tk_pad16(t)] = a[ya];
  __syncthreads();
  cf v[16];
  cf* lbase = lds + line * G2::LS;
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = lbase[tk_pad16(j + i * G2::T)];
  FftStageWave<N, INV, 0>::run(v, lbase, j, tw);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (STREAM)
      tk_st_stream(rows + line * N + j + i * G2::T, v[i]);
    else
      rows[line * N + j + i * G2::T] = v[i];
  }
  __syncthreads();
}

// ---- a whole 128 x 128 tile in LDS (fwd128_lds_kernel of ptycho.hip, the
// prime-factor sub-tile kernel of pfa.hip)
constexpr int TK_L128_LS = 136;
// Column offset of row `row` inside the LDS tile (element (row, col) lives at
// row * LS + ((col + swizzle(row)) & 127)), chosen so that BOTH transforms run
// on the tile without bank conflicts, dword bank = 16 row + 2 col' (mod 64):
//   rows     a 32-lane group = 4 consecutive rows x 8 lanes j: the rows share
//            the swizzle, 16 row covers the four 16-bank quarters;
//   columns  lanes j read rows j + 8 i (bit 2 of the row separates j from
//            j + 4: + 8 banks) and exchange through rows 16 j + r (bits 4..6 of
//            the row = j: + 8 j banks); 4 neighbouring columns fill the 8
//            banks in between.
__device__ __forceinline__ int tk_l128_swizzle(int row) {
  return 4 * (((row >> 2) & 1) + ((row >> 4) & 7));
}


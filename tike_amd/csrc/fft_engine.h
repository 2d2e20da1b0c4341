// Workgroup-level batched 1-D FFT engine for gfx950 (power-of-two lengths).
//
// Stockham autosort, radix <= 16 butterflies in registers, one LDS exchange
// between stages.  A "line" is one 1-D transform of length N handled by
// T = N/E threads, each owning E elements (element index j + i*T, i < E, in
// every stage -- see DESIGN.md "FFT engine").  A workgroup of NT threads
// processes L = NT/T lines per call.  Loads/stores of the first/last stage go
// through caller functors so that producers (patch gather * probe) and
// consumers (|.|^2, crop, scaling) fuse into the transform.
//
// Two lane layouts:
//   ROW: j = tid % T (consecutive lanes walk along the line; a line's
//        elements are contiguous in memory),
//   COL: line = tid % L (consecutive lanes are consecutive lines; used when
//        lines are columns of a row-major tile so that global accesses are
//        L*8-byte contiguous segments).
#pragma once

#include "common.h"

// Twiddle tables: exp(-2*pi*i*k/N) for N = 32..1024 stored at offset N of one
// 2048-entry device buffer, created once per device by tk_twiddles()
// (fft2.hip) and passed to kernels as an argument.
const cf* tk_twiddles();  // host: table of the calling thread's current device

template <int N>
struct FftPlan;
template <>
struct FftPlan<32> {
  static constexpr int E = 4, NST = 3, NT = 256, MINW = 4;
  static constexpr int R[3] = {4, 4, 2};
};
template <>
struct FftPlan<64> {
  static constexpr int E = 8, NST = 2, NT = 256, MINW = 4;
  static constexpr int R[3] = {8, 8, 1};
};
template <>
struct FftPlan<128> {
  static constexpr int E = 16, NST = 2, NT = 256, MINW = 3;
  static constexpr int R[3] = {16, 8, 1};
};
template <>
struct FftPlan<256> {
  static constexpr int E = 16, NST = 2, NT = 256, MINW = 3;
  static constexpr int R[3] = {16, 16, 1};
};
template <>
struct FftPlan<512> {
  static constexpr int E = 16, NST = 3, NT = 512, MINW = 2;
  static constexpr int R[3] = {16, 16, 2};
};
template <>
struct FftPlan<1024> {
  static constexpr int E = 16, NST = 3, NT = 1024, MINW = 4;
  static constexpr int R[3] = {16, 16, 4};
};

template <int N>
struct FftGeom {
  using P = FftPlan<N>;
  static constexpr int E = P::E;
  static constexpr int T = N / E;          // threads per line
  static constexpr int NT = P::NT;         // threads per workgroup
  static constexpr int L = NT / T;         // lines per call
  static constexpr int LS = N + N / 16 + 1;  // LDS line stride (elements)
  static constexpr int LDS_ELEMS = L * LS;
  static constexpr int ns(int s) {  // product of radices before stage s
    int n = 1;
    for (int i = 0; i < s; ++i) n *= P::R[i];
    return n;
  }
  static_assert(L <= N, "a workgroup call must not span more than one tile");
};

__device__ __forceinline__ int tk_pad16(int i) { return i + (i >> 4); }

template <int N, bool COL>
struct FftLane {
  int line, j;
  __device__ __forceinline__ FftLane() {
    using G = FftGeom<N>;
    if (COL) {
      line = threadIdx.x % G::L;
      j = threadIdx.x / G::L;
    } else {
      j = threadIdx.x % G::T;
      line = threadIdx.x / G::T;
    }
  }
};

// Lane coordinates made opaque to the optimiser.  Kernels call this at the
// start of every pass so that per-pass invariants (LDS/global offsets,
// twiddles) are recomputed per tile instead of being hoisted out of the tile
// loop, where the row-pass and column-pass sets would be live together and
// double the register footprint (244 -> ~125 VGPRs for N = 256).
template <int N, bool COL>
__device__ __forceinline__ FftLane<N, COL> fft_lane() {
  FftLane<N, COL> l;
  asm volatile("" : "+v"(l.line), "+v"(l.j));
  return l;
}

// Per-thread inter-stage twiddles (registers).  w[s][i] multiplies register
// slot i = b + r*B before the radix-R butterflies of stage s >= 1.
template <int N>
struct FftTw {
  cf w[FftPlan<N>::NST][FftPlan<N>::E];
  __device__ __forceinline__ void init(const cf* __restrict__ g_tw, int j) {
    using G = FftGeom<N>;
    using P = FftPlan<N>;
#pragma unroll
    for (int s = 1; s < P::NST; ++s) {
      const int R = P::R[s], B = G::E / R, Ns = G::ns(s);
#pragma unroll
      for (int i = 0; i < G::E; ++i) {
        const int b = i % B, r = i / B;
        const int k = (j + b * G::T) & (Ns - 1);
        w[s][i] = g_tw[N + k * r * (N / (Ns * R))];
      }
    }
  }
};

template <int N, bool INV, int S>
struct FftStage {
  using G = FftGeom<N>;
  using P = FftPlan<N>;
  static __device__ __forceinline__ void run(cf (&v)[G::E], cf* __restrict__ lbase, int j,
                                             const FftTw<N>& tw) {
    constexpr int R = P::R[S];
    constexpr int B = G::E / R;
    constexpr int Ns = G::ns(S);
    constexpr bool LAST = (S == P::NST - 1);
#pragma unroll
    for (int b = 0; b < B; ++b) {
      cf u[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        u[r] = v[b + r * B];
        if (S > 0 && r > 0) u[r] = mul_tw<INV>(u[r], tw.w[S][b + r * B]);
      }
      Dft<R, INV>::run(u);
      if (LAST) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[b + r * B] = u[r];
      } else {
        const int jj = j + b * G::T;
        const int k = jj & (Ns - 1);
        const int j0 = (jj - k) * R + k;  // (jj / Ns) * Ns * R + k
#pragma unroll
        for (int r = 0; r < R; ++r) lbase[tk_pad16(j0 + r * Ns)] = u[r];
      }
    }
    if constexpr (!LAST) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < G::E; ++i) v[i] = lbase[tk_pad16(j + i * G::T)];
      __syncthreads();
      FftStage<N, INV, S + 1>::run(v, lbase, j, tw);
    }
  }
};

// Transform the L lines owned by this workgroup call.
//   load(line, e)  -> cf   element e of local line `line`
//   store(line, e, cf)     natural-order output element e
// Every thread of the workgroup must call this (it contains barriers).
// LOAD_CHUNK > 0 fences the scheduler every LOAD_CHUNK elements of the load
// phase: for expensive loaders (bilinear gather * probe) this bounds the
// loads in flight per thread and with it the register footprint.
// sync_after_load: the loader reads the LDS line buffers themselves (inputs
// staged there by the caller), so a barrier must separate those reads from
// the first inter-stage write.
template <int N, bool INV, bool COL, int LOAD_CHUNK = 0, class Load, class Store>
__device__ __forceinline__ void fft_lines(cf* __restrict__ lds, const FftLane<N, COL>& ln,
                                          const FftTw<N>& tw, Load&& load, Store&& store,
                                          bool sync_after_load = false) {
  using G = FftGeom<N>;
  cf v[G::E];
#pragma unroll
  for (int i = 0; i < G::E; ++i) {
    v[i] = load(ln.line, ln.j + i * G::T);
    if (LOAD_CHUNK > 0 && (i % (LOAD_CHUNK > 0 ? LOAD_CHUNK : 1)) == LOAD_CHUNK - 1)
      __builtin_amdgcn_sched_barrier(0);
  }
  if (sync_after_load) __syncthreads();
  FftStage<N, INV, 0>::run(v, lds + ln.line * G::LS, ln.j, tw);
#pragma unroll
  for (int i = 0; i < G::E; ++i) store(ln.line, ln.j + i * G::T, v[i]);
}

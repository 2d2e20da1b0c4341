// Shape-general line-FFT engine for gfx950 (round 6): mixed-radix Stockham
// autosort in LDS, line length n = product of radices from {2,...,8,10,...,13,16,20,24}
// chosen at run time by a host plan (MixPlan).  It is what serves every
// detector size the register engines (fft_engine.h / fft_engine2.h: powers of
// two, 32..1024) do not -- the reference hands any shape to cuFFT
// (operators/cupy/cache.py:32-82, propagation.py:43-73).
//
// A workgroup holds `nlines` lines in LDS (two buffers, line stride p.ls).  One
// stage = every butterfly of every line: thread t takes butterflies t, t + NT,
// ...; a butterfly reads its R inputs from buffer a (stride n / R: consecutive
// lanes read consecutive elements), multiplies by the twiddles w_n^(k r step)
// from an LDS table of the n-th roots of unity, runs the radix-R DFT in
// registers (fft_radix.h) and writes natural (autosorted) positions of buffer
// b.  One barrier per stage.  Loads and stores of the lines go through caller
// functors, so producers (patch gather x probe) and consumers (intensity,
// crop, products) fuse into the transform, as with the register engines.
//
// Index arithmetic by float reciprocals (exact for the ranges asserted by the
// planner: quotients < 2^8, divisors <= 4096): gfx950 has no integer divide.
//
// mix_butterfly / MixPlan are plain C++: tests/csrc/test_fft_mixed.cpp runs the
// stages on the host against a float64 DFT for every planned size.
#pragma once

#if defined(__HIPCC__)
#include "common.h"  // hip_runtime.h in front of fft_radix.h
#else
#include "fft_radix.h"
#endif

#define TK_MIX_MAX_STAGES 12
#define TK_MIX_MAX_N 4096

struct MixPlan {
  int n;    // line length
  int nst;  // number of stages
  int ls;   // LDS line stride in elements (padded)
  int radix[TK_MIX_MAX_STAGES];
  int step[TK_MIX_MAX_STAGES];  // n / (product of the radices up to and incl. the stage)
};

// LDS position of element i of a line: one slot of padding per 16 elements, so
// that the stride-R writes of the early stages spread over the banks.
TK_HD int mix_pad(int i) { return i + (i >> 4); }
TK_HD int mix_line_stride(int n) { return n + (n >> 4) + 1; }

// q = a / d for 0 <= a < 2^16, 1 <= d <= 4096, a / d < 2^8 (see header)
TK_HD int mix_div(int a, float rcp_d) { return (int)(((float)a + 0.5f) * rcp_d); }

// Plan for n (host): the FEWEST stages with radices from {2, 3, 4, 5, 6, 7, 8,
// 10, 11, 12, 13, 16, 20, 24} (a stage is an LDS round trip and a barrier, and
// the index arithmetic of a butterfly is shared by its R elements: 384 = 24 x
// 16 runs 1.4x faster than 3 x 8 x 16), ties to the smaller sum of radices;
// larger radices first.  Returns false when n has a prime factor above 13
// (the caller then takes Bluestein's route over a power of two).
static inline int mix_plan_search(int n, int* out) {
  static const int allowed[14] = {24, 20, 16, 13, 12, 11, 10, 8, 7, 6, 5, 4, 3, 2};
  if (n == 1) return 0;
  int best = -1, best_sum = 0, tmp[TK_MIX_MAX_STAGES], keep[TK_MIX_MAX_STAGES];
  for (int r : allowed) {
    if (n % r) continue;
    const int c = mix_plan_search(n / r, tmp);
    if (c < 0 || c + 1 > TK_MIX_MAX_STAGES) continue;
    int sum = r;
    for (int i = 0; i < c; ++i) sum += tmp[i];
    if (best < 0 || c + 1 < best || (c + 1 == best && sum < best_sum)) {
      best = c + 1;
      best_sum = sum;
      keep[0] = r;
      for (int i = 0; i < c; ++i) keep[i + 1] = tmp[i];
    }
  }
  for (int i = 0; i < best; ++i) out[i] = keep[i];
  return best;
}

static inline bool mix_make_plan(int n, MixPlan* p) {
  if (n < 1 || n > TK_MIX_MAX_N) return false;
  p->n = n;
  p->ls = mix_line_stride(n);
  int m = n;
  for (int f : {2, 3, 5, 7, 11, 13})
    while (m % f == 0) m /= f;
  if (m != 1) return false;
  p->nst = mix_plan_search(n, p->radix);
  if (p->nst < 0) return false;
  // larger radices first (the first stage multiplies by no twiddles)
  for (int i = 0; i < p->nst; ++i)
    for (int j = i + 1; j < p->nst; ++j)
      if (p->radix[j] > p->radix[i]) {
        const int t = p->radix[i];
        p->radix[i] = p->radix[j];
        p->radix[j] = t;
      }
  if (p->nst == 0) p->radix[p->nst++] = 1;  // n == 1
  // (no integer division in the kernels: gfx950 has none in hardware)
  for (int i = 0, ns = 1; i < p->nst; ++i) {
    ns *= p->radix[i];
    p->step[i] = n / ns;
  }
  return true;
}

// One radix-R butterfly of a line: a, b = the line's two LDS buffers, tw = n-th
// roots of unity (forward values), nb = n / R, Ns = product of the radices of
// the stages in front, step = n / (Ns * R).
template <int R, bool INV>
TK_HD void mix_butterfly(const cf* __restrict__ a, cf* __restrict__ b, const cf* __restrict__ tw,
                         int nb, int Ns, int step, float rcp_ns, int jj) {
  const int q = mix_div(jj, rcp_ns), k = jj - q * Ns;
  cf u[R];
#pragma unroll
  for (int r = 0; r < R; ++r) u[r] = a[mix_pad(jj + r * nb)];
  if (Ns > 1) {
    const int ks = k * step;
#pragma unroll
    for (int r = 1; r < R; ++r) u[r] = mul_tw<INV>(u[r], tw[ks * r]);
  }
  Dft<R, INV>::run(u);
  const int j0 = q * Ns * R + k;
#pragma unroll
  for (int r = 0; r < R; ++r) b[mix_pad(j0 + r * Ns)] = u[r];
}

template <bool INV>
struct Dft<1, INV> {
  static TK_HD void run(cf*) {}
};

#if defined(__HIPCC__)
// Every butterfly of stage (R, Ns) over `nlines` lines; no barrier inside.
template <int R, bool INV>
__device__ __forceinline__ void mix_stage(const cf* __restrict__ a, cf* __restrict__ b,
                                          const cf* __restrict__ tw, int n, int ls, int Ns,
                                          int step, int nlines) {
  const int nb = n / R;  // (R is a constant: a multiply and a shift)
  const float rcp_nb = 1.0f / (float)nb, rcp_ns = 1.0f / (float)Ns;
  const int total = nlines * nb;
  // few butterflies per stage (4 lines x 16 of radix 24) fill only the first
  // wave(s) of a workgroup: with several 256-thread workgroups per CU, start
  // at a wave that depends on the workgroup and on the stage (2-D transform
  // at 192^2 +11 %, 384^2 +6 %; the one-workgroup-per-CU column kernel of
  // general.hip loses by it and keeps the plain order)
  const int first = blockDim.x == 256
                        ? (threadIdx.x + 64 * ((blockIdx.x + Ns) & 3)) & 255
                        : threadIdx.x;
  for (int t = first; t < total; t += blockDim.x) {
    const int line = mix_div(t, rcp_nb), jj = t - line * nb;
    mix_butterfly<R, INV>(a + line * ls, b + line * ls, tw, nb, Ns, step, rcp_ns, jj);
  }
}

// All stages of the plan over `nlines` lines held in buffer a (line stride
// p.ls).  Every thread of the workgroup must call this (a barrier closes every
// stage; the caller's loads must be visible on entry: barrier before).
// Returns the buffer that holds the natural-order result (a or b).
template <bool INV>
__device__ __forceinline__ cf* mix_stages(cf* a, cf* b, const cf* __restrict__ tw,
                                          const MixPlan& p, int nlines) {
  int Ns = 1;
  for (int s = 0; s < p.nst; ++s) {
    const int R = p.radix[s];
    switch (R) {
      case 2: mix_stage<2, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
      case 3: mix_stage<3, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
      case 4: mix_stage<4, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
      case 5: mix_stage<5, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
      case 6: mix_stage<6, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
      case 7: mix_stage<7, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
      case 8: mix_stage<8, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
      case 10: mix_stage<10, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
      case 11: mix_stage<11, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
      case 12: mix_stage<12, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
      case 13: mix_stage<13, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
      case 16: mix_stage<16, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
      case 20: mix_stage<20, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
      case 24: mix_stage<24, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
      default: mix_stage<1, INV>(a, b, tw, p.n, p.ls, Ns, p.step[s], nlines); break;
    }
    __syncthreads();
    cf* t = a;
    a = b;
    b = t;
    Ns *= R;
  }
  return a;
}

// Host side of the engine (fft2.hip): plans and their device tables, cached per
// (n, device) like the reference's cuFFT plans (cache.py:32-46).
struct MixTables {
  MixPlan plan;      // stages of the transform of length plan.n (= n, or the
                     // Bluestein length M when n has a large prime factor)
  int n;             // the caller's line length
  bool bluestein;
  const cf* tw;      // plan.n-th roots of unity, forward values
  const cf* chirp;   // Bluestein: exp(-i pi j^2 / n), j < n
  const cf* bhat;    // Bluestein: FFT_M of the wrapped conjugate chirp, / M
};
const MixTables* tk_mix_tables(int n);  // nullptr: unsupported size
#endif

// Shared device/host helpers for the tike_amd HIP library (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "fft_radix.h"

#define TK_OK 0
#define TK_ERR_ARG 1000001      // bad argument (shape relation violated)
#define TK_ERR_UNSUPPORTED 1000002

#define TK_CHECK_ARG(cond) \
  do {                     \
    if (!(cond)) return TK_ERR_ARG; \
  } while (0)

// hipGetLastError() is per-thread state shared with every other HIP user in the
// process (PyTorch included): clear stale errors on entry so that a launch
// check only ever reports this library's own launches.
#define TK_ENTER() (void)hipGetLastError()

#define TK_LAUNCH_CHECK()                   \
  do {                                      \
    hipError_t e__ = hipGetLastError();     \
    if (e__ != hipSuccess) return (int)e__; \
  } while (0)

static inline bool tk_is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// Number of persistent workgroups for grid-stride kernels: enough to fill
// 256 CUs several times over, capped by the amount of work.
static inline int tk_grid(long work_items, int per_cu = 8) {
  const long cap = 256L * per_cu;
  long g = work_items < cap ? work_items : cap;
  return (int)(g < 1 ? 1 : g);
}

// Streaming (read-once / write-once) accesses: the non-temporal hint keeps
// them from displacing the lines that ARE re-read soon (the FFT intermediate,
// the probe, the object window) in the 4 MiB L2 of the XCD.
__device__ __forceinline__ cf tk_ld_stream(const cf* p) {
  const double d = __builtin_nontemporal_load(reinterpret_cast<const double*>(p));
  return __builtin_bit_cast(cf, d);
}
__device__ __forceinline__ void tk_st_stream(cf* p, cf v) {
  __builtin_nontemporal_store(__builtin_bit_cast(double, v), reinterpret_cast<double*>(p));
}
__device__ __forceinline__ void tk_st_stream(float* p, float v) {
  __builtin_nontemporal_store(v, p);
}

// Element `byte_off` bytes past a UNIFORM base pointer: written so that the
// compiler selects the scalar-base addressing mode (SGPR pair + one 32-bit
// VGPR offset shared by all rows of the slice) instead of a 64-bit address
// pair per row.
template <class T>
__device__ __forceinline__ const T* tk_at(const T* base, unsigned byte_off) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ T* tk_at(T* base, unsigned byte_off) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off);
}
// The same with the base pinned in scalar registers first: without it the
// compiler folds (base + lane offset) once and then adds the row offsets as
// 64-bit VECTOR additions (two instructions, a hazard nop and a register pair
// per row).
template <class T>
__device__ __forceinline__ T* tk_at_pinned(T* base, unsigned byte_off) {
  asm volatile("" : "+s"(base));
  return tk_at(base, byte_off);
}

__device__ __forceinline__ float tk_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for 256-thread blocks; result valid in every thread.
// `red` must hold >= 4 floats of LDS; contains barriers.
__device__ __forceinline__ float tk_block_sum256(float v, float* red) {
  v = tk_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// The same for 512-thread blocks (`red`: >= 8 floats), a fixed tree.
__device__ __forceinline__ float tk_block_sum512(float v, float* red) {
  v = tk_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

// Bilinear weights and integer corner of a scan position (y, x), as the
// reference computes them (convolution.cu:101-134).
struct TkCorner {
  int sy, sx;
  float w00, w01, w10, w11;  // (1-fx)(1-fy), fx(1-fy), (1-fx)fy, fx*fy
};

__device__ __forceinline__ TkCorner tk_corner(const float* __restrict__ scan, long n) {
  TkCorner c;
  const float y = scan[2 * n], x = scan[2 * n + 1];
  const float fy0 = floorf(y), fx0 = floorf(x);
  const float fy = y - fy0, fx = x - fx0;
  c.sy = (int)fy0;
  c.sx = (int)fx0;
  c.w00 = (1.0f - fx) * (1.0f - fy);
  c.w01 = fx * (1.0f - fy);
  c.w10 = (1.0f - fx) * fy;
  c.w11 = fx * fy;
  return c;
}

// Bilinear gather of image pixel (sy+py, sx+px) with linear addressing of the
// three trailing taps, skipping zero-weight and out-of-allocation taps
// (reference convolution.cu:35-48).  `total` = elements in the
// allocation starting at img.  Caller guarantees the leading tap is in bounds.
__device__ __forceinline__ cf tk_gather(const cf* __restrict__ img, long ii, int W, long total,
                                        const TkCorner& c) {
  // Branch-free: the four loads are issued back to back (a branch per tap
  // serialises four memory latencies).  Taps outside the allocation are read
  // from a clamped address and weighted by exactly zero; zero-weight taps
  // inside the allocation contribute exactly zero, as in the reference.
  const long last = total - 1;
  const long i1 = ii + 1 <= last ? ii + 1 : last;
  const long i2 = ii + W <= last ? ii + W : last;
  const long i3 = ii + W + 1 <= last ? ii + W + 1 : last;
  const float w1 = ii + 1 <= last ? c.w01 : 0.0f;
  const float w2 = ii + W <= last ? c.w10 : 0.0f;
  const float w3 = ii + W + 1 <= last ? c.w11 : 0.0f;
  const cf a = img[ii], b = img[i1], d = img[i2], e = img[i3];
  cf r = mk(a.x * c.w00, a.y * c.w00);
  r.x += b.x * w1;
  r.y += b.y * w1;
  r.x += d.x * w2;
  r.y += d.y * w2;
  r.x += e.x * w3;
  r.y += e.y * w3;
  return r;
}

// Probe at a scan position: either an explicit array (shared, or one per
// position) or the shared probe plus eigen probes weighted per position
// (reference probe.py:272-303 get_varying_probe, synthesised on the fly).
struct TkProbe {
  const cf* probe;       // (1|N, S, pw, pw)
  long pos_stride;       // 0 when shared, S*pw*pw when per position
  const cf* eigen;       // (C, Sm, pw, pw) or nullptr
  const float* weights;  // (N, C+1, S) or nullptr; row 0 scales the shared probe
  const cf* unique;      // (N, Sm, pw, pw) or nullptr: the varying probe of the
                         // first Sm modes already synthesised (tike_varying_probe)
  int C, Sm, S, pw;

  __device__ __forceinline__ cf at(long n, int s, long pix) const {
    const long pp = (long)pw * pw;
    if (weights == nullptr) return probe[n * pos_stride + s * pp + pix];
    if (unique != nullptr && s < Sm) return unique[(n * Sm + s) * pp + pix];
    const float* w = weights + n * (long)(C + 1) * S;
    cf v = probe[s * pp + pix] * w[s];
    if (eigen != nullptr && s < Sm) {
      for (int c = 0; c < C; ++c) {
        const cf e = eigen[((long)c * Sm + s) * pp + pix];
        const float wc = w[(c + 1) * S + s];
        v.x += wc * e.x;
        v.y += wc * e.y;
      }
    }
    return v;
  }
};

static inline TkProbe tk_make_probe(const void* probe, int probe_per_scan, const void* eigen,
                                    const float* weights, int C, int Sm, int S, int pw,
                                    const void* unique = nullptr) {
  TkProbe p;
  p.unique = (const cf*)unique;
  p.probe = (const cf*)probe;
  p.pos_stride = probe_per_scan ? (long)S * pw * pw : 0L;
  p.eigen = (const cf*)eigen;
  p.weights = weights;
  p.C = C;
  p.Sm = Sm;
  p.S = S;
  p.pw = pw;
  return p;
}

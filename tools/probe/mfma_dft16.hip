// The MFMA experiment SURVEY section 7 / north_star ask for: one radix-16 DFT stage
// (what every FFT kernel here is made of) as a matrix product on the matrix
// cores instead of the in-register butterfly on the vector ALU.
//
//   Y (16 x B complex) = W16 (16 x 16 complex) X   ->   real-ified
//   [Yr ; Yi] (32 x B) = [[Wr, -Wi], [Wi, Wr]] (32 x 32) [Xr ; Xi] (32 x B)
//
// = 16 x v_mfma_f32_32x32x2_f32 per 32 vectors (f32 in, f32 accumulate: exact
// f32 FMA chains -- the only MFMA form that meets the 1e-5 parity tolerance).
// Compared with Dft<16>::run (fft_radix.h, the product's butterfly):
//   1. results of both against a float64 host DFT;
//   2. time per 16-point transform, each alone at 1 / 2 / 4 waves per SIMD;
//   3. both at once on one CU: half the waves of every SIMD run the MFMA form,
//      half the VALU form (do the two pipes overlap, and what does it buy?).
//   hipcc --offload-arch=gfx950 -O3 -I../../tike_amd/csrc mfma_dft16.hip -o mfma_dft16
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "fft_radix.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));

// A operand of step kk: lane l holds M[row = l % 32][k = 2 kk + l / 32]
__device__ __forceinline__ void load_matrix(const float* __restrict__ M, float (&a)[16]) {
  const int l = threadIdx.x & 63;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) a[kk] = M[(l & 31) * 32 + 2 * kk + (l >> 5)];
}

// 32 vectors per wave: lane (n = l % 32, h = l / 32) holds the reals k = 2 kk + h
// of vector n (k < 16: Re X[k], k >= 16: Im X[k - 16]); the result comes back as
// rows (i % 4) + 8 (i / 4) + 4 h in register i.
__device__ __forceinline__ f16v dft16_mfma(const float (&a)[16], const float (&b)[16]) {
  f16v d = {0};
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) d = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], d, 0, 0, 0);
  return d;
}

// mode 0: VALU butterflies, 1: MFMA, 2: even waves VALU / odd waves MFMA
__global__ __launch_bounds__(1024) void bench(const float* __restrict__ M, const cf* __restrict__ x,
                                              cf* __restrict__ y, float* __restrict__ ym,
                                              long long* ticks, int iters, int mode) {
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const bool use_mfma = mode == 1 || (mode == 2 && (wave & 1));
  const long gw = (long)blockIdx.x * (blockDim.x >> 6) + wave;
  long long t0, t1;
  if (use_mfma) {
    float a[16], b[16];
    load_matrix(M, a);
    const int n = l & 31, h = l >> 5;
    const cf* xv = x + (gw * 64 + n) * 16;  // vector n of this wave (first 32 used)
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int k = 2 * kk + h;
      b[kk] = k < 16 ? xv[k].x : xv[k - 16].y;
    }
    f16v d = {0};
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      d = dft16_mfma(a, b);
      // feed a result back so that nothing is hoisted (keeps magnitudes: / 4)
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) b[kk] = b[kk] * 0.75f + d[kk] * (0.25f / 4);
    }
    t1 = __builtin_amdgcn_s_memtime();
    float keep = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) keep += b[kk];
    // one clean transform for the check
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int k = 2 * kk + h;
      b[kk] = k < 16 ? xv[k].x : xv[k - 16].y;
    }
    if (keep == 1234.5f) b[0] = keep;  // (the timed loop's result stays live)
    d = dft16_mfma(a, b);
    if (iters == 0 || true) {
#pragma unroll
      for (int i = 0; i < 16; ++i) ym[(gw * 32 + n) * 32 + (i % 4) + 8 * (i / 4) + 4 * h] = d[i];
    }
  } else {
    cf v[16];
    const cf* xv = x + (gw * 64 + l) * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = xv[k];
    cf w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = v[k];
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      Dft<16, false>::run(w);
#pragma unroll
      for (int k = 0; k < 16; ++k) w[k] = mk(v[k].x * 0.75f + w[k].x * (0.25f / 4), v[k].y * 0.75f + w[k].y * (0.25f / 4));
    }
    t1 = __builtin_amdgcn_s_memtime();
    Dft<16, false>::run(v);
    float keep = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) keep += w[k].x;
    if (keep == 1234.5f) v[0].x = keep;
#pragma unroll
    for (int k = 0; k < 16; ++k) y[(gw * 64 + l) * 16 + k] = v[k];
  }
  if (l == 0) ticks[gw] = t1 - t0;
}

int main() {
  const int grid = 256, maxw = 16;
  const long nvec = (long)grid * maxw * 64;
  std::vector<cf> hx(nvec * 16);
  srand(1);
  for (auto& v : hx) v = mk(rand() / (float)RAND_MAX - 0.5f, rand() / (float)RAND_MAX - 0.5f);
  std::vector<float> hM(32 * 32);
  for (int o = 0; o < 16; ++o)
    for (int k = 0; k < 16; ++k) {
      const double ang = -2.0 * M_PI * o * k / 16.0;
      const float wr = (float)cos(ang), wi = (float)sin(ang);
      hM[o * 32 + k] = wr; hM[o * 32 + 16 + k] = -wi;
      hM[(16 + o) * 32 + k] = wi; hM[(16 + o) * 32 + 16 + k] = wr;
    }
  float *M, *ym; cf *x, *y; long long* ticks;
  CK(hipMalloc(&M, hM.size() * 4)); CK(hipMalloc(&x, hx.size() * 8)); CK(hipMalloc(&y, hx.size() * 8));
  CK(hipMalloc(&ym, nvec * 32 * 4)); CK(hipMalloc(&ticks, grid * maxw * 8));
  CK(hipMemcpy(M, hM.data(), hM.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
  // ---- 1. correctness of both forms against a float64 DFT
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(bench, dim3(grid), dim3(256), 0, 0, M, x, y, ym, ticks, 0, mode);
    CK(hipDeviceSynchronize());
    std::vector<cf> hy(hx.size()); std::vector<float> hym(nvec * 32);
    CK(hipMemcpy(hy.data(), y, hy.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hym.data(), ym, hym.size() * 4, hipMemcpyDeviceToHost));
    double err = 0, ref = 0;
    const long nw = (long)grid * 4;
    for (long w = 0; w < nw; ++w)
      for (int n = 0; n < (mode ? 32 : 64); ++n) {
        const cf* v = &hx[(w * 64 + n) * 16];
        for (int o = 0; o < 16; ++o) {
          std::complex<double> s = 0;
          for (int k = 0; k < 16; ++k)
            s += std::complex<double>(v[k].x, v[k].y) * std::polar(1.0, -2.0 * M_PI * o * k / 16.0);
          std::complex<double> got = mode ? std::complex<double>(hym[(w * 32 + n) * 32 + o], hym[(w * 32 + n) * 32 + 16 + o])
                                          : std::complex<double>(hy[(w * 64 + n) * 16 + o].x, hy[(w * 64 + n) * 16 + o].y);
          err += std::norm(got - s); ref += std::norm(s);
        }
      }
    printf("%s radix-16 vs float64 DFT: normwise relative error %.3e\n", mode ? "MFMA (32x32x2 f32)" : "VALU (Dft<16>)    ", sqrt(err / ref));
  }
  // ---- 2./3. time per transform
  printf("\n%-34s | waves/SIMD | ticks per 16-point transform per wave-lane-group | SIMD ns per transform\n", "form");
  const int iters = 2000;
  for (int mode = 0; mode < 3; ++mode)
    for (int wps : {1, 2, 4}) {
      const int threads = 256 * wps;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(bench, dim3(grid), dim3(threads), 0, 0, M, x, y, ym, ticks, iters, mode);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(bench, dim3(grid), dim3(threads), 0, 0, M, x, y, ym, ticks, iters, mode);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const int nw = grid * threads / 64;
      std::vector<long long> h(nw);
      CK(hipMemcpy(h.data(), ticks, nw * 8, hipMemcpyDeviceToHost));
      // transforms per wave and iteration: VALU 64, MFMA 32
      std::vector<double> tv, tm;
      for (int w = 0; w < nw; ++w) {
        const bool mf = mode == 1 || (mode == 2 && ((w % (threads / 64)) & 1));
        (mf ? tm : tv).push_back((double)h[w] / iters / (mf ? 32 : 64));
      }
      auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
      const double tr_per_simd = (mode == 0 ? 64.0 * wps : mode == 1 ? 32.0 * wps : (wps == 1 ? 48.0 : 48.0 * wps)) * iters;
      // (mode 2, 1 wave/SIMD: waves alternate over SIMDs, so each SIMD sees one form only)
      printf("%-34s | %d | VALU %7.2f  MFMA %7.2f | %8.3f\n",
             mode == 0 ? "VALU Dft<16> on every wave" : mode == 1 ? "MFMA on every wave" : "odd waves MFMA, even waves VALU",
             wps, med(tv), med(tm), ms * 1e6 / tr_per_simd);
    }
  return 0;
}

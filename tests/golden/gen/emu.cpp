// Host build of the reference's own patch kernel -- FIXTURE GENERATION ONLY.
//
// The reference's convolution.cu is #included by path from /root/reference
// (never copied) behind a minimal CUDA-execution-model prelude, so that the
// reference's Python (run under cupy_shim) calls the reference's own kernel
// arithmetic.  Built by make_fixtures.py into a temporary directory:
//   g++ -O2 -shared -fPIC -o libemu.so emu.cpp
// Nothing of this is used by the product, the oracle or the GPU box.
#include <cassert>
#include <cmath>
#include <cstddef>
#define __CUDA_ARCH__ 900
#define __device__
#define __global__
struct float2 { float x, y; };
struct double2 { double x, y; };
struct dim3_ { int x, y, z; };
static thread_local dim3_ blockIdx, gridDim, threadIdx, blockDim;
static inline float atomicAdd(float* a, float v) { float o = *a; *a += v; return o; }
static inline double atomicAdd(double* a, double v) { double o = *a; *a += v; return o; }
using std::floor;
#include "/root/reference/src/tike/operators/cupy/convolution.cu"

template <typename F, typename... A>
static void launch(F f, int gx, int gy, int gz, int bx, A... a) {
  gridDim = {gx, gy, gz};
  blockDim = {bx, 1, 1};
  for (int z = 0; z < gz; ++z)
    for (int y = 0; y < gy; ++y)
      for (int x = 0; x < gx; ++x)
        for (int t = 0; t < bx; ++t) {
          blockIdx = {x, y, z};
          threadIdx = {t, 0, 0};
          f(a...);
        }
}

extern "C" void fwd_patch_c64(float2* img, float2* pat, const float* scan, int nimage, int ny,
                              int nx, int nscan, int nrepeat, int pw, int padded, int bx) {
  launch(fwd_patch<float2, float2, float>, nscan, nimage, pw, bx, img, pat, scan, nimage, ny,
         nx, nscan, nrepeat, pw, padded);
}
extern "C" void adj_patch_c64(float2* img, float2* pat, const float* scan, int nimage, int ny,
                              int nx, int nscan, int nrepeat, int pw, int padded, int npatch,
                              int bx) {
  launch(adj_patch<float2, float2, float>, nscan, nimage, pw, bx, img, pat, scan, nimage, ny,
         nx, nscan, nrepeat, pw, padded, npatch);
}

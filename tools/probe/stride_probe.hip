// Does the hand-off layout cost the readers bandwidth?  The column pass reads, per
// (tile, k1), 16 rows of 2 KiB that lie 32 KiB apart; forward pass 1 writes 16
// consecutive rows (32 KiB).  Read and write streams with both shapes.
//   hipcc --offload-arch=gfx950 -O3 stride_probe.hip -o stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

// tile = 256 rows x 256 f2 (512 KiB).  STRIDED: workgroup (tile, k1) touches rows 16 r + k1;
// else rows 16 k1 + r (32 KiB contiguous).
template <bool STRIDED, bool WRITE, bool NT>
__global__ __launch_bounds__(256) void k(f2* p, long ntile, float* out) {
  float acc = 0.f;
  for (long v = blockIdx.x; v < ntile * 16; v += gridDim.x) {
    const long tile = v >> 4; const int k1 = (int)(v & 15);
    f2* t = p + tile * 65536 + threadIdx.x;
    f2 u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      f2* q = t + (STRIDED ? (16 * r + k1) : (16 * k1 + r)) * 256;
      if (WRITE) { f2 w = {1.f, (float)r}; if (NT) __builtin_nontemporal_store(w, q); else *q = w; }
      else u[r] = NT ? __builtin_nontemporal_load(q) : *q;
    }
    if (!WRITE) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc += u[r].x;
    }
  }
  if (acc == 123.456f) *out = acc;
}
template <class F> float timeit(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 5;
}
int main() {
  const long ntile = 8000; const double bytes = ntile * 524288.0;
  f2* p; float* out; CK(hipMalloc(&p, (size_t)bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(p, 0, (size_t)bytes));
  for (int g : {1024, 4096, 16384}) {
    printf("grid %d\n", g);
    auto R = [&](const char* n, auto fn) { float ms = timeit(fn); printf("  %-34s %7.3f ms %7.1f GB/s\n", n, ms, bytes / ms / 1e6); };
    R("read  contiguous 32 KiB, nt", [&] { k<false, false, true><<<g, 256>>>(p, ntile, out); });
    R("read  16 x 2 KiB / 32 KiB, nt", [&] { k<true, false, true><<<g, 256>>>(p, ntile, out); });
    R("read  16 x 2 KiB / 32 KiB, plain", [&] { k<true, false, false><<<g, 256>>>(p, ntile, out); });
    R("write contiguous 32 KiB, nt", [&] { k<false, true, true><<<g, 256>>>(p, ntile, out); });
    R("write 16 x 2 KiB / 32 KiB, nt", [&] { k<true, true, true><<<g, 256>>>(p, ntile, out); });
  }
  return 0;
}

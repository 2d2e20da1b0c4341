// Streaming bandwidth ceilings of the chip for the access shapes the FFT
// kernels use: write-only, read-only and copy, 8-byte (complex64) and 16-byte
// accesses, plain and non-temporal.  hipcc --offload-arch=gfx950 -O3 bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <class T, bool NT>
__global__ __launch_bounds__(256) void fill(T* p, long n) {
  T v;
  for (int i = 0; i < sizeof(T) / 4; ++i) ((float*)&v)[i] = 1.0f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
  }
}
template <class T, bool NT>
__global__ __launch_bounds__(256) void sum(const T* p, long n, float* out) {
  float a = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    T v = NT ? __builtin_nontemporal_load(p + i) : p[i];
    a += ((float*)&v)[0];
  }
  if (a == 123.456f) *out = a;
}
template <class T, bool NT>
__global__ __launch_bounds__(256) void copy(const T* p, T* q, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    T v = NT ? __builtin_nontemporal_load(p + i) : p[i];
    if (NT) __builtin_nontemporal_store(v, q + i); else q[i] = v;
  }
}
// 32 KiB contiguous blocks per workgroup (16 rows of 2 KiB), like the FFT passes
template <bool NT>
__global__ __launch_bounds__(256) void fill_rows(f2* p, long nblk) {
  f2 v = {1.f, 2.f};
  for (long b = blockIdx.x; b < nblk; b += gridDim.x) {
    f2* q = p + b * 4096 + threadIdx.x;
#pragma unroll
    for (int r = 0; r < 16; ++r) { if (NT) __builtin_nontemporal_store(v, q + r * 256); else q[r * 256] = v; }
  }
}
template <class F> float timeit(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 5;
}
int main() {
  const long bytes = 4L << 30;
  char *p, *q; float* out;
  CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(p, 0, bytes)); CK(hipMemset(q, 0, bytes));
  for (int g : {2048, 8192, 32768}) {
    printf("grid %d\n", g);
    auto R = [&](const char* name, auto fn, double nb) {
      float ms = timeit(fn);
      printf("  %-28s %7.3f ms  %7.1f GB/s\n", name, ms, nb / ms / 1e6);
    };
    const dim3 G(g), B(256);
    const long n8 = bytes / 8, n16 = bytes / 16;
    R("fill  8B plain", [&] { fill<f2, false><<<G, B>>>((f2*)p, n8); }, bytes);
    R("fill  8B nt", [&] { fill<f2, true><<<G, B>>>((f2*)p, n8); }, bytes);
    R("fill 16B plain", [&] { fill<f4, false><<<G, B>>>((f4*)p, n16); }, bytes);
    R("fill 16B nt", [&] { fill<f4, true><<<G, B>>>((f4*)p, n16); }, bytes);
    R("fill rows 8B plain", [&] { fill_rows<false><<<G, B>>>((f2*)p, bytes / 32768); }, bytes);
    R("fill rows 8B nt", [&] { fill_rows<true><<<G, B>>>((f2*)p, bytes / 32768); }, bytes);
    R("read  8B plain", [&] { sum<f2, false><<<G, B>>>((const f2*)p, n8, out); }, bytes);
    R("read  8B nt", [&] { sum<f2, true><<<G, B>>>((const f2*)p, n8, out); }, bytes);
    R("read 16B plain", [&] { sum<f4, false><<<G, B>>>((const f4*)p, n16, out); }, bytes);
    R("copy  8B plain", [&] { copy<f2, false><<<G, B>>>((const f2*)p, (f2*)q, n8); }, 2.0 * bytes);
    R("copy  8B nt", [&] { copy<f2, true><<<G, B>>>((const f2*)p, (f2*)q, n8); }, 2.0 * bytes);
    R("copy 16B nt", [&] { copy<f4, true><<<G, B>>>((const f4*)p, (f4*)q, n16); }, 2.0 * bytes);
  }
  return 0;
}

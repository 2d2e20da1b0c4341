// Does a T-sized intermediate survive in the 256 MB Infinity Cache between the
// kernel that writes it and the kernel that reads it?  Producer / consumer
// pairs over working sets from 32 MB to 4 GB, plain and non-temporal accesses.
//   hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void fill(f4* p, long n, float s) {
  f4 v = {s, s, s, s};
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
  }
}
template <bool NT>
__global__ __launch_bounds__(256) void sum(const f4* p, long n, float* out) {
  float a = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    f4 v = NT ? __builtin_nontemporal_load(p + i) : p[i];
    a += v.x;
  }
  if (a == 123.456f) *out = a;
}
template <bool NT>
__global__ __launch_bounds__(256) void copy(const f4* p, f4* q, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    f4 v = NT ? __builtin_nontemporal_load(p + i) : p[i];
    if (NT) __builtin_nontemporal_store(v, q + i); else q[i] = v;
  }
}

int main() {
  const long cap = 4L << 30;
  char *a, *b; float* out;
  CK(hipMalloc(&a, cap)); CK(hipMalloc(&b, cap)); CK(hipMalloc(&out, 4));
  CK(hipMemset(a, 0, cap)); CK(hipMemset(b, 0, cap));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const dim3 G(4096), B(256);
  printf("%8s %4s | %22s | %22s | %30s\n", "MB", "nt", "write then read (GB/s)", "read only, repeated", "copy a->b then b->a (GB/s)");
  for (long mb : {32L, 64L, 96L, 128L, 192L, 256L, 384L, 512L, 1024L, 4096L}) {
    const long bytes = mb << 20, n = bytes / 16;
    const int reps = (int)(mb <= 256 ? 64 : (mb <= 1024 ? 16 : 4));
    for (int nt = 0; nt < 2; ++nt) {
      auto W = [&](float s) { if (nt) fill<true><<<G, B>>>((f4*)a, n, s); else fill<false><<<G, B>>>((f4*)a, n, s); };
      auto R = [&]() { if (nt) sum<true><<<G, B>>>((const f4*)a, n, out); else sum<false><<<G, B>>>((const f4*)a, n, out); };
      auto C = [&](char* s, char* d) { if (nt) copy<true><<<G, B>>>((const f4*)s, (f4*)d, n); else copy<false><<<G, B>>>((const f4*)s, (f4*)d, n); };
      float ms;
      W(1.f); R(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) { W((float)i); R(); }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      const double wr = 2.0 * bytes * reps / ms / 1e6;
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) R();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      const double rr = 1.0 * bytes * reps / ms / 1e6;
      C(a, b); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) { C(a, b); C(b, a); }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      const double cc = 4.0 * bytes * reps / ms / 1e6;
      printf("%8ld %4d | %22.0f | %22.0f | %30.0f\n", mb, nt, wr, rr, cc);
    }
  }
  return 0;
}

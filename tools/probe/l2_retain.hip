// Does the XCD L2 retain freshly stored lines for later loads?
// Each workgroup loops: store CH bytes of its own scratch, drain, load them back.
// Total scratch = grid * CH is chosen well inside the 8 x 4 MiB of L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int SC1, int NEIGH>
__global__ __launch_bounds__(256) void k(float4* scratch, int chunk16, int iters, float* sink) {
  // chunk16 = float4 elements per workgroup
  float4* mine = scratch + (long)blockIdx.x * chunk16;
  // neighbour on the same XCD under round-robin dispatch (b + 8)
  const int nb = (blockIdx.x + 8 * NEIGH) % gridDim.x;
  const float4* theirs = scratch + (long)nb * chunk16;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < chunk16; i += 256)
      mine[i] = make_float4(it, i, blockIdx.x, 1.f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < chunk16; i += 256) {
      float4 v;
      if (SC1) {
        v.x = __hip_atomic_load(&theirs[i].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.y = 0;
      } else {
        v = theirs[i];
      }
      acc += v.x + v.y;
    }
    __syncthreads();
  }
  if (acc == -1.f) *sink = acc;
}

template <int SC1, int NEIGH>
void run(const char* name, float4* s, int grid, int chunk_bytes, int iters, float* sink) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<SC1, NEIGH>), dim3(grid), dim3(256), 0, 0, s, chunk_bytes / 16, iters, sink);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<SC1, NEIGH>), dim3(grid), dim3(256), 0, 0, s, chunk_bytes / 16, iters, sink);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double bytes = 2.0 * grid * (double)chunk_bytes * iters;
  printf("%-28s grid %4d chunk %7d B total %6.1f MiB  %7.3f ms  %8.1f GB/s (st+ld)\n", name, grid,
         chunk_bytes, grid * (double)chunk_bytes / 1048576, ms, bytes / ms / 1e6);
}

int main() {
  float4* s; float* sink;
  CK(hipMalloc(&s, 1L << 31)); CK(hipMalloc(&sink, 4));
  for (int chunk : {16384, 65536, 1048576}) {
    int iters = chunk >= 1048576 ? 16 : 256;
    run<0, 0>("own, plain loads", s, 512, chunk, iters, sink);
    run<1, 0>("own, sc1 loads (4B)", s, 512, chunk, iters, sink);
    run<0, 1>("same-XCD neighbour, plain", s, 512, chunk, iters, sink);
  }
  return 0;
}

// the XCD-cooperative two-pass 256 x 256 transform WITH
// arithmetic, measured against the product's one-workgroup-per-tile kernel.
//   solo  one workgroup per tile: 16 pass-1 row groups -> intermediate in the
//         output tile -> barrier -> 16 pass-2 column groups (fft2_v2_kernel)
//   coop  16 workgroups of ONE XCD share a tile: member m runs pass 1 of row
//         group m into a per-team slot that is reused (two slots, alternating:
//         its lines stay in that XCD's 4 MiB L2), team barrier (a counter in
//         memory), member m runs pass 2 of column group m from the slot
//         (loads that bypass the CU's L1) and stores the final rows.
// Same butterflies, same order: the outputs must be bit-identical.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../tike_amd/csrc -I../../include \
//         xcd_fft.hip -o xcd_fft
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fft_engine2.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int N = 256;
constexpr int GS = 16;  // team size = row groups = column groups

struct Ctl {
  unsigned registered;
  unsigned xcd_count[8];
  unsigned timeout;
  unsigned pad[6];
  unsigned team_ctr[8 * 64 * 16];  // [xcd][team], one 64-byte line each
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ unsigned ld_relaxed(unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a load served by L2 (not by this CU's L1, which may hold a stale line)
__device__ __forceinline__ cf ld_l2(const cf* p) {
  const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  return mk(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}

template <int WPS>
__global__ __launch_bounds__(N, WPS) void solo_kernel(const cf* __restrict__ in,
                                                     cf* __restrict__ out, long ntile,
                                                     float scale, const cf* __restrict__ twtab) {
  using G2 = Fft2Geom<N>;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const cf* __restrict__ src = in + tile * (long)N * N;
    cf* __restrict__ dst = out + tile * (long)N * N;
    int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
    asm volatile("" : "+v"(line), "+v"(j));
    const FftTwLds<N> tw{twl, j};
    for (int r = 0; r < G2::RB; ++r)
      fft2_pass1<N, false>(lds, twtab, tw, line, j, r,
                           [&](int y, int e, auto) { return tk_ld_stream(src + y * N + e); }, dst);
    __syncthreads();
    for (int k1 = 0; k1 < 16; ++k1)
      fft2_pass2<N, false>(dst, k1,
                           [&](int ky, int t, cf v) { tk_st_stream(dst + ky * N + t, v * scale); });
    __syncthreads();
  }
}

template <int WPS>
__global__ __launch_bounds__(N, WPS) void coop_kernel(const cf* __restrict__ in,
                                                     cf* __restrict__ out, long ntile,
                                                     float scale, const cf* __restrict__ twtab,
                                                     Ctl* ctl, cf* slots) {
  using G2 = Fft2Geom<N>;
  __shared__ cf lds[G2::LDS_ELEMS + FftTwLds<N>::ELEMS];
  __shared__ unsigned sh[4];
  cf* twl = lds + G2::LDS_ELEMS;
  FftTwLds<N>::fill(twl, twtab);
  const int t = threadIdx.x;
  // census: which XCD am I on, which team of 16 there, which member
  if (t == 0) {
    const unsigned x = xcc_id() & 7;
    const unsigned slot = atomicAdd(&ctl->xcd_count[x], 1u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    atomicAdd(&ctl->registered, 1u);
    unsigned spins = 0;
    while (ld_relaxed(&ctl->registered) < gridDim.x) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 22)) { atomicOr(&ctl->timeout, 1u); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    unsigned base = 0, total = 0;
    for (unsigned k = 0; k < 8; ++k) {
      const unsigned g = ld_relaxed(&ctl->xcd_count[k]) / GS;
      if (k < x) base += g;
      total += g;
    }
    const unsigned mine = ld_relaxed(&ctl->xcd_count[x]) / GS;
    sh[0] = x;
    sh[1] = slot < mine * GS ? base + slot / GS : 0xffffffffu;
    sh[2] = slot % GS;
    sh[3] = total;
  }
  __syncthreads();
  if (sh[1] == 0xffffffffu || ld_relaxed(&ctl->timeout)) return;  // leftover workgroup
  const long team = sh[1], nteam = sh[3];
  const int m = (int)sh[2];
  unsigned* ctr = &ctl->team_ctr[(sh[0] * 64 + (sh[1] % 64)) * 16];
  int line = threadIdx.x / G2::T, j = threadIdx.x % G2::T;
  asm volatile("" : "+v"(line), "+v"(j));
  const FftTwLds<N> tw{twl, j};
  unsigned phase = 0;
  for (long tile = team; tile < ntile; tile += nteam) {
    const cf* __restrict__ src = in + tile * (long)N * N;
    cf* __restrict__ dst = out + tile * (long)N * N;
    // two slots per team, alternating: a member can be one tile ahead of the
    // others (it reads slot A in pass 2 of tile i while nobody can write A
    // before having passed the barrier of tile i + 1)
    cf* mid = slots + (team * 2 + (phase & 1)) * (long)N * N;
    fft2_pass1<N, false>(lds, twtab, tw, line, j, m,
                         [&](int y, int e, auto) { return tk_ld_stream(src + y * N + e); }, mid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows are in L2
    __syncthreads();
    ++phase;
    if (t == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (ld_relaxed(ctr) < phase * GS) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 20)) { atomicOr(&ctl->timeout, 2u); break; }
      }
    }
    __syncthreads();
    {
      cf u[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) u[r] = ld_l2(mid + (16 * r + m) * N + t);
      Dft<16, false>::run(u);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) tk_st_stream(dst + (m + 16 * k2) * N + t, u[k2] * scale);
    }
  }
}

int main(int argc, char** argv) {
  const long ntile = argc > 1 ? atol(argv[1]) : 4096;
  cf *in, *out0, *out1, *slots, *tw;
  Ctl* ctl;
  const size_t bytes = (size_t)ntile * N * N * sizeof(cf);
  CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out0, bytes)); CK(hipMalloc(&out1, bytes));
  CK(hipMalloc(&slots, 2L * 64 * 8 * N * N * sizeof(cf)));
  CK(hipMalloc(&ctl, sizeof(Ctl)));
  std::vector<cf> host(2048);
  for (int n = 32; n <= 1024; n *= 2)
    for (int k = 0; k < n; ++k) {
      const double a = -2.0 * M_PI * (double)k / (double)n;
      host[n + k] = mk((float)std::cos(a), (float)std::sin(a));
    }
  CK(hipMalloc(&tw, host.size() * sizeof(cf)));
  CK(hipMemcpy(tw, host.data(), host.size() * sizeof(cf), hipMemcpyHostToDevice));
  {
    std::vector<cf> h((size_t)ntile * N * N);
    unsigned s = 12345;
    for (auto& v : h) {
      s = s * 1664525u + 1013904223u; v.x = (float)(s >> 8) / (1 << 24) - 0.5f;
      s = s * 1664525u + 1013904223u; v.y = (float)(s >> 8) / (1 << 24) - 0.5f;
    }
    CK(hipMemcpy(in, h.data(), bytes, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const float scale = 1.0f / N;
  auto time_solo = [&](int grid, auto kern, const char* name) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(N), 0, 0, in, out0, ntile, scale, tw);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-28s grid %5d  %8.3f ms  %6.3f Mtile/s  %5.1f %% of 8 TB/s (2 T per tile)\n", name,
           grid, best, ntile / best / 1e3, 100.0 * ntile * 2.0 * N * N * 8 / (best * 1e-3) / 8e12);
  };
  time_solo(1024, solo_kernel<4>, "solo, 4 workgroups / CU");
  time_solo(512, solo_kernel<2>, "solo, 2 workgroups / CU");
  auto time_coop = [&](int grid, auto kern, const char* name) {
    float best = 1e30f; Ctl h; memset(&h, 0, sizeof(h));
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(ctl, 0, sizeof(Ctl)));
      CK(hipMemset(out1, 0, bytes));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(N), 0, 0, in, out1, ntile, scale, tw, ctl, slots);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
      if (h.timeout) break;
    }
    std::vector<cf> a((size_t)N * N * 8), b((size_t)N * N * 8);
    long bad = 0;
    for (long tile : {0L, ntile / 2, ntile - 8}) {
      CK(hipMemcpy(a.data(), out0 + tile * N * N, a.size() * sizeof(cf), hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), out1 + tile * N * N, b.size() * sizeof(cf), hipMemcpyDeviceToHost));
      bad += memcmp(a.data(), b.data(), a.size() * sizeof(cf)) != 0;
    }
    printf("%-28s grid %5d  %8.3f ms  %6.3f Mtile/s  %5.1f %%  timeout=%u mismatching blocks=%ld  per-XCD:",
           name, grid, best, ntile / best / 1e3,
           100.0 * ntile * 2.0 * N * N * 8 / (best * 1e-3) / 8e12, h.timeout, bad);
    for (int k = 0; k < 8; ++k) printf(" %u", h.xcd_count[k]);
    printf("\n");
  };
  time_coop(256, coop_kernel<1>, "coop, 1 workgroup / CU");
  time_coop(512, coop_kernel<2>, "coop, 2 workgroups / CU");
  time_coop(1024, coop_kernel<4>, "coop, 4 workgroups / CU");
  return 0;
}

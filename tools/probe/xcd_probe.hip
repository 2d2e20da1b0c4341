// Probe: can a group of workgroups that share one XCD hand a 512 KiB tile
// intermediate to each other through that XCD's L2 (no write-back to HBM)?
// Mimics the memory pattern of the two-pass 256x256 FFT (no arithmetic):
//   pass 1 (member m):  read input rows {m + 16*y2}, write mid rows 16m..16m+15
//   group barrier
//   pass 2 (member m):  read mid rows {16r + m}, write rows {m + 16*k2} in place
// Modes: 0 = solo (one workgroup does both passes of a tile, today's kernels)
//        1 = group of 16 same-XCD workgroups per tile (XCC_ID census)
// Build: hipcc --offload-arch=gfx950 -O3 -o xcd_probe xcd_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float2 cf;
constexpr int N = 256;
constexpr int GS = 16;  // group size

struct Ctl {
  unsigned registered;      // census
  unsigned xcd_count[8];
  unsigned timeout;
  unsigned pad[6];
  unsigned group_ctr[8 * 64 * 16];  // [xcd][group] padded to 64 B
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ unsigned ld_relaxed(unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// load that bypasses this CU's L1 (sc1), served by L2
__device__ __forceinline__ cf ld_sc1(const cf* p) {
  unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
  cf r;
  r.x = __uint_as_float((unsigned)v);
  r.y = __uint_as_float((unsigned)(v >> 32));
  return r;
}

template <int NT>
__device__ __forceinline__ cf ldin(const cf* p) {
  if (NT) {
    cf r;
    r.x = __builtin_nontemporal_load(&p->x);
    r.y = __builtin_nontemporal_load(&p->y);
    return r;
  }
  return *p;
}

__device__ __forceinline__ cf mkv(long tile, int row, int col) {
  return make_float2((float)(tile * 7 + row), (float)(col + 3 * row));
}

template <int MODE, int WGSCOPE, int WPS, int NT>
__global__ __launch_bounds__(256, WPS) void probe(const cf* __restrict__ in, cf* __restrict__ out,
                                                long ntile, Ctl* ctl, unsigned* errors, cf* scratch) {
  const int t = threadIdx.x;
  __shared__ unsigned sh[4];
  long first, stride;
  int m = 0;
  unsigned* gctr = nullptr;
  if (MODE == 0) {
    first = blockIdx.x;
    stride = gridDim.x;
  } else {
    // census: register on this XCD, wait for everybody
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
      sh[1] = slot < mine * GS ? base + slot / GS : 0xffffffffu;  // global group index
      sh[2] = slot % GS;
      sh[3] = total;
    }
    __syncthreads();
    if (sh[1] == 0xffffffffu || ld_relaxed(&ctl->timeout)) return;  // leftover workgroup
    first = sh[1];
    stride = sh[3];
    m = sh[2];
    gctr = &ctl->group_ctr[(sh[0] * 64 + (sh[1] % 64)) * 16];
    // note: (sh[1] % 64) is unique within an XCD as long as an XCD has <= 64 groups
  }
  unsigned phase = 0;
  unsigned bad = 0;
  cf pv[16];
  for (long tile = first; tile < ntile; tile += stride) {
    const cf* src = in + tile * (long)N * N;
    cf* dst = out + tile * (long)N * N;
    cf* fin = dst;
    if (MODE == 2 && scratch) dst = scratch + first * (long)N * N;  // reused per-group slot
    // ---- pass 1
    if (MODE == 2) {
      // software-pipelined: this tile's input rows were loaded one iteration
      // ago; store them, then issue the next tile's loads before the barrier
      if (tile == first) {
#pragma unroll
        for (int y2 = 0; y2 < 16; ++y2) pv[y2] = ldin<NT>(src + (m + 16 * y2) * N + t);
      }
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) {
        cf o = pv[k1];
        o.x += 1.0f;
        dst[(16 * m + k1) * N + t] = o;
      }
      if (tile + stride < ntile) {
        const cf* nsrc = in + (tile + stride) * (long)N * N;
#pragma unroll
        for (int y2 = 0; y2 < 16; ++y2) pv[y2] = ldin<NT>(nsrc + (m + 16 * y2) * N + t);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // the 16 stores are older
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else
    for (int r = (MODE ? m : 0); r < (MODE ? m + 1 : 16); ++r) {
      cf v[16];
#pragma unroll
      for (int y2 = 0; y2 < 16; ++y2) v[y2] = ldin<NT>(src + (r + 16 * y2) * N + t);
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) {
        cf o = v[k1];
        o.x += 1.0f;
        dst[(16 * r + k1) * N + t] = o;
      }
    }
    if (MODE == 0) {
      __syncthreads();
    } else {
      if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      ++phase;
      if (t == 0) {
        if (WGSCOPE == 1)
          __hip_atomic_fetch_add(gctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
          __hip_atomic_fetch_add(gctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (ld_relaxed(gctr) < phase * GS) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 20)) { atomicOr(&ctl->timeout, 2u); break; }
        }
      }
      __syncthreads();
    }
    // ---- pass 2 (in place)
    for (int k1 = (MODE ? m : 0); k1 < (MODE ? m + 1 : 16); ++k1) {
      cf u[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const cf* p = dst + (16 * r + k1) * N + t;
        u[r] = (MODE && WGSCOPE < 2) ? ld_sc1(p) : *p;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // row 16r+k1 of mid came from input row r + 16*k1
        const cf e = mkv(tile, r + 16 * k1, t);
        if (u[r].x != e.x + 1.0f || u[r].y != e.y) ++bad;
      }
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        cf o = u[k2];
        o.y += 2.0f;
        if (NT) {
          __builtin_nontemporal_store(o.x, &fin[(k1 + 16 * k2) * N + t].x);
          __builtin_nontemporal_store(o.y, &fin[(k1 + 16 * k2) * N + t].y);
        } else {
          fin[(k1 + 16 * k2) * N + t] = o;
        }
      }
    }
    if (MODE == 0) __syncthreads();
  }
  if (bad) atomicAdd(errors, bad);
}

__global__ void fill(cf* in, long ntile) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < ntile * N * N;
       i += (long)gridDim.x * blockDim.x) {
    const long tile = i / (N * N);
    const int row = (int)((i / N) % N), col = (int)(i % N);
    in[i] = mkv(tile, row, col);
  }
}

template <int MODE, int WGSCOPE, int WPS, int NT = 0>
static void run(const char* name, const cf* in, cf* out, long ntile, int grid, Ctl* ctl,
                unsigned* errors, cf* scratch = nullptr) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  Ctl h;
  unsigned herr = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(ctl, 0, sizeof(Ctl)));
    CK(hipMemset(errors, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<MODE, WGSCOPE, WPS, NT>), dim3(grid), dim3(256), 0, 0, in, out, ntile, ctl,
                       errors, scratch);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    CK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost));
    if (h.timeout || herr) break;
  }
  printf("%-34s grid %4d  %8.3f ms  %6.3f Mtile/s  timeout=%u errors=%u  xcd counts:", name, grid,
         best, ntile / best / 1e3, h.timeout, herr);
  for (int k = 0; k < 8; ++k) printf(" %u", h.xcd_count[k]);
  printf("\n");
}

int main(int argc, char** argv) {
  const long ntile = argc > 1 ? atol(argv[1]) : 4096;
  cf *in, *out;
  Ctl* ctl;
  unsigned* errors;
  CK(hipMalloc(&in, ntile * N * N * sizeof(cf)));
  CK(hipMalloc(&out, ntile * N * N * sizeof(cf)));
  CK(hipMalloc(&ctl, sizeof(Ctl)));
  CK(hipMalloc(&errors, 4));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, in, ntile);
  CK(hipDeviceSynchronize());
  cf* scratch;
  CK(hipMalloc(&scratch, 256L * N * N * sizeof(cf)));
  run<0, 0, 2>("solo (one WG per tile)", in, out, ntile, 512, ctl, errors);
  run<2, 2, 2, 0>("group16 pipe, acq, slot", in, out, ntile, 512, ctl, errors, scratch);
  run<2, 2, 2, 1>("group16 pipe, acq, slot, nt", in, out, ntile, 512, ctl, errors, scratch);
  run<2, 0, 2, 1>("group16 pipe, sc1, slot, nt", in, out, ntile, 512, ctl, errors, scratch);
  run<2, 2, 2, 1>("group16 pipe, acq, slot, nt", in, out, ntile, 256, ctl, errors, scratch);
  run<2, 2, 4, 1>("group16 pipe, acq, slot, nt", in, out, ntile, 1024, ctl, errors, scratch);
  return 0;
}

// Round-4 probe: do the T-sized hand-offs of the c3 chain
// (forward pass 1 -> gradient pass -> pass 2 + gradients) get cheaper when producer
// and consumer kernels are RESIDENT AT THE SAME TIME on disjoint CU sets and the
// hand-off is a short ring that lives in the 256 MiB Infinity Cache?
//
// Synthetic kernels with the access shapes, launch shapes and (roughly) the VALU
// load of the three product kernels; every hand-off goes through the guide's
// inter-workgroup protocol (MI355X_MICROARCH.md, "Workgroup dispatch ... visibility").
//   A  (tike_fwd_pass1):  item (pos, row group): stores 16 rows x 2 KiB of every mode
//   B  (tike_fwd_grad_ifft2_pass1, resident kernel): item (pos, k1), 512 threads,
//      half h holds rows {16r + k1} of modes [4h, 4h+4) in 128 registers; stores
//      16 rows of every mode into the second hand-off
//   C  (tike_ifft2_pass2_gradients): workgroup = (row residue, 64-column group),
//      walks positions; wave = 2 modes, lane = column; 16 row segments of 512 B
// Modes:
//   seq    the three launches back to back on one stream over full-size buffers
//   ring   three streams with disjoint CU masks, rings of R slots, flags
// Every spin is bounded (abort flag), so a broken assumption ends the run
// instead of hanging the GPU.
//   hipcc --offload-arch=gfx950 -O3 ring_probe.hip -o ring_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <set>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float2 cf;
constexpr int N = 256;            // tile edge
// Row pitch (elements) and mode-tile pitch of both hand-offs: 256 / 65536 in the
// product; padded values take the rows {16 r + k1} off the 32 KiB stride.
__constant__ int c_pitch = 256;
__constant__ long c_tile = 65536;
#define PITCH c_pitch
#define TILE c_tile
constexpr int LINE = 16;          // unsigneds per counter (64 B)

struct Ctl {
  unsigned abort_flag;
  unsigned errors;
  unsigned timeouts;
  unsigned pad[13];
};

enum StoreKind { ST_PLAIN = 0, ST_NT = 1, ST_SC1 = 2 };
enum LoadKind { LD_ACQ = 0, LD_SC1 = 1 };

template <int SK>
__device__ __forceinline__ void st(cf* p, cf v) {
  if (SK == ST_NT) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
  } else if (SK == ST_SC1) {
    unsigned long long u = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
    __hip_atomic_store((unsigned long long*)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *p = v;
  }
}
template <int LK>
__device__ __forceinline__ cf ld(const cf* p) {
  if (LK == LD_SC1) {
    unsigned long long u = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    cf r; r.x = __uint_as_float((unsigned)u); r.y = __uint_as_float((unsigned)(u >> 32));
    return r;
  }
  return *p;
}

__device__ __forceinline__ unsigned poll(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// lane 0 of wave 0 waits until *ctr >= want (bounded), then (LK == LD_ACQ) acquires
template <int LK>
__device__ __forceinline__ void wait_for(const unsigned* ctr, unsigned want, Ctl* ctl) {
  if (threadIdx.x == 0) {
    unsigned n = 0;
    while (poll(ctr) < want) {
      __builtin_amdgcn_s_sleep(8);
      if ((++n & 255) == 0) {
        if (poll(&ctl->abort_flag)) break;
        if (n > (1u << 22)) { atomicAdd(&ctl->timeouts, 1u); atomicExch(&ctl->abort_flag, 1u); break; }
      }
    }
    if (LK == LD_ACQ) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
}

// every storing wave has drained; one lane publishes
template <int SK>
__device__ __forceinline__ void publish(unsigned* ctr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (SK != ST_SC1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ cf mkv(int pos, int s, int row, int col) {
  return make_float2((float)(pos * 3 + s) + 0.25f, (float)(row * 256 + col));
}

__device__ __forceinline__ cf work(cf v, int K, float a) {
  // K packed FMAs that leave v unchanged in exact arithmetic (a = 1, b = 0 at run time)
  cf b = make_float2(a - 1.0f, a - 1.0f);
#pragma unroll 4
  for (int k = 0; k < K; ++k) { v.x = fmaf(v.x, a, b.x); v.y = fmaf(v.y, a, b.y); }
  return v;
}

// ---- A: producer of hand-off 1 -------------------------------------------------
template <int SYNC, int SK>
__global__ __launch_bounds__(256) void kernA(cf* __restrict__ h1, int R1, int S, int K, float a,
                                            unsigned* ready1, const unsigned* done1, Ctl* ctl) {
  const int pos = blockIdx.x >> 4, rg = blockIdx.x & 15, t = threadIdx.x;
  if (SYNC) {
    if (pos >= R1) wait_for<LD_SC1>(done1 + (long)(pos - R1) * LINE, 16u, ctl);
    if (poll(&ctl->abort_flag)) return;
  }
  cf* slot = h1 + (long)(pos % R1) * S * TILE;
  for (int s = 0; s < S; ++s) {
    cf* tile = slot + (long)s * TILE + (long)rg * 16 * PITCH;
#pragma unroll
    for (int r = 0; r < 16; ++r) st<SK>(tile + r * PITCH + t, work(mkv(pos, s, rg * 16 + r, t), K, a));
  }
  if (SYNC) publish<SK>(ready1 + (long)pos * LINE);
}

// ---- B: consumes hand-off 1, produces hand-off 2 (512 threads, 128 data registers) --
template <int SYNC, int SK, int LK, int CHECK>
__global__ __launch_bounds__(512) void kernB(const cf* __restrict__ h1, cf* __restrict__ h2, int R1, int R2,
                                            int K, float a, const unsigned* ready1, unsigned* done1,
                                            unsigned* ready2, const unsigned* done2, Ctl* ctl,
                                            const float* __restrict__ data, float* __restrict__ sink) {
  constexpr int S = 8, MH = 4;
  const int pos = blockIdx.x >> 4, k1 = blockIdx.x & 15;
  const int t = threadIdx.x & 255, h = threadIdx.x >> 8;
  if (SYNC) {
    wait_for<LK>(ready1 + (long)pos * LINE, 16u, ctl);
    if (poll(&ctl->abort_flag)) return;
  }
  const cf* slot = h1 + (long)(pos % R1) * S * TILE;
  cf v[MH][16];
#pragma unroll
  for (int m = 0; m < MH; ++m) {
    const cf* tile = slot + (long)(h * MH + m) * TILE;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[m][r] = ld<LK>(tile + (long)(16 * r + k1) * PITCH + t);
  }
  // the measured counts of this (position, k1) row set: D-sized read, as the product's kernel
  float d = 0.f;
  if (h == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) d += data[((long)pos * N + (16 * r + k1)) * N + t];
  }
  unsigned bad = 0;
#pragma unroll
  for (int m = 0; m < MH; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (CHECK) {
        const cf e = mkv(pos, h * MH + m, 16 * r + k1, t);
        bad += (v[m][r].x != e.x) | (v[m][r].y != e.y);
      }
      v[m][r] = work(v[m][r], K, a);
    }
  if (CHECK && bad) atomicAdd(&ctl->errors, bad);
  if (SYNC) {
    // all loads of this item have landed: hand the slot back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(done1 + (long)pos * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (pos >= R2) wait_for<LD_SC1>(done2 + (long)(pos - R2) * LINE, 64u, ctl);
    if (poll(&ctl->abort_flag)) return;
  }
  cf* oslot = h2 + (long)(pos % R2) * S * TILE;
#pragma unroll
  for (int m = 0; m < MH; ++m) {
    cf* tile = oslot + (long)(h * MH + m) * TILE + (long)k1 * 16 * PITCH;
#pragma unroll
    for (int r = 0; r < 16; ++r) st<SK>(tile + r * PITCH + t, v[m][r]);
  }
  if (d == 123.456f) sink[0] = d;
  if (SYNC) publish<SK>(ready2 + (long)pos * LINE);
}

// ---- C: consumes hand-off 2 -----------------------------------------------------
// grid = 64 slices x J walkers; workgroup (slice, j) walks positions j, j + J, ...
template <int SYNC, int LK, int CHECK>
__global__ __launch_bounds__(256) void kernC(const cf* __restrict__ h2, int R2, int npos, int J, int K, float a,
                                            const unsigned* ready2, unsigned* done2, Ctl* ctl,
                                            float* __restrict__ sink) {
  constexpr int S = 8;
  const int slice = blockIdx.x & 63, j = blockIdx.x >> 6;
  const int ya = slice & 15, cg = slice >> 4;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  cf acc = make_float2(0.f, 0.f);
  unsigned bad = 0;
  for (int pos = j; pos < npos; pos += J) {
    if (SYNC) {
      wait_for<LK>(ready2 + (long)pos * LINE, 16u, ctl);
      if (poll(&ctl->abort_flag)) return;
    }
    const cf* slot = h2 + (long)(pos % R2) * S * TILE;
    cf v[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const cf* tile = slot + (long)(2 * w + m) * TILE + cg * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) v[m][r] = ld<LK>(tile + (long)(ya + 16 * r) * PITCH);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (CHECK) {
          // B stored row (16 r' + k1) of A's tile as row (16 k1 + r'): row q of h2 is A's row 16 (q & 15) + (q >> 4)
          const int q = ya + 16 * r;
          const cf e = mkv(pos, 2 * w + m, 16 * (q & 15) + (q >> 4), cg * 64 + lane);
          bad += (v[m][r].x != e.x) | (v[m][r].y != e.y);
        }
        const cf x = work(v[m][r], K, a);
        acc.x += x.x; acc.y += x.y;
      }
    if (SYNC) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(done2 + (long)pos * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (CHECK && bad) atomicAdd(&ctl->errors, bad);
  if (acc.x == 123.456f) sink[1] = acc.y;
}

// ---- census: which CUs does a masked stream run on? ------------------------------
__global__ __launch_bounds__(256) void census(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xff00);  // cu_id[11:8] sh_id[12] se_id[15:13]
  }
  // stay long enough for the whole grid to spread
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(20);
}

static void mask_range(std::vector<uint32_t>& m, int lo, int hi) {
  for (int b = lo; b < hi; ++b) m[b >> 5] |= 1u << (b & 31);
}

int main(int argc, char** argv) {
  int npos = 1000, R1 = 16, R2 = 16, nA = 64, nB = 112, nC = 80, KA = 28, KB = 38, KC = 20, J = 3;
  int sk = ST_PLAIN, lk = LD_ACQ, check = 1, reps = 3;
  std::string mode = "all";
  int pitch = 256; long tilepad = 0;
  for (int i = 1; i < argc; ++i) {
    auto eq = [&](const char* k) { return !strncmp(argv[i], k, strlen(k)) ? argv[i] + strlen(k) : nullptr; };
    const char* v;
    if ((v = eq("--npos="))) npos = atoi(v);
    else if ((v = eq("--r1="))) R1 = atoi(v);
    else if ((v = eq("--r2="))) R2 = atoi(v);
    else if ((v = eq("--cus="))) sscanf(v, "%d,%d,%d", &nA, &nB, &nC);
    else if ((v = eq("--k="))) sscanf(v, "%d,%d,%d", &KA, &KB, &KC);
    else if ((v = eq("--j="))) J = atoi(v);
    else if ((v = eq("--store="))) sk = !strcmp(v, "nt") ? ST_NT : (!strcmp(v, "sc1") ? ST_SC1 : ST_PLAIN);
    else if ((v = eq("--load="))) lk = !strcmp(v, "sc1") ? LD_SC1 : LD_ACQ;
    else if ((v = eq("--check="))) check = atoi(v);
    else if ((v = eq("--reps="))) reps = atoi(v);
    else if ((v = eq("--mode="))) mode = v;
    else if ((v = eq("--pitch="))) pitch = atoi(v);
    else if ((v = eq("--tilepad="))) tilepad = atol(v);
    else { printf("unknown argument %s\n", argv[i]); return 2; }
  }
  constexpr int S = 8;
  const long tile_elems = (long)pitch * N + tilepad;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(c_pitch), &pitch, sizeof(int)));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(c_tile), &tile_elems, sizeof(long)));
  printf("row pitch %d elements, tile pitch %ld elements\n", pitch, tile_elems);
  const long slot = (long)S * tile_elems * sizeof(cf);
  printf("npos %d  slot %.1f MiB  rings %d / %d slots (%.0f + %.0f MiB)  CUs %d/%d/%d  K %d/%d/%d  J %d  store %d load %d check %d\n",
         npos, slot / 1048576.0, R1, R2, R1 * slot / 1048576.0, R2 * slot / 1048576.0, nA, nB, nC, KA, KB, KC, J, sk, lk, check);

  cf *h1, *h2; float *data, *sink; unsigned* ctr; Ctl* ctl;
  const long full = (long)npos * slot;
  CK(hipMalloc(&h1, full)); CK(hipMalloc(&h2, full));
  CK(hipMalloc(&data, (long)npos * 65536L * 4)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(data, 0, (long)npos * 65536L * 4));
  CK(hipMalloc(&ctr, (long)4 * npos * LINE * 4)); CK(hipMalloc(&ctl, sizeof(Ctl)));
  unsigned *ready1 = ctr, *done1 = ctr + (long)npos * LINE, *ready2 = ctr + (long)2 * npos * LINE, *done2 = ctr + (long)3 * npos * LINE;

  hipStream_t s0; CK(hipStreamCreate(&s0));
  hipEvent_t e0, e1, ea, eb, ec;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb)); CK(hipEventCreate(&ec));
  const float a = 1.0f;
  const double hb = 4.0 * npos * S * 65536.0 * sizeof(cf);  // hand-off bytes: written once and read once, twice

  auto reset = [&]() {
    CK(hipMemsetAsync(ctr, 0, (long)4 * npos * LINE * 4, s0));
    CK(hipMemsetAsync(ctl, 0, sizeof(Ctl), s0));
    CK(hipStreamSynchronize(s0));
  };
  auto report = [&](const char* what, float ms) {
    Ctl c; CK(hipMemcpy(&c, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    printf("%-34s %8.3f ms = %6.3f ms per 1000 positions, %5.2f TB/s on the hand-off bytes  errors %u timeouts %u abort %u\n",
           what, ms, ms * 1000.0 / npos, hb / ms / 1e9, c.errors, c.timeouts, c.abort_flag);
    fflush(stdout);
  };

#define LAUNCH_A(SYNC, st_, R) do { \
    if (sk == ST_PLAIN) kernA<SYNC, ST_PLAIN><<<npos * 16, 256, 0, st_>>>(h1, R, S, KA, a, ready1, done1, ctl); \
    else if (sk == ST_NT) kernA<SYNC, ST_NT><<<npos * 16, 256, 0, st_>>>(h1, R, S, KA, a, ready1, done1, ctl); \
    else kernA<SYNC, ST_SC1><<<npos * 16, 256, 0, st_>>>(h1, R, S, KA, a, ready1, done1, ctl); } while (0)
#define LAUNCH_B2(SYNC, SKK, LKK, st_, Ra, Rb) do { \
    if (check) kernB<SYNC, SKK, LKK, 1><<<npos * 16, 512, 0, st_>>>(h1, h2, Ra, Rb, KB, a, ready1, done1, ready2, done2, ctl, data, sink); \
    else kernB<SYNC, SKK, LKK, 0><<<npos * 16, 512, 0, st_>>>(h1, h2, Ra, Rb, KB, a, ready1, done1, ready2, done2, ctl, data, sink); } while (0)
#define LAUNCH_B(SYNC, st_, Ra, Rb) do { \
    if (sk == ST_PLAIN && lk == LD_ACQ) LAUNCH_B2(SYNC, ST_PLAIN, LD_ACQ, st_, Ra, Rb); \
    else if (sk == ST_PLAIN) LAUNCH_B2(SYNC, ST_PLAIN, LD_SC1, st_, Ra, Rb); \
    else if (sk == ST_NT && lk == LD_ACQ) LAUNCH_B2(SYNC, ST_NT, LD_ACQ, st_, Ra, Rb); \
    else if (sk == ST_NT) LAUNCH_B2(SYNC, ST_NT, LD_SC1, st_, Ra, Rb); \
    else if (lk == LD_ACQ) LAUNCH_B2(SYNC, ST_SC1, LD_ACQ, st_, Ra, Rb); \
    else LAUNCH_B2(SYNC, ST_SC1, LD_SC1, st_, Ra, Rb); } while (0)
#define LAUNCH_C(SYNC, st_, R, JJ) do { \
    if (lk == LD_ACQ) { if (check) kernC<SYNC, LD_ACQ, 1><<<64 * (JJ), 256, 0, st_>>>(h2, R, npos, JJ, KC, a, ready2, done2, ctl, sink); \
                        else kernC<SYNC, LD_ACQ, 0><<<64 * (JJ), 256, 0, st_>>>(h2, R, npos, JJ, KC, a, ready2, done2, ctl, sink); } \
    else { if (check) kernC<SYNC, LD_SC1, 1><<<64 * (JJ), 256, 0, st_>>>(h2, R, npos, JJ, KC, a, ready2, done2, ctl, sink); \
           else kernC<SYNC, LD_SC1, 0><<<64 * (JJ), 256, 0, st_>>>(h2, R, npos, JJ, KC, a, ready2, done2, ctl, sink); } } while (0)

  // ---- sequential: full-size buffers, whole chip, one stream -----------------------
  if (mode == "all" || mode == "seq") {
    const int Jseq = 36;  // the product's pass 2 walks chunks of ~28 positions
    for (int rep = 0; rep < reps; ++rep) {
      reset();
      float ta, tb, tc, tt;
      CK(hipEventRecord(e0, s0));
      LAUNCH_A(0, s0, npos);
      CK(hipEventRecord(ea, s0));
      LAUNCH_B(0, s0, npos, npos);
      CK(hipEventRecord(eb, s0));
      LAUNCH_C(0, s0, npos, Jseq);
      CK(hipEventRecord(e1, s0));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ta, e0, ea)); CK(hipEventElapsedTime(&tb, ea, eb)); CK(hipEventElapsedTime(&tc, eb, e1));
      CK(hipEventElapsedTime(&tt, e0, e1));
      printf("seq: A %.3f  B %.3f  C %.3f ms\n", ta, tb, tc);
      report("sequential, full-size hand-offs", tt);
    }
  }

  // ---- CU-masked streams -----------------------------------------------------------
  if (mode == "all" || mode == "ring" || mode == "census") {
    const int words = 8;  // 256 CUs
    std::vector<uint32_t> mA(words, 0), mB(words, 0), mC(words, 0);
    mask_range(mA, 0, nA); mask_range(mB, nA, nA + nB); mask_range(mC, nA + nB, nA + nB + nC);
    hipStream_t sA, sB, sC;
    CK(hipExtStreamCreateWithCUMask(&sA, words, mA.data()));
    CK(hipExtStreamCreateWithCUMask(&sB, words, mB.data()));
    CK(hipExtStreamCreateWithCUMask(&sC, words, mC.data()));
    {
      const int G = 2048;
      unsigned* cz; CK(hipMalloc(&cz, 3 * G * 4));
      census<<<G, 256, 0, sA>>>(cz); census<<<G, 256, 0, sB>>>(cz + G); census<<<G, 256, 0, sC>>>(cz + 2 * G);
      CK(hipDeviceSynchronize());
      std::vector<unsigned> hz(3 * G); CK(hipMemcpy(hz.data(), cz, 3 * G * 4, hipMemcpyDeviceToHost));
      std::set<unsigned> sets[3];
      int perx[3][8] = {};
      for (int k = 0; k < 3; ++k) for (int i = 0; i < G; ++i) sets[k].insert(hz[k * G + i]);
      for (int k = 0; k < 3; ++k) for (unsigned key : sets[k]) perx[k][(key >> 16) & 7]++;
      int overlap = 0;
      for (unsigned key : sets[0]) overlap += sets[1].count(key) + sets[2].count(key);
      for (unsigned key : sets[1]) overlap += sets[2].count(key);
      for (int k = 0; k < 3; ++k) {
        printf("census stream %c: %zu distinct CUs; per XCD:", 'A' + k, sets[k].size());
        for (int x = 0; x < 8; ++x) printf(" %d", perx[k][x]);
        printf("\n");
      }
      printf("census: CUs shared between masked streams: %d\n", overlap);
      CK(hipFree(cz));
    }
    if (mode != "census") {
      // each kernel alone on its CU share (full-size buffers, no flags): what the share costs
      {
        reset();
        float ta, tb, tc;
        CK(hipEventRecord(e0, sA)); LAUNCH_A(0, sA, npos); CK(hipEventRecord(ea, sA)); CK(hipEventSynchronize(ea));
        CK(hipEventElapsedTime(&ta, e0, ea));
        CK(hipEventRecord(e0, sB)); LAUNCH_B(0, sB, npos, npos); CK(hipEventRecord(eb, sB)); CK(hipEventSynchronize(eb));
        CK(hipEventElapsedTime(&tb, e0, eb));
        CK(hipEventRecord(e0, sC)); LAUNCH_C(0, sC, npos, J); CK(hipEventRecord(ec, sC)); CK(hipEventSynchronize(ec));
        CK(hipEventElapsedTime(&tc, e0, ec));
        Ctl c; CK(hipMemcpy(&c, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
        printf("alone on its CU share (no flags): A %.3f  B %.3f  C %.3f ms  errors %u\n", ta, tb, tc, c.errors);
      }
      for (int rep = 0; rep < reps; ++rep) {
        reset();
        CK(hipEventRecord(e0, s0));
        CK(hipStreamWaitEvent(sA, e0, 0)); CK(hipStreamWaitEvent(sB, e0, 0)); CK(hipStreamWaitEvent(sC, e0, 0));
        // consumers first: were the masks ignored, they would fill the chip and the spins time out (bounded)
        LAUNCH_C(1, sC, R2, J);
        LAUNCH_B(1, sB, R1, R2);
        LAUNCH_A(1, sA, R1);
        CK(hipEventRecord(ea, sA)); CK(hipEventRecord(eb, sB)); CK(hipEventRecord(ec, sC));
        CK(hipStreamWaitEvent(s0, ea, 0)); CK(hipStreamWaitEvent(s0, eb, 0)); CK(hipStreamWaitEvent(s0, ec, 0));
        CK(hipEventRecord(e1, s0));
        CK(hipEventSynchronize(e1));
        float tt, ta, tb, tc;
        CK(hipEventElapsedTime(&tt, e0, e1));
        CK(hipEventElapsedTime(&ta, e0, ea)); CK(hipEventElapsedTime(&tb, e0, eb)); CK(hipEventElapsedTime(&tc, e0, ec));
        printf("ring: A done at %.3f  B %.3f  C %.3f ms\n", ta, tb, tc);
        report("resident together, ring hand-offs", tt);
      }
    }
  }
  return 0;
}

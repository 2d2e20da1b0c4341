// Issue / pipe cost of the vector instructions the FFT kernels are made of, on
// gfx950: cycles per wave64 instruction on one SIMD at 1, 2, 3, 4 waves per
// SIMD (s_memtime ticks of one wave / instructions of that wave, and the
// aggregate: SIMD cycles per instruction of ANY wave).
//   hipcc --offload-arch=gfx950 -O3 valu_probe.hip -o valu_probe && ./valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
enum Op { FMA, ADD, MOV, PKFMA, PKADD, PKMUL, PKMOV, PKFMA_NEG, ADD3U, LSHLADD64, ADDCO, MIX_PK_INT, MFMA16, MFMA32, DSREAD64, DSWRITE64, NOPS };
static const char* names[] = {"v_fma_f32", "v_add_f32", "v_mov_b32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_mov_b32", "v_pk_fma_f32 neg/op_sel", "v_add3_u32", "v_lshl_add_u64", "v_add_co+v_addc_co (pair)", "4 v_pk_fma + 4 v_add3_u32 (per op)", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_32x32x2_f32", "ds_read_b64", "ds_write_b64", "s_nop 0"};

template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, long long* ticks, int iters) {
  __shared__ f2 lds[2048];
  f2 a[8], b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
  float s[8];
  unsigned u[8];
  unsigned long long w[8];
  f4 acc4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  typedef float f16 __attribute__((ext_vector_type(16)));
  f16 acc16 = {0};
  for (int i = 0; i < 8; ++i) {
    a[i] = f2{(float)threadIdx.x + i, 1.0f};
    s[i] = threadIdx.x * 0.5f + i;
    u[i] = threadIdx.x + i;
    w[i] = threadIdx.x * 8ull + i;
  }
  lds[threadIdx.x] = a[0];
  lds[threadIdx.x + 1024] = a[1];
  __syncthreads();
  const unsigned la = (threadIdx.x % 1024) * 8;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      if (OP == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b.x), "v"(c.x));
        REP8(X)
#undef X
      } else if (OP == ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(b.x));
        REP8(X)
#undef X
      } else if (OP == MOV) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(s[i]) : "v"(b.x));
        REP8(X)
#undef X
      } else if (OP == PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == PKMUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == PKMOV) {
#define X(i) asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=v"(a[i]) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == PKFMA_NEG) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == ADD3U) {
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        REP8(X)
#undef X
      } else if (OP == LSHLADD64) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
        REP8(X)
#undef X
      } else if (OP == ADDCO) {
#define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]), "v"(u[(i + 3) & 7]) : "vcc");
        REP8(X)
#undef X
      } else if (OP == MIX_PK_INT) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_add3_u32 %1, %1, %1, %1" : "+v"(a[i]), "+v"(u[i]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == MFMA16) {
#define X(i) acc4[i & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(s[i], b.x, acc4[i & 1], 0, 0, 0);
        REP8(X)
#undef X
      } else if (OP == MFMA32) {
#define X(i) acc16 = __builtin_amdgcn_mfma_f32_32x32x2f32(s[i], b.x, acc16, 0, 0, 0);
        REP8(X)
#undef X
      } else if (OP == DSREAD64) {
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[i]) : "v"(la), "n"(i * 8));
        REP8(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (OP == DSWRITE64) {
#define X(i) asm volatile("ds_write_b64 %0, %1" : : "v"(la), "v"(a[i]) : "memory");
        REP8(X)
#undef X
      } else if (OP == NOPS) {
#define X(i) asm volatile("s_nop 0");
        REP8(X)
#undef X
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y + s[i] + (float)u[i] + (float)w[i];
  r += acc4[0].x + acc4[1].y + acc16[3];
  if (r == 12345.678f) out[0] = r;
  if ((threadIdx.x & 63) == 0)
    ticks[(blockIdx.x * (blockDim.x / 64)) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(int waves_per_simd, float* out, long long* ticks, double* wall_ns_per_inst, double* tick_per_inst) {
  const int iters = 2000, threads = 256 * waves_per_simd, grid = 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(threads), 0, 0, out, ticks, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(threads), 0, 0, out, ticks, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const int nw = grid * threads / 64;
  std::vector<long long> h(nw);
  CK(hipMemcpy(h.data(), ticks, nw * sizeof(long long), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double ninst = (double)iters * 32 * (OP == ADDCO || OP == MIX_PK_INT ? 2 : 1);
  *tick_per_inst = h[nw / 2] / ninst;
  *wall_ns_per_inst = ms * 1e6 / (ninst * waves_per_simd);  // per SIMD: every SIMD runs waves_per_simd waves
}

int main() {
  float* out; long long* ticks;
  CK(hipMalloc(&out, 4)); CK(hipMalloc(&ticks, 256 * 16 * 8));
  printf("%-36s | per-wave ticks/inst at 1,2,3,4 waves/SIMD | SIMD ns per inst (any wave) at 1,2,3,4\n", "instruction");
#define ROW(OP) { double w[4], t[4]; for (int n = 1; n <= 4; ++n) run<OP>(n, out, ticks, &w[n-1], &t[n-1]); \
  printf("%-36s | %6.2f %6.2f %6.2f %6.2f | %6.3f %6.3f %6.3f %6.3f\n", names[OP], t[0], t[1], t[2], t[3], w[0], w[1], w[2], w[3]); }
  ROW(FMA) ROW(ADD) ROW(MOV) ROW(PKFMA) ROW(PKADD) ROW(PKMUL) ROW(PKMOV) ROW(PKFMA_NEG) ROW(ADD3U) ROW(LSHLADD64) ROW(ADDCO) ROW(MIX_PK_INT) ROW(MFMA16) ROW(MFMA32) ROW(DSREAD64) ROW(DSWRITE64) ROW(NOPS)
  return 0;
}

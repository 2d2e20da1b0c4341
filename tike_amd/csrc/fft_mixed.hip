// Batched 2-D complex64 FFT of ANY tile size for gfx950 (round 6): the
// shape-general route behind tike.operators.Propagation -- the reference hands
// every shape to cuFFT (operators/cupy/propagation.py:43-73, cache.py:32-82).
//
//   n = 2^a 3^b 5^c 7^d 11^e 13^f <= 4096: mixed-radix Stockham lines in LDS
//       (fft_mixed.h), a row launch and a column launch (in place on `out`);
//   any other n <= 2048 (127, 45 * 23, primes ...): Bluestein's chirp-z over
//       the same engine with a power-of-two length M >= 2n - 1, both
//       transforms of length M and the spectrum product inside the workgroup,
//       the chirp and the transformed chirp from per-(n, device) tables.
//
// The power-of-two sizes 32..1024 keep their register engines (fft2.hip).
#include <cmath>
#include <complex>
#include <map>
#include <mutex>
#include <vector>

#include "fft_mixed.h"
#include "internal.h"
#include "tike_amd.h"

// ------------------------------------------------------------ plan cache
static std::mutex g_mix_mutex;
static std::map<long, MixTables*> g_mix_tables;  // key = device * 65536 + n

static void host_fft_pow2(std::vector<std::complex<double>>& x) {
  const size_t n = x.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(x[i], x[j]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    for (size_t i = 0; i < n; i += len)
      for (size_t k = 0; k < len / 2; ++k) {
        const std::complex<double> w = std::polar(1.0, -2.0 * M_PI * (double)k / (double)len);
        const std::complex<double> u = x[i + k], v = x[i + k + len / 2] * w;
        x[i + k] = u + v;
        x[i + k + len / 2] = u - v;
      }
  }
}

static const cf* upload(const std::vector<std::complex<double>>& h) {
  std::vector<cf> f(h.size());
  for (size_t i = 0; i < h.size(); ++i) f[i] = mk((float)h[i].real(), (float)h[i].imag());
  cf* d = nullptr;
  if (hipMalloc((void**)&d, sizeof(cf) * f.size()) != hipSuccess) return nullptr;
  if (hipMemcpy(d, f.data(), sizeof(cf) * f.size(), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(d);
    return nullptr;
  }
  return d;
}

const MixTables* tk_mix_tables(int n) {
  int dev = 0;
  if (n < 1 || hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_mix_mutex);
  const long key = (long)dev * 65536 + n;
  auto it = g_mix_tables.find(key);
  if (it != g_mix_tables.end()) return it->second;
  MixTables* t = new MixTables();
  t->n = n;
  t->chirp = t->bhat = nullptr;
  t->bluestein = !mix_make_plan(n, &t->plan);
  if (t->bluestein) {
    int M = 1;
    while (M < 2 * n - 1) M *= 2;
    if (n > TK_MIX_MAX_N || M > TK_MIX_MAX_N || !mix_make_plan(M, &t->plan)) {
      delete t;
      g_mix_tables[key] = nullptr;
      return nullptr;
    }
    // c_j = exp(-i pi j^2 / n), the angle reduced in integers (j^2 mod 2n)
    std::vector<std::complex<double>> c(n), b(M, 0.0);
    for (long j = 0; j < n; ++j)
      c[j] = std::polar(1.0, -M_PI * (double)((j * j) % (2L * n)) / (double)n);
    for (int j = 0; j < n; ++j) {
      b[j] = std::conj(c[j]);
      if (j) b[M - j] = std::conj(c[j]);
    }
    host_fft_pow2(b);
    for (auto& v : b) v /= (double)M;  // the 1 / M of the inverse of length M
    t->chirp = upload(c);
    t->bhat = upload(b);
    if (!t->chirp || !t->bhat) return nullptr;
  }
  const int m = t->plan.n;
  std::vector<std::complex<double>> w(m);
  for (int k = 0; k < m; ++k) w[k] = std::polar(1.0, -2.0 * M_PI * (double)k / (double)m);
  t->tw = upload(w);
  if (!t->tw) return nullptr;
  g_mix_tables[key] = t;
  return t;
}

// ---------------------------------------------------------------- kernels
// One launch = one direction of the 2-D transform.  Lines are numbered over the
// whole batch: row g = tile * n + y starts at element g * n (stride 1), column
// g = tile * n + x at tile * n^2 + x (stride n).  A workgroup takes groups of L
// consecutive lines (L a power of two): rows are then one contiguous run of
// L * n elements, columns L * 8-byte segments per tile row.
// LDS: [twiddles plan.n][buffer a: L * ls][buffer b: L * ls][line bases: L longs]
//
// Memory phases: every thread requests MIX_U elements before it uses the first
// (a run-time loop around one load is one request in flight per thread: 2 us of
// latency per element; 384^2: 0.36 -> 0.54 M tiles/s), and the first MIX_U x
// 256 elements of the NEXT group are requested before the butterflies of the
// group in hand; 3-5 workgroups per CU so that one group's butterflies run
// under the others' loads and stores (profiles/r06_experiments.md: occupancy
// counts for more than the depth of the request queue -- 8 requests at 4
// waves/SIMD with 92 B/lane of scratch lose 12 % against 4 requests at 3).
// Groups are dealt to the XCDs in contiguous ranges (workgroup b runs on XCD
// b % 8): the column groups that split a 128-byte line are neighbours in time
// AND share an L2.
#define MIX_U 4

struct MixGroups {
  long first, end, step;  // this workgroup's groups: first, first + step, ... < end
};
__device__ __forceinline__ MixGroups mix_groups(long ngroup) {
  const long nb = gridDim.x;
  MixGroups g;
  if (nb % 8 != 0 || ngroup < 64) {
    g.first = blockIdx.x;
    g.end = ngroup;
    g.step = nb;
  } else {
    const long per = (ngroup + 7) / 8, xcd = blockIdx.x & 7;
    g.first = xcd * per + (blockIdx.x >> 3);
    g.end = (xcd + 1) * per < ngroup ? (xcd + 1) * per : ngroup;
    g.step = nb >> 3;
  }
  return g;
}

template <bool INV, bool BLU, bool COLS>
__global__ __launch_bounds__(256) void mix_pass_kernel(const cf* in, cf* out, MixPlan p, int n,
                                                       long nlines, int L, int logL,
                                                       const cf* __restrict__ twg,
                                                       const cf* __restrict__ chirp,
                                                       const cf* __restrict__ bhat,
                                                       float scale) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  cf* twl = reinterpret_cast<cf*>(lds_raw);
  cf* bufa = twl + p.n;
  cf* bufb = bufa + (long)L * p.ls;
  long* lbase2 = reinterpret_cast<long*>(bufb + (long)L * p.ls);  // two sets of L
  for (int k = threadIdx.x; k < p.n; k += blockDim.x) twl[k] = twg[k];
  const float rcp_n = 1.0f / (float)n;
  const MixGroups mg = mix_groups((nlines + L - 1) / L);
  // element idx of group grp -> (line, e, address or -1); COLS: the lines a
  // short last group does not have are -1
  auto where = [&](long grp, const long* lbase, int idx, int& line, int& e, long& off) {
    // (opaque: (line, e) of a prefetch slot do not depend on the group; hoisted
    // out of the group loop they would stay alive across the butterflies)
    asm volatile("" : "+v"(idx));
    const long g0 = grp * L;
    const int nl = (int)(nlines - g0 < L ? nlines - g0 : L);
    if (COLS) {
      line = idx & (L - 1);
      e = idx >> logL;
      off = line < nl && e < n ? lbase[line] + (long)e * n : -1;
    } else {
      line = mix_div(idx, rcp_n);
      e = idx - line * n;
      off = idx < nl * n ? g0 * n + idx : -1;
    }
  };
  auto bases = [&](long grp, long* lbase) {
    if (COLS && threadIdx.x < L) {
      // (32-bit: ntile * n < 2^31 is checked by the launcher)
      const long g = grp * L + threadIdx.x, tile = (unsigned)g / (unsigned)n;
      lbase[threadIdx.x] = tile * (long)n * n + (g - tile * n);
    }
  };
  // the first MIX_U * 256 elements of a group travel in registers: requested
  // one group AHEAD, while the butterflies of the group in hand run
  cf pre[MIX_U];
  auto request = [&](long grp, const long* lbase) {
#pragma unroll
    for (int u = 0; u < MIX_U; ++u) {
      int line, e;
      long off;
      where(grp, lbase, threadIdx.x + u * 256, line, e, off);
      // (unconditionally -- a load behind a condition is a branch around the
      // load --: what the group does not have reads element 0)
      pre[u] = in[off >= 0 ? off : 0];
    }
  };
  int cur = 0;
  if (mg.first < mg.end) {
    bases(mg.first, lbase2);
    if (COLS) __syncthreads();
    request(mg.first, lbase2);
  }
  for (long grp = mg.first; grp < mg.end; grp += mg.step) {
    const long* lbase = lbase2 + cur * L;
    const long g0 = grp * L;
    const int nl = (int)(nlines - g0 < L ? nlines - g0 : L);
    const int total = COLS ? (n << logL) : nl * n;
    auto put = [&](int line, int e, cf x) {
      if (BLU) {
        if (INV) x = conjf(x);
        x = x * chirp[e];
      }
      bufa[line * p.ls + mix_pad(e)] = x;
    };
    // ---- the group's samples into LDS (BLU: x * chirp -- the conjugate of x
    // for the inverse --, zero fill up to M)
#pragma unroll
    for (int u = 0; u < MIX_U; ++u) asm volatile("" : "+v"(pre[u].x));
#pragma unroll
    for (int u = 0; u < MIX_U; ++u) {
      int line, e;
      long off;
      where(grp, lbase, threadIdx.x + u * 256, line, e, off);
      if (off >= 0) put(line, e, pre[u]);
    }
    for (int base = threadIdx.x + 256 * MIX_U; base < total; base += 256 * MIX_U) {
      cf v[MIX_U];
      int line[MIX_U], e[MIX_U];
      long off[MIX_U];
#pragma unroll
      for (int u = 0; u < MIX_U; ++u) {
        where(grp, lbase, base + u * 256, line[u], e[u], off[u]);
        v[u] = in[off[u] >= 0 ? off[u] : 0];
      }
#pragma unroll
      for (int u = 0; u < MIX_U; ++u) asm volatile("" : "+v"(v[u].x));
#pragma unroll
      for (int u = 0; u < MIX_U; ++u)
        if (off[u] >= 0) put(line[u], e[u], v[u]);
    }
    if (BLU) {
      const int padn = p.n - n;  // zeros behind the n samples, up to M
      const float rcp_pad = 1.0f / (float)padn;
      for (int idx = threadIdx.x; idx < nl * padn; idx += blockDim.x) {
        const int line = mix_div(idx, rcp_pad), e = n + idx - line * padn;
        bufa[line * p.ls + mix_pad(e)] = mk(0.f, 0.f);
      }
    }
    // ---- the next group's request goes out before this group's butterflies
    // (in place: this group's samples have been read above -- program order on
    // the same addresses only when in == out AND the groups coincide, which
    // they never do)
    const long nxt = grp + mg.step;
    if (nxt < mg.end) bases(nxt, lbase2 + (cur ^ 1) * L);
    __syncthreads();
    if (nxt < mg.end) request(nxt, lbase2 + (cur ^ 1) * L);
    cf* res;
    if (BLU) {
      res = mix_stages<false>(bufa, bufb, twl, p, nl);
      cf* other = res == bufa ? bufb : bufa;
      const float rcp_m = 1.0f / (float)p.n;
      for (int idx = threadIdx.x; idx < nl * p.n; idx += blockDim.x) {
        const int line = mix_div(idx, rcp_m), e = idx - line * p.n;
        cf* q = res + line * p.ls + mix_pad(e);
        *q = *q * bhat[e];
      }
      __syncthreads();
      res = mix_stages<true>(res, other, twl, p, nl);
    } else {
      res = mix_stages<INV>(bufa, bufb, twl, p, nl);
    }
    // ---- store
    for (int idx = threadIdx.x; idx < total; idx += 256) {
      int line, e;
      long off;
      where(grp, lbase, idx, line, e, off);
      if (off < 0) continue;
      cf v = res[line * p.ls + mix_pad(e)];
      if (BLU) {
        v = v * chirp[e];
        if (INV) v = conjf(v);
      }
      out[off] = v * scale;
    }
    __syncthreads();
    cur ^= 1;
  }
}

// Lines per group: as many as fit `budget` bytes of LDS beside the twiddles, a
// power of two, at most `cap`.
static int mix_lines_per_group(const MixPlan& p, size_t budget, int cap) {
  const size_t tw = sizeof(cf) * (size_t)p.n,
               per = 2 * sizeof(cf) * (size_t)p.ls + 2 * sizeof(long);
  if (budget <= tw + per) return 0;
  size_t fit = (budget - tw) / per;
  int L = 1;
  while ((size_t)(2 * L) <= fit && 2 * L <= cap) L *= 2;
  return L;
}

template <bool INV, bool BLU, bool COLS>
static int launch_mix_pass(const cf* in, cf* out, const MixTables* t, long nlines, int L,
                           float scale, hipStream_t stream) {
  int logL = 0;
  while ((1 << logL) < L) ++logL;
  const size_t lds = sizeof(cf) * (size_t)t->plan.n +
                     (2 * sizeof(cf) * (size_t)t->plan.ls + 2 * sizeof(long)) * (size_t)L;
  auto kern = mix_pass_kernel<INV, BLU, COLS>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const long ngroup = (nlines + L - 1) / L;
  hipLaunchKernelGGL(kern, dim3(tk_grid(ngroup, 8)), dim3(256), lds, stream, in, out, t->plan,
                     t->n, nlines, L, logL, t->tw, t->chirp, t->bhat, scale);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

template <bool INV, bool BLU>
static int mix_fft2(const cf* in, cf* out, const MixTables* t, long ntile, float scale,
                    int l_rows, int l_cols, hipStream_t stream) {
  const long nlines = ntile * t->n;
  // the caller's grouping (tests, tuning): a power of two that fits; else
  // rows: a group is one contiguous run of memory whatever L is, so L only has
  // to give the workgroup enough butterflies per stage (>= 2 per thread) --
  // small groups, five workgroups per CU; columns: L sets the width of every
  // access (L * 8 bytes): 64-byte segments where three workgroups still fit a
  // CU, 32-byte ones else (the other half of the line: same XCD, same time)
  auto fits = [&](size_t budget, int cap) { return mix_lines_per_group(t->plan, budget, cap); };
  auto forced = [&](int want) {
    int L = 1;
    while (2 * L <= want) L *= 2;
    const int fit = fits(150 * 1024, 1 << 20);
    return L < fit ? L : fit;
  };
  int Lr, Lc;
  if (l_rows > 0) {
    Lr = forced(l_rows);
  } else {
    int want = 1;
    while (want * t->plan.n < 2048 && want < 64) want *= 2;
    Lr = fits(30 * 1024, want);
    if (Lr < 1) Lr = fits(150 * 1024, 1);
  }
  if (l_cols > 0) {
    Lc = forced(l_cols);
  } else {
    Lc = fits(52 * 1024, t->plan.n <= 64 ? 64 : 8);
    if (Lc < 4) Lc = fits(76 * 1024, 4);
    if (Lc < 2) Lc = fits(150 * 1024, 2);
  }
  if (Lr < 1 || Lc < 1) return TK_ERR_UNSUPPORTED;
  int rc = launch_mix_pass<INV, BLU, false>(in, out, t, nlines, Lr, 1.0f, stream);
  if (rc) return rc;
  return launch_mix_pass<INV, BLU, true>(out, out, t, nlines, Lc, scale, stream);
}

int tk_fft2_general(const cf* in, cf* out, long ntile, int n, int inverse, float scale,
                    int l_rows, int l_cols, hipStream_t stream) {
  TK_CHECK_ARG(in && out && n >= 1 && ntile >= 0);
  if (ntile == 0) return TK_OK;
  if (ntile * (long)n >= (1L << 31)) return TK_ERR_ARG;
  const MixTables* t = tk_mix_tables(n);
  if (!t) return TK_ERR_UNSUPPORTED;
  if (t->bluestein)
    return inverse ? mix_fft2<true, true>(in, out, t, ntile, scale, l_rows, l_cols, stream)
                   : mix_fft2<false, true>(in, out, t, ntile, scale, l_rows, l_cols, stream);
  return inverse ? mix_fft2<true, false>(in, out, t, ntile, scale, l_rows, l_cols, stream)
                 : mix_fft2<false, false>(in, out, t, ntile, scale, l_rows, l_cols, stream);
}

extern "C" int tike_fft2_general(const void* in, void* out, long ntile, int n, int inverse,
                                 float scale, int lines_per_group_rows,
                                 int lines_per_group_cols, void* stream) {
  TK_ENTER();
  return tk_fft2_general((const cf*)in, (cf*)out, ntile, n, inverse, scale,
                         lines_per_group_rows, lines_per_group_cols, (hipStream_t)stream);
}

extern "C" int tike_fft2_supported(int n) {
  if (n < 1) return 0;
  if (n <= TK_MIX_MAX_N) {
    MixPlan p;
    if (mix_make_plan(n, &p)) return 1;
  }
  int M = 1;
  while (M < 2 * n - 1 && M <= TK_MIX_MAX_N) M *= 2;
  return M <= TK_MIX_MAX_N ? 1 : 0;
}

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
  long* lbase = reinterpret_cast<long*>(bufb + (long)L * p.ls);
  for (int k = threadIdx.x; k < p.n; k += blockDim.x) twl[k] = twg[k];
  const float rcp_n = 1.0f / (float)n;
  const long ngroup = (nlines + L - 1) / L;
  for (long grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
    const long g0 = grp * L;
    const int nl = (int)(nlines - g0 < L ? nlines - g0 : L);
    if (COLS) {
      if (threadIdx.x < nl) {
        const long g = g0 + threadIdx.x, tile = g / n;
        lbase[threadIdx.x] = tile * (long)n * n + (g - tile * n);
      }
      __syncthreads();
    }
    // ---- load (BLU: x * chirp -- the conjugate of x for the inverse --, zero fill)
    const int total = COLS ? (n << logL) : nl * n;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
      int line, e;
      long src;
      if (COLS) {
        line = idx & (L - 1);
        e = idx >> logL;
        if (line >= nl) continue;
        src = lbase[line] + (long)e * n;
      } else {
        line = mix_div(idx, rcp_n);
        e = idx - line * n;
        src = g0 * n + idx;
      }
      cf v = in[src];
      if (BLU) {
        if (INV) v = conjf(v);
        v = v * chirp[e];
      }
      bufa[line * p.ls + mix_pad(e)] = v;
    }
    if (BLU) {
      const int padn = p.n - n;  // zeros behind the n samples, up to M
      const float rcp_pad = 1.0f / (float)padn;
      for (int idx = threadIdx.x; idx < nl * padn; idx += blockDim.x) {
        const int line = mix_div(idx, rcp_pad), e = n + idx - line * padn;
        bufa[line * p.ls + mix_pad(e)] = mk(0.f, 0.f);
      }
    }
    __syncthreads();
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
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
      int line, e;
      long dst;
      if (COLS) {
        line = idx & (L - 1);
        e = idx >> logL;
        if (line >= nl) continue;
        dst = lbase[line] + (long)e * n;
      } else {
        line = mix_div(idx, rcp_n);
        e = idx - line * n;
        dst = g0 * n + idx;
      }
      cf v = res[line * p.ls + mix_pad(e)];
      if (BLU) {
        v = v * chirp[e];
        if (INV) v = conjf(v);
      }
      out[dst] = v * scale;
    }
    __syncthreads();
  }
}

// Lines per group: as many as fit `budget` bytes of LDS beside the twiddles, a
// power of two, at most `cap`.
static int mix_lines_per_group(const MixPlan& p, size_t budget, int cap) {
  const size_t tw = sizeof(cf) * (size_t)p.n, per = 2 * sizeof(cf) * (size_t)p.ls + sizeof(long);
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
                     (2 * sizeof(cf) * (size_t)t->plan.ls + sizeof(long)) * (size_t)L;
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
  // two workgroups per CU where the lines allow it (64 KiB each), else one
  auto pick = [&](int want, int cap) {
    if (want > 0) {  // the caller's grouping (tests, tuning): a power of two that fits
      int L = 1;
      while (2 * L <= want) L *= 2;
      const int fit = mix_lines_per_group(t->plan, 150 * 1024, 1 << 20);
      return L < fit ? L : fit;
    }
    int L = mix_lines_per_group(t->plan, 64 * 1024, cap);
    if (L < 4) L = mix_lines_per_group(t->plan, 150 * 1024, cap < 4 ? cap : 4);
    return L;
  };
  // rows: a group is one contiguous run of memory whatever L is; columns: L
  // sets the width of every access (L * 8 bytes)
  const int cap = t->plan.n <= 64 ? 64 : 16;
  const int Lr = pick(l_rows, cap), Lc = pick(l_cols, cap);
  if (Lr < 1 || Lc < 1) return TK_ERR_UNSUPPORTED;
  int rc = launch_mix_pass<INV, BLU, false>(in, out, t, nlines, Lr, 1.0f, stream);
  if (rc) return rc;
  return launch_mix_pass<INV, BLU, true>(out, out, t, nlines, Lc, scale, stream);
}

int tk_fft2_general(const cf* in, cf* out, long ntile, int n, int inverse, float scale,
                    int l_rows, int l_cols, hipStream_t stream) {
  TK_CHECK_ARG(in && out && n >= 1 && ntile >= 0);
  if (ntile == 0) return TK_OK;
  if (ntile * (long)n >= (1L << 40)) return TK_ERR_ARG;
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

// Batched 2-D complex64 FFT for gfx950: replaces cuFFT behind
// tike.operators.Propagation (reference propagation.py:43-73, cache.py:66-82).
//
// Power-of-two n in [32, 1024]: one workgroup owns one n x n tile and runs
// both passes (rows, then columns in place on the output tile) through the
// register/LDS engine of fft_engine.h.  Every other n: the shape-general
// engine of fft_mixed.hip (mixed radix 2/3/5/7/11/13 up to 4096, Bluestein for
// the rest up to 2048; until round 5 a direct O(n^2) DFT per line).
#include <cmath>
#include <mutex>

#include "fft_engine2.h"
#include "internal.h"
#include "tike_amd.h"

// ---------------------------------------------------------------- twiddles
static cf* g_tw_dev[64] = {nullptr};
static std::mutex g_tw_mutex;

const cf* tk_twiddles() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(g_tw_mutex);
  if (g_tw_dev[dev]) return g_tw_dev[dev];
  static cf host[4096];  // w_n^k at [n + k], n = 32 ... 2048
  for (int n = 32; n <= 2048; n *= 2)
    for (int k = 0; k < n; ++k) {
      const double a = -2.0 * M_PI * (double)k / (double)n;
      host[n + k] = mk((float)std::cos(a), (float)std::sin(a));
    }
  cf* d = nullptr;
  if (hipMalloc((void**)&d, sizeof(host)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, host, sizeof(host), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  g_tw_dev[dev] = d;
  return d;
}

extern "C" int tike_abi_version(void) { return TIKE_ABI_VERSION; }

#include "build_id.h"
extern "C" const char* tike_build_id(void) { return TIKE_BUILD_ID; }

extern "C" int tike_init(void) { return tk_twiddles() ? TK_OK : (int)hipErrorNotInitialized; }

// ------------------------------------------------------- deterministic mode
static bool g_det_on = false;
static float* g_det_scratch = nullptr;
static size_t g_det_bytes = 0;

extern "C" int tike_set_deterministic(int on, void* scratch, long bytes) {
  TK_CHECK_ARG(bytes >= 0 && (scratch != nullptr || bytes == 0));
  g_det_on = on != 0;
  g_det_scratch = on ? (float*)scratch : nullptr;
  g_det_bytes = on ? (size_t)bytes : 0;
  return TK_OK;
}

bool tk_deterministic() { return g_det_on; }

float* tk_det_scratch(size_t bytes) {
  return g_det_on && bytes <= g_det_bytes ? g_det_scratch : nullptr;
}

__global__ __launch_bounds__(256) void ordered_sum_kernel(float* __restrict__ out,
                                                          const float* __restrict__ part,
                                                          long n, int nparts, int accumulate,
                                                          int out_stride) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    float s = accumulate ? out[i * out_stride] : 0.f;
    for (int c = 0; c < nparts; ++c) s += part[(long)c * n + i];
    out[i * out_stride] = s;
  }
}

int tk_ordered_sum(float* out, const float* part, long n, int nparts, bool accumulate,
                   hipStream_t stream, int out_stride) {
  if (n <= 0) return TK_OK;
  hipLaunchKernelGGL(ordered_sum_kernel, dim3(tk_grid((n + 255) / 256, 8)), dim3(256), 0, stream,
                     out, part, n, nparts, accumulate ? 1 : 0, out_stride);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// costs[n] = sum_slot part[n * nslots + slot], slots ascending
__global__ __launch_bounds__(256) void cost_finish_kernel(float* __restrict__ costs,
                                                          const float* __restrict__ part,
                                                          long nscan, int nslots) {
  for (long n = blockIdx.x * 256L + threadIdx.x; n < nscan; n += gridDim.x * 256L) {
    float s = 0.f;
    for (int k = 0; k < nslots; ++k) s += part[n * nslots + k];
    costs[n] = s;
  }
}

int tk_cost_sink(float* costs, long nscan, int nslots, hipStream_t stream, TkCostSink* sink) {
  sink->costs = costs;
  sink->part = nullptr;
  sink->nslots = nslots;
  if (costs == nullptr) return TK_OK;
  if (tk_deterministic()) {
    sink->part = tk_det_scratch(sizeof(float) * (size_t)nscan * nslots);
    if (sink->part == nullptr) return TK_ERR_ARG;  // scratch buffer too small
    return TK_OK;
  }
  hipError_t e = hipMemsetAsync(costs, 0, sizeof(float) * (size_t)nscan, stream);
  return e == hipSuccess ? TK_OK : (int)e;
}

int tk_cost_finish(const TkCostSink& sink, long nscan, hipStream_t stream) {
  if (sink.part == nullptr || sink.costs == nullptr || nscan <= 0) return TK_OK;
  hipLaunchKernelGGL(cost_finish_kernel, dim3(tk_grid((nscan + 255) / 256, 8)), dim3(256), 0,
                     stream, sink.costs, sink.part, nscan, sink.nslots);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

// ------------------------------------------------------------ pow2 kernel
template <int N, bool INV>
__global__ __launch_bounds__(FftPlan<N>::NT, FftPlan<N>::MINW) void fft2_pow2_kernel(
    const cf* in, cf* out, long ntile, float scale, const cf* __restrict__ twtab) {
  using G = FftGeom<N>;
  __shared__ cf lds[G::LDS_ELEMS];
  FftTw<N> tw;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const cf* src = in + tile * (long)N * N;
    cf* dst = out + tile * (long)N * N;
    const FftLane<N, false> row = fft_lane<N, false>();
    tw.init(twtab, row.j);
    for (int g = 0; g < N; g += G::L) {
      fft_lines<N, INV, false>(
          lds, row, tw, [&](int line, int e) { return src[(g + line) * N + e]; },
          [&](int line, int e, cf v) { dst[(g + line) * N + e] = v; });
    }
    __syncthreads();  // row pass results visible to the whole workgroup
    const FftLane<N, true> col = fft_lane<N, true>();
    tw.init(twtab, col.j);
    for (int g = 0; g < N; g += G::L) {
      fft_lines<N, INV, true>(
          lds, col, tw, [&](int line, int e) { return dst[e * N + g + line]; },
          [&](int line, int e, cf v) { dst[e * N + g + line] = v * scale; });
    }
    __syncthreads();
  }
}

// ------------------------------------------------ v2 kernel (out of place)
// One workgroup (N threads) per tile: RB pass-1 iterations of 16 strided rows
// each, then 16 barrier-free pass-2 iterations in place on the output tile
// (fft_engine2.h).  Requires in != out.
template <int N, bool INV>
__global__ __launch_bounds__(N, TK_V2_MINW(N)) void fft2_v2_kernel(
    const cf* __restrict__ in, cf* __restrict__ out, long ntile, float scale,
    const cf* __restrict__ twtab) {
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
      fft2_pass1<N, INV>(lds, twtab, tw, line, j, r,
                         [&](int y, int e, auto) { return tk_ld_stream(src + y * N + e); }, dst);
    __syncthreads();
    for (int k1 = 0; k1 < 16; ++k1)
      fft2_pass2<N, INV>(dst, k1,
                         [&](int ky, int t, cf v) { tk_st_stream(dst + ky * N + t, v * scale); });
    __syncthreads();
  }
}

template <int N>
static int launch_v2(const cf* in, cf* out, long ntile, int inverse, float scale,
                     hipStream_t stream) {
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
  const int grid = tk_grid(ntile, 4);
  if (inverse)
    hipLaunchKernelGGL((fft2_v2_kernel<N, true>), dim3(grid), dim3(N), 0, stream, in, out, ntile,
                       scale, tw);
  else
    hipLaunchKernelGGL((fft2_v2_kernel<N, false>), dim3(grid), dim3(N), 0, stream, in, out,
                       ntile, scale, tw);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

template <int N>
static int launch_pow2(const cf* in, cf* out, long ntile, int inverse, float scale,
                       hipStream_t stream) {
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
  const int grid = tk_grid(ntile, N >= 512 ? 2 : 4);
  if (inverse)
    hipLaunchKernelGGL((fft2_pow2_kernel<N, true>), dim3(grid), dim3(FftPlan<N>::NT), 0, stream,
                       in, out, ntile, scale, tw);
  else
    hipLaunchKernelGGL((fft2_pow2_kernel<N, false>), dim3(grid), dim3(FftPlan<N>::NT), 0,
                       stream, in, out, ntile, scale, tw);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

int tk_fft2(const cf* in, cf* out, long ntile, int n, int inverse, float scale,
            hipStream_t stream) {
  TK_CHECK_ARG(in && out && n >= 1 && ntile >= 0);
  if (ntile == 0) return TK_OK;
  if (in != out) {
    switch (n) {
      case 128: return launch_v2<128>(in, out, ntile, inverse, scale, stream);
      case 256: return launch_v2<256>(in, out, ntile, inverse, scale, stream);
      case 512: return launch_v2<512>(in, out, ntile, inverse, scale, stream);
      default: break;
    }
  }
  switch (n) {
    case 32: return launch_pow2<32>(in, out, ntile, inverse, scale, stream);
    case 64: return launch_pow2<64>(in, out, ntile, inverse, scale, stream);
    case 128: return launch_pow2<128>(in, out, ntile, inverse, scale, stream);
    case 256: return launch_pow2<256>(in, out, ntile, inverse, scale, stream);
    case 512: return launch_pow2<512>(in, out, ntile, inverse, scale, stream);
    case 1024: return launch_pow2<1024>(in, out, ntile, inverse, scale, stream);
    default: break;
  }
  // every other size: mixed-radix lines in LDS, or Bluestein over them
  // (fft_mixed.hip) -- the reference's cuFFT takes any shape
  return tk_fft2_general(in, out, ntile, n, inverse, scale, 0, 0, stream);
}

extern "C" int tike_fft2(const void* in, void* out, long ntile, int n, int inverse, float scale,
                         void* stream) {
  TK_ENTER();
  return tk_fft2((const cf*)in, (cf*)out, ntile, n, inverse, scale, (hipStream_t)stream);
}

// ---------------------------------------------- Fresnel spectrum propagation
// out = IFFT2( FFT2(in) * H )  (adjoint: conj(H)) for `ntile` n x n tiles
// sharing the propagator H (n, n) -- the near-field step between the slices
// of a multislice object (reference operators/cupy/fresnelspectprop.py:52-113).
__global__ __launch_bounds__(256) void spectrum_multiply_kernel(cf* __restrict__ x,
                                                                const cf* __restrict__ h,
                                                                long ntile, long npix,
                                                                int conjugate) {
  const long total = ntile * npix;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    cf w = h[i % npix];
    if (conjugate) w = conjf(w);
    x[i] = x[i] * w;
  }
}

extern "C" int tike_fresnel_spect_prop(const void* in, void* out, const void* propagator,
                                       long ntile, int n, int adjoint, float fwd_scale,
                                       float inv_scale, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(ntile >= 0 && n >= 1);
  if (ntile == 0) return TK_OK;
  TK_CHECK_ARG(in && out && propagator);
  int rc = tk_fft2((const cf*)in, (cf*)out, ntile, n, 0, fwd_scale, stream);
  if (rc) return rc;
  const long npix = (long)n * n;
  hipLaunchKernelGGL(spectrum_multiply_kernel, dim3(tk_grid((ntile * npix + 255) / 256, 8)),
                     dim3(256), 0, stream, (cf*)out, (const cf*)propagator, ntile, npix, adjoint);
  TK_LAUNCH_CHECK();
  return tk_fft2((const cf*)out, (cf*)out, ntile, n, 1, inv_scale, stream);
}

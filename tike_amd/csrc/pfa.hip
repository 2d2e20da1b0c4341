// (Round 6, late: the same kernels also serve 1024 = 4 x 256 and 2048 = 4 x 512
// by one Cooley-Tukey step -- see pfa_ct below.)
// Detector sizes det = p * M, p in {3, 5, 7}, M a power of two with a register
// engine (32 .. 512) -- 96, 160, 192, 224, 320, 384, 448, 640, 768, 896 ... -- by the
// prime-factor (Good-Thomas) decomposition: p and M are coprime, so with
//   input index   n = (M n1 + p n2) mod det          (n1 < p, n2 < M)
//   output index  k = (M qM k1 + p qp k2) mod det     (qM = M^-1 mod p,
//                                                      qp = p^-1 mod M)
// the length-det DFT is a p-point DFT over n1 of M-point DFTs over n2 WITHOUT
// twiddles between them.  In two dimensions a det x det tile is p x p
// sub-tiles of M x M: every sub-tile goes through the power-of-two register
// engine (tk_fft2, 2.4-3x the per-byte rate of the LDS line engine of
// fft_mixed.h), and the p x p combination is pointwise across the sub-tiles.
// The inverse takes its input in the forward's OUTPUT layout (the maps are
// symmetric), so nothing is permuted between the two transforms.
//
// The chunk body of _get_nearplane_gradients (reference
// ptycho/solvers/lstsq.py:422-579) on this decomposition -- three streaming
// kernels around two calls of the power-of-two transform:
//   tike_pfa_fwd_gather        patch x probe, zero-padded, written sub-tile
//                              by sub-tile (row y, column x of the tile go to
//                              sub-tile (y qM mod p, x qM mod p), element
//                              (y qp mod M, x qp mod M))
//   tk_fft2 (M, forward)       nscan * S * p^2 tiles
//   tike_pfa_combine_gradient  per (k2y, k2x): p x p DFT over the sub-tiles ->
//                              far plane F; intensity over the modes, cost,
//                              gradient factor (two sweeps over the modes);
//                              F x factor; inverse p x p DFT; in place
//   tk_fft2 (M, inverse)
//   tike_pfa_inv_products      chi read back through the same map: objproj,
//                              chi0, probe gradient (accumulated over a chunk
//                              of positions in LDS, one atomic per pixel, mode
//                              and chunk)
#include "fft_engine2.h"
#include "internal.h"
#include "tike_amd.h"

struct PfaGeom {
  int p, M, logM, det, qM, qp;  // qM = M^-1 mod p, qp = p^-1 mod M
};

// ... and, with the same kernels, powers of two beyond the fused kernels (1024
// = 4 x 256, 2048 = 4 x 512; 1024 as 2 x 512 was slower: 14.0 vs 15.5 k
// patterns/s at 2 modes) by one Cooley-Tukey step instead of the
// prime-factor one: input index n = n1 + p n2, output index k = k2 + M k1, and
// the twiddle w_det^(n1 k2) between the M-point and the p-point transforms
// (applied by the combine kernel, conjugated on the way back).  P in {2, 4}
// selects these maps at compile time (pfa_ct<P>).
template <int P>
constexpr bool pfa_ct = (P == 2 || P == 4);

static bool pfa_geom(int det, PfaGeom* g) {
  // (64 = 2 x 32 the same way: 813 k patterns/s at 8 modes against 1117 k on the
  // unfused kernels -- not taken)
  if (det == 1024 || det == 2048) {
    g->p = 4;
    g->M = det / 4;
    g->det = det;
    g->logM = det == 1024 ? 8 : 9;
    g->qM = g->qp = 1;  // (unused)
    return true;
  }
  for (int p : {3, 5, 7}) {
    if (det % p) continue;
    const int M = det / p;
    if (M < 32 || M > 512 || (M & (M - 1))) continue;
    g->p = p;
    g->M = M;
    g->det = det;
    g->logM = 0;
    while ((1 << g->logM) < M) ++g->logM;
    g->qM = g->qp = 0;
    for (int q = 1; q < p; ++q)
      if ((M * q) % p == 1) g->qM = q;
    for (int q = 1; q < M; ++q)
      if ((p * q) % M == 1) g->qp = q;
    return g->qM && g->qp;
  }
  return false;
}

// position of tile coordinate v (row or column) on the input side: sub-tile
// index n1 and element index n2
template <int P>
__device__ __forceinline__ void pfa_split(const PfaGeom& g, int v, int& n1, int& n2) {
  if (pfa_ct<P>) {
    n1 = v & (P - 1);
    n2 = v / P;
  } else {
    n1 = (v * g.qM) % P;
    n2 = (v * g.qp) & (g.M - 1);
  }
}
// ... and back: tile coordinate of (sub-tile n1, element n2)
template <int P>
__device__ __forceinline__ int pfa_coord(const PfaGeom& g, int n1, int n2) {
  if (pfa_ct<P>) return n1 + P * n2;
  int v = g.M * n1 + P * n2;  // < 2 det
  v -= v >= g.det ? g.det : 0;
  return v;
}
// frequency of (sub-tile index k1, element k2) on the output side: (M qM k1 +
// p qp k2) mod det = M (qM k1 mod p) + p (qp k2 mod M), minus det if that
// overflows (no integer division by a run-time value)
template <int P>
__device__ __forceinline__ int pfa_freq(const PfaGeom& g, int k1, int k2) {
  if (pfa_ct<P>) return k2 + g.M * k1;
  int k = g.M * ((g.qM * k1) % P) + P * ((g.qp * k2) & (g.M - 1));
  k -= k >= g.det ? g.det : 0;
  return k;
}

// ---- the probe of row y, hoisted: the shared probe rows P_s[y][:] (S x pw)
// and the eigen rows E_c,s[y][:] (C x Sm x pw) are position independent -- a
// work item that walks a chunk of positions loads them into LDS ONCE and a
// position contributes scalar weights only (probe.py:272-303; per position
// they cost as many L2 reads as the far plane itself).  Not for probes given
// one array per position (`pos_stride`, `unique`): those are read per use.
__device__ __forceinline__ bool pfa_hoistable(const TkProbe& p) {
  return p.pos_stride == 0 && p.unique == nullptr;
}
__device__ __forceinline__ int pfa_eigen_rows(const TkProbe& p) {
  return p.weights != nullptr && p.eigen != nullptr ? p.C * p.Sm : 0;
}
// every thread of the workgroup; the caller barriers before the first use
__device__ __forceinline__ void pfa_probe_rows(const TkProbe& p, int y, cf* prw, cf* erw) {
  const long pp = (long)p.pw * p.pw;
  for (int s = 0; s < p.S; ++s)
    for (int x = threadIdx.x; x < p.pw; x += 256)
      prw[s * p.pw + x] = p.probe[s * pp + (long)y * p.pw + x];
  const int ne = pfa_eigen_rows(p);
  for (int k = 0; k < ne; ++k)
    for (int x = threadIdx.x; x < p.pw; x += 256)
      erw[k * p.pw + x] = p.eigen[k * pp + (long)y * p.pw + x];
}
__device__ __forceinline__ cf pfa_probe_at(const TkProbe& p, const cf* prw, const cf* erw,
                                           const float* wn, int s, int x) {
  cf v = prw[s * p.pw + x];
  if (wn != nullptr) {
    v = v * wn[s];
    if (p.eigen != nullptr && s < p.Sm)
      for (int c = 0; c < p.C; ++c) {
        const cf e = erw[(c * p.Sm + s) * p.pw + x];
        const float wc = wn[(c + 1) * p.S + s];
        v.x += wc * e.x;
        v.y += wc * e.y;
      }
  }
  return v;
}

// tk_gather (common.h) in two halves: the four taps requested now, weighted
// later -- a load issued behind a store waits for it (one in-order counter), so
// a kernel that walks positions requests the taps of position n + 1 BEFORE it
// stores the products of position n.
struct PfaTaps {
  cf a, b, d, e;
  bool t1, t2, t3;  // taps inside the allocation
};
__device__ __forceinline__ PfaTaps pfa_taps(const cf* __restrict__ img, long ii, int W,
                                            long total) {
  const long last = total - 1;
  PfaTaps t;
  t.t1 = ii + 1 <= last;
  t.t2 = ii + W <= last;
  t.t3 = ii + W + 1 <= last;
  t.a = img[ii];
  t.b = img[t.t1 ? ii + 1 : last];
  t.d = img[t.t2 ? ii + W : last];
  t.e = img[t.t3 ? ii + W + 1 : last];
  return t;
}
__device__ __forceinline__ cf pfa_taps_value(const PfaTaps& t, const TkCorner& c) {
  const float w1 = t.t1 ? c.w01 : 0.0f, w2 = t.t2 ? c.w10 : 0.0f, w3 = t.t3 ? c.w11 : 0.0f;
  cf r = mk(t.a.x * c.w00, t.a.y * c.w00);
  r.x += t.b.x * w1;
  r.y += t.b.y * w1;
  r.x += t.d.x * w2;
  r.y += t.d.y * w2;
  r.x += t.e.x * w3;
  r.y += t.e.y * w3;
  return r;
}

// ------------------------------------------------------------- forward gather
// work item = (tile row y, chunk of positions): the probe rows once (above),
// then per position -- phase 1, lanes along x: the patch row (-> patches) into
// LDS; phase 2, lanes along the sub-tile layout (sub-tile n1x, element n2x):
// patch x probe read from LDS at x = (M n1x + p n2x) mod det, written as
// contiguous M-element segments.  A (nscan, S, p, p, M, M).
template <int P>
__global__ __launch_bounds__(256) void pfa_fwd_gather_kernel(
    const cf* __restrict__ psi, const float* __restrict__ scan, const TkProbe probe,
    cf* __restrict__ A, cf* __restrict__ patches, PfaGeom g, int nscan, int S, int pw, int H,
    int W, int chunk) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  cf* orow = reinterpret_cast<cf*>(lds_raw);  // pw: the patch row
  cf* prw = orow + pw;                        // S x pw
  cf* erw = prw + S * pw;                     // C x Sm x pw
  const int det = g.det, pad = (det - pw) / 2, M = g.M;
  const long total = (long)H * W;
  const long MM = (long)M * M, tile = (long)det * det;
  const bool hoist = pfa_hoistable(probe);
  const int nchunk = (nscan + chunk - 1) / chunk;
  const unsigned nitem = (unsigned)det * (unsigned)nchunk;
  for (unsigned item = blockIdx.x; item < nitem; item += gridDim.x) {
    const long c = item / (unsigned)det;
    const int y = (int)(item - (unsigned)c * (unsigned)det);
    const long n0 = c * chunk, n1 = n0 + chunk < nscan ? n0 + chunk : nscan;
    const int py = y - pad;
    const bool row_in = py >= 0 && py < pw;
    int n1y, n2y;
    pfa_split<P>(g, y, n1y, n2y);
    const long rowoff = ((long)n1y * P * M + n2y) * M;
    if (!row_in) {  // a row of the zero padding (uniform): zeros for every position
      for (long n = n0; n < n1; ++n)
        for (int sm = 0; sm < S; ++sm) {
          cf* dst = A + (n * S + sm) * tile + rowoff;
          for (int t = threadIdx.x; t < det; t += 256)
            dst[(long)(t >> g.logM) * MM + (t & (M - 1))] = mk(0.f, 0.f);
        }
      continue;
    }
    if (hoist) pfa_probe_rows(probe, py, prw, erw);
    // the taps of the first two pixels of a thread (pw <= 512: all of them)
    // travel one position ahead
    PfaTaps nt[2];
    bool nok[2];
    auto request = [&](long n) {
      const TkCorner cn = tk_corner(scan, n);
      const int iy = cn.sy + py;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int px = threadIdx.x + j * 256;
        const int ix = cn.sx + px;
        nok[j] = px < pw && iy >= 0 && iy < H && ix >= 0 && ix < W;
        // (outside: pixel 0 requested and selected away, no branch around loads)
        nt[j] = pfa_taps(psi, nok[j] ? (long)iy * W + ix : 0L, W, total);
      }
    };
    request(n0);
    for (long n = n0; n < n1; ++n) {
      const TkCorner cn = tk_corner(scan, n);
      const int iy = cn.sy + py;
      const bool row_ok = iy >= 0 && iy < H;
      cf held[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int px = threadIdx.x + j * 256;
        held[j] = nok[j] ? pfa_taps_value(nt[j], cn) : mk(0.f, 0.f);
        if (px < pw) orow[px] = held[j];
      }
      for (int px = threadIdx.x + 512; px < pw; px += 256) {  // (windows wider than 512)
        const int ix = cn.sx + px;
        const bool ok = row_ok && ix >= 0 && ix < W;
        const cf gth = tk_gather(psi, ok ? (long)iy * W + ix : 0L, W, total, cn);
        const cf o = ok ? gth : mk(0.f, 0.f);
        if (patches) patches[(n * pw + py) * pw + px] = o;
        orow[px] = o;
      }
      __syncthreads();
      if (n + 1 < n1) request(n + 1);  // in flight before the stores below
      if (patches) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int px = threadIdx.x + j * 256;
          if (px < pw) patches[(n * pw + py) * pw + px] = held[j];
        }
      }
      const float* wn =
          probe.weights ? probe.weights + n * (long)(probe.C + 1) * probe.S : nullptr;
      cf* out = A + n * S * tile + rowoff;
      // sub-tile order: slot t = n1x * M + n2x (segments of M contiguous elements)
      for (int t = threadIdx.x; t < det; t += 256) {
        const int n1x = t >> g.logM, n2x = t & (M - 1);
        const int x = pfa_coord<P>(g, n1x, n2x);
        const int px = x - pad;
        const bool in = px >= 0 && px < pw;
        const int pxc = in ? px : 0;
        cf* dst = out + (long)n1x * MM + n2x;
        const cf o = orow[pxc];
        for (int sm = 0; sm < S; ++sm) {
          const cf w = hoist ? pfa_probe_at(probe, prw, erw, wn, sm, pxc)
                             : probe.at(n, sm, (long)py * pw + pxc);
          dst[(long)sm * tile] = in ? o * w : mk(0.f, 0.f);
        }
      }
      __syncthreads();
    }
  }
}

// ------------------------------------- M = 128: gather + sub-tile transform
// A 128 x 128 sub-tile is 128 KiB: it fits the LDS of a CU, so the gather and
// the whole 2-D transform of a sub-tile are ONE kernel whose only HBM traffic
// is the transformed sub-tile, written once from registers (the forward
// operator's fwd128_lds_kernel with the prime-factor map in front: 384 = 3 x
// 128, 640, 896).  tike_pfa_fwd_gather + tk_fft2 move W + R + W of the far
// plane for the same result.
// Work item = (position, sub-tile (n1y, n1x)); 1024 threads: thread (line, j)
// owns elements n2x = j + 8 i of sub-tile row n2y = line, i.e. pixel
//   y = (M n1y + p line) mod det,  x = (M n1x + p (j + 8 i)) mod det
// of the zero-padded tile; the patch values are gathered once and shared by
// the modes.  The probe is read from `psub`, the shared probe (and the eigen
// probes) PERMUTED into the sub-tile layout once per launch
// (pfa_permute_probe_kernel): consecutive lanes, consecutive addresses --
// straight from the probe they would be p pixels apart.
template <int P>
__global__ __launch_bounds__(256) void pfa_permute_probe_kernel(const cf* __restrict__ src,
                                                                cf* __restrict__ dst, PfaGeom g,
                                                                int nimg, int pw) {
  const int det = g.det, M = g.M, pad = (det - pw) / 2;
  const long tile = (long)det * det;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nimg * tile; i += gridDim.x * 256L) {
    const long img = i / tile;
    const int e = (int)(i - img * tile);
    // e = ((n1y P + n1x) M + n2y) M + n2x
    const int n2x = e & (M - 1), n2y = (e >> g.logM) & (M - 1);
    const int st = e >> (2 * g.logM), n1x = st % P, n1y = st / P;
    int y = M * n1y + P * n2y, x = M * n1x + P * n2x;
    y -= y >= det ? det : 0;
    x -= x >= det ? det : 0;
    const int py = y - pad, px = x - pad;
    const bool in = py >= 0 && py < pw && px >= 0 && px < pw;
    dst[i] = in ? src[img * (long)pw * pw + (long)py * pw + px] : mk(0.f, 0.f);
  }
}

template <int P>
__global__ __launch_bounds__(1024, 4) void pfa_fwd128_kernel(
    const cf* __restrict__ psi, const float* __restrict__ scan, const TkProbe probe,
    const cf* __restrict__ psub, cf* __restrict__ B, cf* __restrict__ patches, PfaGeom g,
    int nscan, int S, int pw, int H, int W, const cf* __restrict__ twtab) {
  constexpr int N = 128, T = 8, LS = TK_L128_LS, det = P * N;
  __shared__ cf lds[N * LS + FftTwLds<N>::ELEMS];
  cf* twl = lds + N * LS;
  FftTwLds<N>::fill(twl, twtab);
  __syncthreads();
  const int tid = threadIdx.x;
  int line = tid / T, j = tid % T;
  asm volatile("" : "+v"(line), "+v"(j));
  const FftTwLds<N> tw{twl, j};
  __builtin_assume(j >= 0 && j < T && line >= 0 && line < N);
  const long MM = (long)N * N, tile = (long)det * det;
  const long total = (long)H * W;
  const int pad = (det - pw) / 2;
  const int nE = probe.weights != nullptr && probe.eigen != nullptr ? probe.C : 0;
  auto at = [](const cf* base, unsigned byte_off) -> const cf* {
    return reinterpret_cast<const cf*>(reinterpret_cast<const char*>(base) + byte_off);
  };
  const unsigned pbo = (unsigned)(line * N + j) * (unsigned)sizeof(cf);
  for (long item = blockIdx.x; item < (long)nscan * (P * P); item += gridDim.x) {
    const long n = item / (P * P);
    const int st = (int)(item - n * (P * P));  // n1y P + n1x
    const int n1y = st / P, n1x = st - n1y * P;
    const TkCorner c = tk_corner(scan, n);
    int y = N * n1y + P * line;
    y -= y >= det ? det : 0;
    const int py = y - pad;
    const bool row_in = py >= 0 && py < pw;
    const int x0 = N * n1x + P * j;  // + 8 P i, minus det once past the edge
    const bool interior = pad == 0 && c.sy >= 0 && c.sx >= 0 && c.sy + pw < H &&
                          c.sx + pw < W && total < (1L << 28);
    cf pv[16];
    if (interior) {
      // the two taps of a row are adjacent complex values: one 16-byte load
      typedef float tk_v4f __attribute__((ext_vector_type(4)));
      auto ld4 = [](const cf* base, unsigned byte_off) {
        tk_v4f v;
        __builtin_memcpy(&v, reinterpret_cast<const char*>(base) + byte_off, sizeof(v));
        return v;
      };
      const unsigned g0 = (unsigned)((c.sy + y) * W + c.sx) * (unsigned)sizeof(cf);
      const unsigned g1 = g0 + (unsigned)W * (unsigned)sizeof(cf);
#pragma unroll
      for (int h = 0; h < 16; h += 4) {
        tk_v4f u[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int x = x0 + 8 * P * (h + i);
          x -= x >= det ? det : 0;
          u[i] = ld4(psi, g0 + (unsigned)x * 8u);
          l[i] = ld4(psi, g1 + (unsigned)x * 8u);
        }
        asm volatile(""
                     : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(l[0]), "+v"(l[1]),
                       "+v"(l[2]), "+v"(l[3]));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          cf o = mk(u[i].x * c.w00, u[i].y * c.w00);
          o.x += u[i].z * c.w01;
          o.y += u[i].w * c.w01;
          o.x += l[i].x * c.w10;
          o.y += l[i].y * c.w10;
          o.x += l[i].z * c.w11;
          o.y += l[i].w * c.w11;
          pv[h + i] = o;
        }
      }
    } else {
      const int iy = c.sy + py;
      const bool row_ok = row_in && iy >= 0 && iy < H;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int x = x0 + 8 * P * i;
        x -= x >= det ? det : 0;
        const int px = x - pad, ix = c.sx + px;
        const bool ok = row_ok && px >= 0 && px < pw && ix >= 0 && ix < W;
        const cf o = tk_gather(psi, ok ? (long)iy * W + ix : 0L, W, total, c);
        pv[i] = ok ? o : mk(0.f, 0.f);
        __builtin_amdgcn_sched_barrier(0);  // rare path: one element in flight
      }
    }
    if (patches != nullptr && row_in) {
      cf* __restrict__ On = patches + ((long)n * pw + py) * pw;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int x = x0 + 8 * P * i;
        x -= x >= det ? det : 0;
        const int px = x - pad;
        if (px >= 0 && px < pw) On[px] = pv[i];
      }
    }
    const float* __restrict__ wn =
        probe.weights ? probe.weights + n * (long)(probe.C + 1) * probe.S : nullptr;
    cf* __restrict__ out = B + (long)n * S * tile + (long)st * MM;
    for (int s = 0; s < S; ++s) {
      // probe of (position, mode) in the sub-tile layout (zero outside the window)
      const cf* __restrict__ Pn = psub + (long)s * tile + (long)st * MM;
      const float w0 = wn != nullptr ? wn[s] : 1.0f;
      cf v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = *at(Pn, pbo + 64 * i) * w0;
      if (s < probe.Sm) {
        for (int k = 0; k < nE; ++k) {  // uniform, rare (modes owning eigen probes)
          const cf* __restrict__ E =
              psub + ((long)probe.S + (long)k * probe.Sm + s) * tile + (long)st * MM;
          const float wk = wn[(k + 1) * probe.S + s];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const cf e = *at(E, pbo + 64 * i);
            v[i].x += wk * e.x;
            v[i].y += wk * e.y;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = pv[i] * v[i];
      cf* lbase = lds + line * LS;
      FftStageWave<N, false, 0>::run(v, lbase, j, tw);
      const int rsw = tk_l128_swizzle(line);
#pragma unroll
      for (int i = 0; i < 16; ++i) lbase[(j + i * T + rsw) & (N - 1)] = v[i];
      __syncthreads();
      const int col = line;
      auto cat = [&](int e) { return e * LS + ((col + tk_l128_swizzle(e)) & (N - 1)); };
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = lds[cat(j + i * T)];
      FftStageWave<N, false, 0>::run_at(v, lds, j, tw, cat);
      // (stored straight from the column layout: eight 64-byte pieces of eight
      // rows per wave and instruction.  Back through the tile and out row by
      // row, 512 contiguous bytes per instruction, was SLOWER: 3.22 vs 2.79 ms
      // per 1000 positions x 4 modes -- the kernel is bound by its LDS traffic)
      cf* __restrict__ dst = out + (long)s * tile + col;
#pragma unroll
      for (int i = 0; i < 16; ++i) tk_st_stream(dst + (long)(j + i * T) * N, v[i]);
      __syncthreads();  // the next mode's rows overwrite the tile
    }
  }
}

// ------------------------------------------------ p x p combine + gradient
// 1-D DFT over the sub-tile index, in place on p values
template <int P, bool INV>
__device__ __forceinline__ void pfa_dft2(cf (&v)[P][P]) {
#pragma unroll
  for (int a = 0; a < P; ++a) Dft<P, INV>::run(v[a]);  // along x
#pragma unroll
  for (int b = 0; b < P; ++b) {
    cf t[P];
#pragma unroll
    for (int a = 0; a < P; ++a) t[a] = v[a][b];
    Dft<P, INV>::run(t);
#pragma unroll
    for (int a = 0; a < P; ++a) v[a][b] = t[a];
  }
}

// thread = one (k2y, k2x) of one position: the P x P sub-tile values of a mode
// in registers.  B (nscan, S, p, p, M, M), in place.
// SMAX > 0: at most SMAX modes, all of them kept in registers between the
// intensity and the factor (one read of the sub-tiles); SMAX = 0: any number of
// modes, formed twice (the second read comes from L2).
template <int P, int MODEL, int SMAX = 0>
__global__ __launch_bounds__(256) void pfa_combine_gradient_kernel(
    cf* __restrict__ B, const float* __restrict__ data, const unsigned char* __restrict__ mask,
    const TkCostSink costs, PfaGeom g, int nscan, int S, float fwd_scale,
    float unmeasured_scaling, float inv_nmeasured, int grad,
    const cf* __restrict__ twdet) {
  __shared__ float red[4];
  const int M = g.M, det = g.det;
  const long MM = (long)M * M, tile = (long)det * det;
  const int blocks_per = (int)(MM / 256);  // M >= 32: a multiple of 256 pixels
  const unsigned nitem = (unsigned)nscan * (unsigned)blocks_per;
  const float s2 = fwd_scale * fwd_scale;
  for (unsigned item = blockIdx.x; item < nitem; item += gridDim.x) {
    const unsigned n = item / (unsigned)blocks_per;
    const int blk = (int)(item - n * (unsigned)blocks_per);
    const int e = blk * 256 + threadIdx.x;  // k2y * M + k2x
    const int k2y = e >> g.logM, k2x = e & (M - 1);
    cf* base = B + (long)n * S * tile + e;
    // counts and mask bits of the P x P pixels this thread owns, requested
    // with everything else (NaN at unmeasured pixels is selected away)
    float d[P][P];
    bool ms[P][P];
#pragma unroll
    for (int a = 0; a < P; ++a) {
      const int ky = pfa_freq<P>(g, a, k2y);
#pragma unroll
      for (int b = 0; b < P; ++b) {
        const int kx = pfa_freq<P>(g, b, k2x);
        const long pix = (long)ky * det + kx;
        d[a][b] = data[(long)n * tile + pix];
        ms[a][b] = mask ? mask[pix] != 0 : true;
      }
    }
    // Cooley-Tukey step (P = 2, 4): w_det^(n1y k2y + n1x k2x) on the sub-tile
    // (n1y, n1x) in front of the p x p DFT, its conjugate behind the inverse
    cf wy[pfa_ct<P> ? P : 1], wx[pfa_ct<P> ? P : 1];
    if (pfa_ct<P>) {
#pragma unroll
      for (int a = 0; a < P; ++a) {
        wy[a] = twdet[a * k2y];
        wx[a] = twdet[a * k2x];
      }
    }
    auto twiddle = [&](cf (&v)[P][P], auto inv) {
      if (pfa_ct<P>) {
#pragma unroll
        for (int a = 0; a < P; ++a)
#pragma unroll
          for (int b = 0; b < P; ++b) {
            const cf w = wy[a] * wx[b];
            v[a][b] = v[a][b] * (decltype(inv)::value ? conjf(w) : w);
          }
      }
    };
    float I[P][P];
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
      for (int b = 0; b < P; ++b) I[a][b] = 0.f;
    cf keep[SMAX > 0 ? SMAX : 1][P][P];
    if (SMAX > 0) {
#pragma unroll
      for (int s = 0; s < SMAX; ++s) {
        if (s < S) {  // uniform
#pragma unroll
          for (int a = 0; a < P; ++a)
#pragma unroll
            for (int b = 0; b < P; ++b) keep[s][a][b] = base[s * tile + (a * P + b) * MM];
          twiddle(keep[s], std::false_type{});
          pfa_dft2<P, false>(keep[s]);
#pragma unroll
          for (int a = 0; a < P; ++a)
#pragma unroll
            for (int b = 0; b < P; ++b) I[a][b] += norm2(keep[s][a][b]) * s2;
        }
      }
    } else {
      for (int s = 0; s < S; ++s) {
        cf v[P][P];
#pragma unroll
        for (int a = 0; a < P; ++a)
#pragma unroll
          for (int b = 0; b < P; ++b) v[a][b] = base[s * tile + (a * P + b) * MM];
        twiddle(v, std::false_type{});
        pfa_dft2<P, false>(v);
#pragma unroll
        for (int a = 0; a < P; ++a)
#pragma unroll
          for (int b = 0; b < P; ++b) I[a][b] += norm2(v[a][b]) * s2;
      }
    }
    float cost = 0.f;
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
      for (int b = 0; b < P; ++b) {
        float gg, term;
        if (MODEL == 0) {
          const float sI = sqrtf(I[a][b]), sd = sqrtf(d[a][b]);
          const float diff = sI - sd;
          term = diff * diff;
          gg = -(1.0f - sd / (sI + 1e-9f));
        } else {
          term = I[a][b] - d[a][b] * logf(I[a][b] + 1e-9f);
          gg = -(1.0f - d[a][b] / (I[a][b] + 1e-9f));
        }
        cost += ms[a][b] ? term : 0.f;
        I[a][b] = (ms[a][b] ? gg : unmeasured_scaling - 1.0f) * fwd_scale;
      }
    if (costs.costs) {
      cost = tk_block_sum256(cost, red);
      if (threadIdx.x == 0) tk_cost_add(costs, n, blk, cost * inv_nmeasured);
    }
    if (!grad) continue;
    // second sweep: F x factor, back through the p x p DFT (the far plane of
    // a mode is formed twice; its second read comes from L2)
    if (SMAX > 0) {
#pragma unroll
      for (int s = 0; s < SMAX; ++s) {
        if (s < S) {
#pragma unroll
          for (int a = 0; a < P; ++a)
#pragma unroll
            for (int b = 0; b < P; ++b) keep[s][a][b] = keep[s][a][b] * I[a][b];
          pfa_dft2<P, true>(keep[s]);
          twiddle(keep[s], std::true_type{});
#pragma unroll
          for (int a = 0; a < P; ++a)
#pragma unroll
            for (int b = 0; b < P; ++b) base[s * tile + (a * P + b) * MM] = keep[s][a][b];
        }
      }
      continue;
    }
    for (int s = 0; s < S; ++s) {
      cf v[P][P];
#pragma unroll
      for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) v[a][b] = base[s * tile + (a * P + b) * MM];
      twiddle(v, std::false_type{});
      pfa_dft2<P, false>(v);
#pragma unroll
      for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) v[a][b] = v[a][b] * I[a][b];
      pfa_dft2<P, true>(v);
      twiddle(v, std::true_type{});
#pragma unroll
      for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) base[s * tile + (a * P + b) * MM] = v[a][b];
    }
  }
}

// ------------------------------------------------ inverse: crop + products
// work item = (probe row y, chunk of positions); lanes along x.  C is the
// inverse transform's output in the sub-tile layout; chi = inv_scale * C at
// the mapped place of (pad + y, pad + x), read straight through the map (64
// lanes touch ~24 cache lines of the three M-element segments of the row;
// staging the row through LDS -- contiguous reads, two barriers per position
// -- was slower: 0.87 vs 0.66 ms per 200 positions x 4 modes at 384^2).
template <int P>
__global__ __launch_bounds__(256) void pfa_inv_products_kernel(
    const cf* __restrict__ C, const cf* __restrict__ patches, const TkProbe probe,
    cf* __restrict__ objproj, cf* __restrict__ chi0, float* __restrict__ mpu, float mpu_scale,
    float* __restrict__ part, PfaGeom g, int nscan, int S, int pw, int chunk, float inv_scale) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  cf* acc = reinterpret_cast<cf*>(lds_raw);  // S x pw
  cf* prw = acc + S * pw;                    // S x pw
  cf* erw = prw + S * pw;                    // C x Sm x pw
  const int det = g.det, pad = (det - pw) / 2, M = g.M;
  const long MM = (long)M * M, tile = (long)det * det;
  const int nchunk = (nscan + chunk - 1) / chunk;
  const unsigned nitem = (unsigned)pw * (unsigned)nchunk;
  const bool grad = mpu != nullptr || part != nullptr;
  const bool hoist = pfa_hoistable(probe);
  for (unsigned item = blockIdx.x; item < nitem; item += gridDim.x) {
    const long c = item / (unsigned)pw;
    const int y = (int)(item - (unsigned)c * (unsigned)pw);
    const long n0 = c * chunk, n1 = n0 + chunk < nscan ? n0 + chunk : nscan;
    int n1y, n2y;
    pfa_split<P>(g, pad + y, n1y, n2y);
    const long rowoff = ((long)n1y * P * M + n2y) * M;
    if (grad)
      for (int idx = threadIdx.x; idx < S * pw; idx += 256) acc[idx] = mk(0.f, 0.f);
    if (hoist) pfa_probe_rows(probe, y, prw, erw);
    __syncthreads();  // (zeroed / loaded by slot, used by column)
    // the work of pixel x of position n on values already in registers
    auto consume = [&](long n, const float* wn, int x, const cf* v4, int s0, cf O, cf& op) {
      const long rowpix = ((long)n * pw + y) * pw;
      const long pix = (long)y * pw + x;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int sm = s0 + u;
        if (sm < S) {
          const cf w = hoist ? pfa_probe_at(probe, prw, erw, wn, sm, x) : probe.at(n, sm, pix);
          const cf chi = v4[u] * inv_scale;
          op = op + conjf(w) * chi;
          if (grad) acc[sm * pw + x] = acc[sm * pw + x] + O * chi;
          if (sm == 0 && chi0) chi0[rowpix + x] = chi;
        }
      }
    };
    if (S <= 4 && pw <= 512) {
      // at most four modes, two pixels per thread: the values of position
      // n + 1 are requested BEFORE the stores of position n (a load issued
      // behind a store waits for it)
      cf nv[2][4], nO[2];
      auto request = [&](long n) {
        const cf* src = C + n * S * tile + rowoff;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int x = threadIdx.x + j * 256, xc = x < pw ? x : 0;
          int n1x, n2x;
          pfa_split<P>(g, pad + xc, n1x, n2x);
          const cf* q = src + (long)n1x * MM + n2x;
#pragma unroll
          for (int u = 0; u < 4; ++u) nv[j][u] = q[(long)(u < S ? u : S - 1) * tile];
          nO[j] = patches[((long)n * pw + y) * pw + xc];
        }
      };
      request(n0);
      for (long n = n0; n < n1; ++n) {
        const float* wn =
            probe.weights ? probe.weights + n * (long)(probe.C + 1) * probe.S : nullptr;
        cf cv[2][4], cO[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          cO[j] = conjf(nO[j]);
#pragma unroll
          for (int u = 0; u < 4; ++u) cv[j][u] = nv[j][u];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
          asm volatile("" : "+v"(cv[j][0].x), "+v"(cv[j][1].x), "+v"(cv[j][2].x),
                       "+v"(cv[j][3].x));
        if (n + 1 < n1) request(n + 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int x = threadIdx.x + j * 256;
          if (x < pw) {
            cf op = mk(0.f, 0.f);
            consume(n, wn, x, cv[j], 0, cO[j], op);
            if (objproj) objproj[((long)n * pw + y) * pw + x] = op;
          }
        }
      }
    } else {
      for (long n = n0; n < n1; ++n) {
        const float* wn =
            probe.weights ? probe.weights + n * (long)(probe.C + 1) * probe.S : nullptr;
        const cf* src = C + n * S * tile + rowoff;
        const long rowpix = ((long)n * pw + y) * pw;
        for (int x = threadIdx.x; x < pw; x += 256) {
          int n1x, n2x;
          pfa_split<P>(g, pad + x, n1x, n2x);
          const cf* q = src + (long)n1x * MM + n2x;
          const cf O = conjf(patches[rowpix + x]);
          cf op = mk(0.f, 0.f);
          for (int s0 = 0; s0 < S; s0 += 4) {
            cf v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = q[(long)(s0 + u < S ? s0 + u : S - 1) * tile];
            consume(n, wn, x, v, s0, O, op);
          }
          if (objproj) objproj[rowpix + x] = op;
        }
      }
    }
    if (grad) {
      for (int sm = 0; sm < S; ++sm) {
        const long o = 2 * (((long)sm * pw + y) * pw);
        float* dst = part ? part + c * 2L * S * pw * pw + o : mpu + o;
        for (int x = threadIdx.x; x < pw; x += 256) {
          const cf v = acc[sm * pw + x] * mpu_scale;
          if (part) {
            dst[2 * x] = v.x;
            dst[2 * x + 1] = v.y;
          } else {
            unsafeAtomicAdd(dst + 2 * x, v.x);
            unsafeAtomicAdd(dst + 2 * x + 1, v.y);
          }
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ entries
// dynamic LDS of the gather (`rows` = 1: the patch row) and of the products
// (`rows` = S: the accumulators): + the hoisted probe and eigen rows
static size_t pfa_lds(const TkProbe& p, int rows) {
  const int ne = p.weights != nullptr && p.eigen != nullptr ? p.C * p.Sm : 0;
  return sizeof(cf) * (size_t)p.pw * (size_t)(rows + p.S + ne);
}

extern "C" int tike_pfa_supported(int S, int pw, int det) {
  PfaGeom g;
  if (S < 1 || pw < 1 || det < pw || !pfa_geom(det, &g)) return 0;
  // (LDS of the products: accumulators + probe rows + up to 2 S eigen rows)
  return sizeof(cf) * (size_t)(4 * S) * pw <= 150 * 1024 ? 1 : 0;
}

extern "C" int tike_pfa_fwd_gather(const void* psi, const float* scan, const void* probe,
                                   int probe_per_scan, const void* unique,
                                   const void* eigen_probe, const float* eigen_weights,
                                   int num_eigen, int eigen_modes, void* subtiles,
                                   void* patches, int nscan, int S, int pw, int det, int H, int W,
                                   void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 1 && det >= pw && H >= 1 && W >= 1);
  TK_CHECK_ARG(!(eigen_weights && probe_per_scan));
  PfaGeom g;
  if (!tike_pfa_supported(S, pw, det) || !pfa_geom(det, &g)) return TK_ERR_UNSUPPORTED;
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(psi && scan && probe && subtiles);
  const TkProbe P = tk_make_probe(probe, probe_per_scan, eigen_probe, eigen_weights, num_eigen,
                                  eigen_modes, S, pw, unique);
  // (tile row, chunk of positions) work items, about eight per CU
  int nchunk = (2048 + det - 1) / det;
  if (nchunk > (nscan + 7) / 8) nchunk = (nscan + 7) / 8;
  if (nchunk < 1) nchunk = 1;
  const int chunk = (nscan + nchunk - 1) / nchunk;
  nchunk = (nscan + chunk - 1) / chunk;
  const dim3 grid(tk_grid((long)det * nchunk, 16)), block(256);
  const size_t lds = pfa_lds(P, 1);
  if (lds > 150 * 1024) return TK_ERR_UNSUPPORTED;
  if (lds > 48 * 1024) {
    hipError_t e3 = hipFuncSetAttribute((const void*)pfa_fwd_gather_kernel<3>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e5 = hipFuncSetAttribute((const void*)pfa_fwd_gather_kernel<5>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e7 = hipFuncSetAttribute((const void*)pfa_fwd_gather_kernel<7>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e7 == hipSuccess)
      e7 = hipFuncSetAttribute((const void*)pfa_fwd_gather_kernel<2>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e7 == hipSuccess)
      e7 = hipFuncSetAttribute((const void*)pfa_fwd_gather_kernel<4>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e3 != hipSuccess || e5 != hipSuccess || e7 != hipSuccess)
      return (int)(e3 != hipSuccess ? e3 : e5 != hipSuccess ? e5 : e7);
  }
  if (g.p == 2)
    hipLaunchKernelGGL(pfa_fwd_gather_kernel<2>, grid, block, lds, stream, (const cf*)psi, scan,
                       P, (cf*)subtiles, (cf*)patches, g, nscan, S, pw, H, W, chunk);
  else if (g.p == 4)
    hipLaunchKernelGGL(pfa_fwd_gather_kernel<4>, grid, block, lds, stream, (const cf*)psi, scan,
                       P, (cf*)subtiles, (cf*)patches, g, nscan, S, pw, H, W, chunk);
  else if (g.p == 3)
    hipLaunchKernelGGL(pfa_fwd_gather_kernel<3>, grid, block, lds, stream, (const cf*)psi, scan,
                       P, (cf*)subtiles, (cf*)patches, g, nscan, S, pw, H, W, chunk);
  else if (g.p == 5)
    hipLaunchKernelGGL(pfa_fwd_gather_kernel<5>, grid, block, lds, stream, (const cf*)psi, scan,
                       P, (cf*)subtiles, (cf*)patches, g, nscan, S, pw, H, W, chunk);
  else
    hipLaunchKernelGGL(pfa_fwd_gather_kernel<7>, grid, block, lds, stream, (const cf*)psi, scan,
                       P, (cf*)subtiles, (cf*)patches, g, nscan, S, pw, H, W, chunk);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_pfa_fft2(const void* in, void* out, long ntile, int det, int inverse,
                             void* stream_) {
  TK_ENTER();
  PfaGeom g;
  if (!pfa_geom(det, &g)) return TK_ERR_UNSUPPORTED;
  TK_CHECK_ARG(in != out);
  return tk_fft2((const cf*)in, (cf*)out, ntile * g.p * g.p, g.M, inverse, 1.0f,
                 (hipStream_t)stream_);
}

// gather + forward sub-tile transforms in one launch (M = 128: det = 384, 640,
// 896; a shared probe): subtiles receives what tike_pfa_fwd_gather followed by
// tike_pfa_fft2 leaves there.  probe_scratch ((S + C Sm) det^2 c64): the probe
// and the eigen probes in the sub-tile layout, rewritten by every call.
extern "C" int tike_pfa_fwd_subtiles_supported(int S, int pw, int det) {
  PfaGeom g;
  return tike_pfa_supported(S, pw, det) && pfa_geom(det, &g) && g.M == 128 ? 1 : 0;
}

extern "C" int tike_pfa_fwd_subtiles(const void* psi, const float* scan, const void* probe,
                                     const void* eigen_probe, const float* eigen_weights,
                                     int num_eigen, int eigen_modes, void* probe_scratch,
                                     void* subtiles, void* patches, int nscan, int S, int pw,
                                     int det, int H, int W, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 1 && det >= pw && H >= 1 && W >= 1);
  PfaGeom g;
  if (!tike_pfa_fwd_subtiles_supported(S, pw, det) || !pfa_geom(det, &g))
    return TK_ERR_UNSUPPORTED;
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(psi && scan && probe && probe_scratch && subtiles);
  const cf* tw = tk_twiddles();
  if (!tw) return (int)hipErrorNotInitialized;
  const TkProbe Pr = tk_make_probe(probe, 0, eigen_probe, eigen_weights, num_eigen, eigen_modes,
                                   S, pw, nullptr);
  const int ne = eigen_weights && eigen_probe ? num_eigen * eigen_modes : 0;
  const long tile = (long)det * det;
  cf* psub = (cf*)probe_scratch;
#define TK_PFW(P_)                                                                             \
  do {                                                                                         \
    hipLaunchKernelGGL(pfa_permute_probe_kernel<P_>, dim3(tk_grid((S * tile + 255) / 256, 8)), \
                       dim3(256), 0, stream, (const cf*)probe, psub, g, S, pw);                \
    if (ne > 0)                                                                                \
      hipLaunchKernelGGL(pfa_permute_probe_kernel<P_>,                                         \
                         dim3(tk_grid((ne * tile + 255) / 256, 8)), dim3(256), 0, stream,      \
                         (const cf*)eigen_probe, psub + S * tile, g, ne, pw);                  \
    hipLaunchKernelGGL(pfa_fwd128_kernel<P_>, dim3(tk_grid((long)nscan * P_ * P_, 1)),         \
                       dim3(1024), 0, stream, (const cf*)psi, scan, Pr, (const cf*)psub,       \
                       (cf*)subtiles, (cf*)patches, g, nscan, S, pw, H, W, tw);                \
  } while (0)
  if (g.p == 3)
    TK_PFW(3);
  else if (g.p == 5)
    TK_PFW(5);
  else
    TK_PFW(7);
#undef TK_PFW
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_pfa_combine_gradient(void* subtiles, const float* data,
                                         const unsigned char* measured, float* costs,
                                         int nscan, int S, int det, float fwd_scale, int model,
                                         float unmeasured_scaling, long num_measured,
                                         int apply_gradient, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && (model == 0 || model == 1) && num_measured > 0);
  PfaGeom g;
  if (!pfa_geom(det, &g)) return TK_ERR_UNSUPPORTED;
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(subtiles && data);
  const int blocks_per = g.M * g.M / 256;
  if ((long)nscan * blocks_per >= (1L << 31)) return TK_ERR_ARG;
  TkCostSink sink;
  int rc = tk_cost_sink(costs, nscan, blocks_per, stream, &sink);
  if (rc) return rc;
  // one work item per workgroup: the loads of a second item would queue
  // behind the stores of the first (one in-order counter)
  const dim3 grid((unsigned)((long)nscan * blocks_per)), block(256);
  const float inv = 1.0f / (float)num_measured;
  // (the Cooley-Tukey sizes: w_det^m, m < det)
  const cf* twdet = nullptr;
  if (g.p == 2 || g.p == 4) {
    const cf* tw = tk_twiddles();
    if (!tw) return (int)hipErrorNotInitialized;
    twdet = tw + det;
  }
#define TK_PFA_CG(PP, MM, SM)                                                                \
  hipLaunchKernelGGL((pfa_combine_gradient_kernel<PP, MM, SM>), grid, block, 0, stream,         \
                     (cf*)subtiles, data, measured, sink, g, nscan, S, fwd_scale,               \
                     unmeasured_scaling, inv, apply_gradient, twdet)
  // (3 x 3 sub-tiles of up to 4 modes fit the registers: one read)
  const bool resident = g.p == 3 && S <= 4 && apply_gradient;
  if (resident && model == 0) TK_PFA_CG(3, 0, 4);
  if (resident && model == 1) TK_PFA_CG(3, 1, 4);
  if (!resident && g.p == 3 && model == 0) TK_PFA_CG(3, 0, 0);
  if (!resident && g.p == 3 && model == 1) TK_PFA_CG(3, 1, 0);
  if (g.p == 5 && model == 0) TK_PFA_CG(5, 0, 0);
  if (g.p == 5 && model == 1) TK_PFA_CG(5, 1, 0);
  if (g.p == 7 && model == 0) TK_PFA_CG(7, 0, 0);
  if (g.p == 7 && model == 1) TK_PFA_CG(7, 1, 0);
  if (g.p == 2 && model == 0) TK_PFA_CG(2, 0, 0);
  if (g.p == 2 && model == 1) TK_PFA_CG(2, 1, 0);
  if (g.p == 4 && model == 0) TK_PFA_CG(4, 0, 0);
  if (g.p == 4 && model == 1) TK_PFA_CG(4, 1, 0);
#undef TK_PFA_CG
  TK_LAUNCH_CHECK();
  return tk_cost_finish(sink, nscan, stream);
}

extern "C" int tike_pfa_inv_products(const void* subtiles, const void* patches,
                                     const void* probe, int probe_per_scan, const void* unique,
                                     const void* eigen_probe, const float* eigen_weights,
                                     int num_eigen, int eigen_modes, void* objproj, void* chi0,
                                     void* m_probe_update, float probe_update_scale, int nscan,
                                     int S, int pw, int det, float inv_scale, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 1 && det >= pw);
  TK_CHECK_ARG(!(eigen_weights && probe_per_scan));
  PfaGeom g;
  if (!tike_pfa_supported(S, pw, det) || !pfa_geom(det, &g)) return TK_ERR_UNSUPPORTED;
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(subtiles && patches && probe);
  const TkProbe P = tk_make_probe(probe, probe_per_scan, eigen_probe, eigen_weights, num_eigen,
                                  eigen_modes, S, pw, unique);
  int nchunk = (int)((2048 + pw - 1) / pw);
  if (nchunk > (nscan + 7) / 8) nchunk = (nscan + 7) / 8;
  if (nchunk < 1) nchunk = 1;
  int chunk = (nscan + nchunk - 1) / nchunk;
  nchunk = (nscan + chunk - 1) / chunk;
  float* part = nullptr;
  const long nmpu = 2L * S * pw * pw;
  if (m_probe_update && tk_deterministic()) {
    part = tk_det_scratch(sizeof(float) * (size_t)nmpu * nchunk);
    if (!part) {  // scratch too small: one chunk, one contributor per address
      nchunk = 1;
      chunk = nscan;
      part = tk_det_scratch(sizeof(float) * (size_t)nmpu);
      if (!part) return TK_ERR_ARG;
    }
  }
  const size_t lds = pfa_lds(P, S);
  if (lds > 150 * 1024) return TK_ERR_UNSUPPORTED;
  if (lds > 48 * 1024) {
    hipError_t e3 = hipFuncSetAttribute((const void*)pfa_inv_products_kernel<3>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e5 = hipFuncSetAttribute((const void*)pfa_inv_products_kernel<5>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e7 = hipFuncSetAttribute((const void*)pfa_inv_products_kernel<7>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e7 == hipSuccess)
      e7 = hipFuncSetAttribute((const void*)pfa_inv_products_kernel<2>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e7 == hipSuccess)
      e7 = hipFuncSetAttribute((const void*)pfa_inv_products_kernel<4>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e3 != hipSuccess || e5 != hipSuccess || e7 != hipSuccess)
      return (int)(e3 != hipSuccess ? e3 : e5 != hipSuccess ? e5 : e7);
  }
  const dim3 grid(tk_grid((long)pw * nchunk, 16)), block(256);
  if (g.p == 2)
    hipLaunchKernelGGL(pfa_inv_products_kernel<2>, grid, block, lds, stream, (const cf*)subtiles,
                       (const cf*)patches, P, (cf*)objproj, (cf*)chi0, (float*)m_probe_update,
                       probe_update_scale, part, g, nscan, S, pw, chunk, inv_scale);
  else if (g.p == 4)
    hipLaunchKernelGGL(pfa_inv_products_kernel<4>, grid, block, lds, stream, (const cf*)subtiles,
                       (const cf*)patches, P, (cf*)objproj, (cf*)chi0, (float*)m_probe_update,
                       probe_update_scale, part, g, nscan, S, pw, chunk, inv_scale);
  else if (g.p == 3)
    hipLaunchKernelGGL(pfa_inv_products_kernel<3>, grid, block, lds, stream, (const cf*)subtiles,
                       (const cf*)patches, P, (cf*)objproj, (cf*)chi0, (float*)m_probe_update,
                       probe_update_scale, part, g, nscan, S, pw, chunk, inv_scale);
  else if (g.p == 5)
    hipLaunchKernelGGL(pfa_inv_products_kernel<5>, grid, block, lds, stream, (const cf*)subtiles,
                       (const cf*)patches, P, (cf*)objproj, (cf*)chi0, (float*)m_probe_update,
                       probe_update_scale, part, g, nscan, S, pw, chunk, inv_scale);
  else
    hipLaunchKernelGGL(pfa_inv_products_kernel<7>, grid, block, lds, stream, (const cf*)subtiles,
                       (const cf*)patches, P, (cf*)objproj, (cf*)chi0, (float*)m_probe_update,
                       probe_update_scale, part, g, nscan, S, pw, chunk, inv_scale);
  TK_LAUNCH_CHECK();
  if (part) return tk_ordered_sum((float*)m_probe_update, part, nmpu, nchunk, true, stream);
  return TK_OK;
}

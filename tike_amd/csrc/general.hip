// The lstsq_grad / rpie chunk body for ANY shape (round 6): probe window
// narrower than the detector, any number of modes, any detector size the
// mixed-radix engine plans (n = 2^a 3^b 5^c 7^d 11^e 13^f <= 4096) -- what the
// reference runs at one speed through cuFFT (ptycho/solvers/lstsq.py:422-579)
// and the fused power-of-two kernels of ptycho.hip / lstsq.hip do not serve.
// Until round 5 these shapes took the unfused round-1 path: patch x probe
// stored zero-padded, two generic transforms in place, a gradient pass over
// the stored far plane, two more transforms, a crop, then chi read back by the
// gradient kernel: ~14 T of traffic per position (T = 8 S det^2).
//
// Three launches on the LDS line engine (fft_mixed.h), zero padding never
// stored, the far plane and chi never stored:
//
//   K1 tike_gen_fwd_rows           (position, group of probe rows):
//        bilinear patch rows gathered ONCE for all modes (-> patches), x probe
//        (shared, per position, or eigen probes on the fly), zero-padded to
//        det IN LDS, row transforms -> hand1 (nscan, S, pw, det)
//   K2 tike_gen_cols_gradient      (position, group of L detector columns):
//        sweep 1 over the modes: columns of hand1 (rows outside the probe
//        window are zeros made in LDS) -> forward column transform -> the
//        intensity of the group's pixels accumulates in LDS; gradient factor
//        and cost from the counts; sweep 2: transform again, x factor, inverse
//        column transform, rows of the probe window only -> hand2 (nscan, S,
//        pw, det)
//   K3 tike_gen_inv_rows_gradients (probe row y, chunk of positions):
//        the S lines of row y of one position -> inverse row transforms ->
//        crop -> chi in LDS: objproj = sum_s conj(P_s) chi_s, chi0, and the
//        probe gradient of the row accumulates in LDS over the chunk (one
//        atomic per pixel, mode and chunk)
//
// then tike_scatter_patches and the packed tail as on the fused path.  Traffic
// per position: 5 T pw/det + D + 5 P (hand1 written once and read twice, hand2
// written and read once).
#include "fft_mixed.h"
#include "internal.h"
#include "tike_amd.h"

#define GEN_NT 256

static size_t gen_lds_limit() { return 150 * 1024; }

// ------------------------------------------------------------------- K1
// Index arithmetic is kept out of the element loops (profiles/r06_experiments.md:
// the first version spent 4.2 wave instructions per element, 5x the
// butterflies, on float-reciprocal divisions, 64-bit offsets and TkProbe::at):
// a line's base is uniform (scalar), a lane adds e.
__device__ __forceinline__ const cf* gen_probe_row(const TkProbe& probe, long n, int s, int y) {
  return probe.probe + n * probe.pos_stride + ((long)s * probe.pw + y) * probe.pw;
}

__global__ __launch_bounds__(GEN_NT) void gen_fwd_rows_kernel(
    const cf* __restrict__ psi, const float* __restrict__ scan, const TkProbe probe,
    cf* __restrict__ hand1, cf* __restrict__ patches, MixPlan p, const cf* __restrict__ twg,
    int nscan, int S, int pw, int det, int H, int W) {
  // work item = (position n, probe row y): the S lines of that row
  extern __shared__ __align__(16) unsigned char lds_raw[];
  cf* twl = reinterpret_cast<cf*>(lds_raw);
  cf* bufa = twl + det;
  cf* bufb = bufa + S * p.ls;
  cf* prow = bufb + S * p.ls;  // the row of the patch
  for (int k = threadIdx.x; k < det; k += GEN_NT) twl[k] = twg[k];
  const int pad = (det - pw) / 2;
  const long total = (long)H * W;
  const unsigned nitem = (unsigned)nscan * (unsigned)pw;
  const bool plain = probe.weights == nullptr;  // an explicit array, no eigen probes
  for (unsigned item = blockIdx.x; item < nitem; item += gridDim.x) {
    const unsigned n = item / (unsigned)pw;
    const int y0 = (int)(item - n * (unsigned)pw);
    const TkCorner c = tk_corner(scan, n);
    // ---- the patch row, once for all modes
    {
      const int y = c.sy + y0;
      const bool row_ok = y >= 0 && y < H;
      cf* out = patches ? patches + ((long)n * pw + y0) * pw : nullptr;
      for (int px = threadIdx.x; px < pw; px += GEN_NT) {
        const int x = c.sx + px;
        const bool ok = row_ok && x >= 0 && x < W;
        // (outside the image: pixel 0 is requested and selected away -- a load
        // behind a condition is a branch around the load)
        const cf g = tk_gather(psi, ok ? (long)y * W + x : 0L, W, total, c);
        const cf v = ok ? g : mk(0.f, 0.f);
        prow[px] = v;
        if (out) out[px] = v;
      }
    }
    __syncthreads();
    // ---- line s: patch row x probe row, zero-padded to det; the probe
    // values of four modes are requested together
    for (int e = threadIdx.x; e < det; e += GEN_NT) {
      const int px = e - pad;
      const bool in = px >= 0 && px < pw;
      const int pe = mix_pad(e);
      const cf o = prow[in ? px : 0];
      for (int s0 = 0; s0 < S; s0 += 4) {
        cf w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int sm = s0 + u < S ? s0 + u : S - 1;
          w[u] = plain ? gen_probe_row(probe, n, sm, y0)[in ? px : 0]
                       : probe.at(n, sm, in ? (long)y0 * pw + px : 0L);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (s0 + u < S) bufa[(s0 + u) * p.ls + pe] = in ? o * w[u] : mk(0.f, 0.f);
      }
    }
    __syncthreads();
    cf* res = mix_stages<false>(bufa, bufb, twl, p, S);
    for (int sm = 0; sm < S; ++sm) {
      cf* dst = hand1 + (((long)n * S + sm) * pw + y0) * det;
      const cf* src = res + sm * p.ls;
      for (int e = threadIdx.x; e < det; e += GEN_NT) dst[e] = src[mix_pad(e)];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------- K2
template <int MODEL>
__global__ __launch_bounds__(GEN_NT) void gen_cols_gradient_kernel(
    const cf* __restrict__ hand1, const float* __restrict__ data,
    const unsigned char* __restrict__ mask, const TkCostSink costs, cf* __restrict__ hand2,
    MixPlan p, const cf* __restrict__ twg, int nscan, int S, int pw, int det, int L, int logL,
    float fwd_scale, float unmeasured_scaling, float inv_nmeasured) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  cf* twl = reinterpret_cast<cf*>(lds_raw);
  cf* bufa = twl + det;
  cf* bufb = bufa + (long)L * p.ls;
  float* inten = reinterpret_cast<float*>(bufb + (long)L * p.ls);  // L x det
  __shared__ float red[4];
  for (int k = threadIdx.x; k < det; k += GEN_NT) twl[k] = twg[k];
  const int pad = (det - pw) / 2;
  const int ngrp = (det + L - 1) / L;
  const long nitem = (long)nscan * ngrp;
  const float rcp_pad = 1.0f / (float)(det - pw > 0 ? det - pw : 1);
  for (unsigned item = blockIdx.x; item < (unsigned)nitem; item += gridDim.x) {
    const long n = item / (unsigned)ngrp;
    const int grp = (int)(item - (unsigned)n * (unsigned)ngrp);
    const int x0 = grp * L;
    const int nc = det - x0 < L ? det - x0 : L;
    // columns x0 .. x0 + nc of mode s into buffer a: rows of the probe window
    // from hand1, zeros above and below
    auto load = [&](int s) {
      const cf* src = hand1 + (n * S + s) * (long)pw * det + x0;
      for (int base = threadIdx.x; base < (pw << logL); base += GEN_NT * 8) {
        cf v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * GEN_NT;
          const int l = idx & (L - 1), r = idx >> logL;
          const bool ok = r < pw && l < nc;
          v[u] = src[ok ? (long)r * det + l : 0L];
        }
        asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x), "+v"(v[4].x),
                     "+v"(v[5].x), "+v"(v[6].x), "+v"(v[7].x));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * GEN_NT;
          const int l = idx & (L - 1), r = idx >> logL;
          if (r < pw && l < nc) bufa[l * p.ls + mix_pad(pad + r)] = v[u];
        }
      }
      for (int idx = threadIdx.x; idx < nc * (det - pw); idx += GEN_NT) {
        const int l = mix_div(idx, rcp_pad), q = idx - l * (det - pw);
        bufa[l * p.ls + mix_pad(q < pad ? q : q + pw)] = mk(0.f, 0.f);
      }
    };
    for (int idx = threadIdx.x; idx < nc * det; idx += GEN_NT) inten[idx] = 0.f;
    __syncthreads();
    // ---- sweep 1: intensity of the group's pixels
    for (int s = 0; s < S; ++s) {
      load(s);
      __syncthreads();
      const cf* res = mix_stages<false>(bufa, bufb, twl, p, nc);
      for (int idx = threadIdx.x; idx < (det << logL); idx += GEN_NT) {
        const int l = idx & (L - 1), k = idx >> logL;
        if (l < nc) inten[l * det + k] += norm2(res[l * p.ls + mix_pad(k)] * fwd_scale);
      }
      __syncthreads();
    }
    // ---- gradient factor (objective.py:31-44,97-109; lstsq.py:491-502) and
    // cost; counts of unmeasured pixels may be NaN: selected away, never used
    float cost = 0.f;
    for (int idx = threadIdx.x; idx < (det << logL); idx += GEN_NT) {
      const int l = idx & (L - 1), k = idx >> logL;
      const long pix = (long)k * det + x0 + (l < nc ? l : 0);
      const float dv = data[n * (long)det * det + pix];
      const bool measured = mask ? mask[pix] != 0 : true;
      if (l >= nc) continue;
      const float I = inten[l * det + k];
      float g, term;
      if (MODEL == 0) {
        const float sI = sqrtf(I), sd = sqrtf(dv);
        const float diff = sI - sd;
        term = diff * diff;
        g = -(1.0f - sd / (sI + 1e-9f));
      } else {
        term = I - dv * logf(I + 1e-9f);
        g = -(1.0f - dv / (I + 1e-9f));
      }
      cost += measured ? term : 0.f;
      inten[l * det + k] = (measured ? g : unmeasured_scaling - 1.0f) * fwd_scale;
    }
    if (costs.costs) {
      cost = tk_block_sum256(cost, red);
      if (threadIdx.x == 0) tk_cost_add(costs, n, grp, cost * inv_nmeasured);
    }
    __syncthreads();
    if (hand2 == nullptr) continue;  // cost only
    // ---- sweep 2: transform again, x factor, inverse column transform, the
    // rows of the probe window out
    for (int s = 0; s < S; ++s) {
      load(s);
      __syncthreads();
      cf* res = mix_stages<false>(bufa, bufb, twl, p, nc);
      for (int idx = threadIdx.x; idx < (det << logL); idx += GEN_NT) {
        const int l = idx & (L - 1), k = idx >> logL;
        if (l < nc) {
          cf* q = res + l * p.ls + mix_pad(k);
          *q = *q * inten[l * det + k];
        }
      }
      __syncthreads();
      const cf* back = mix_stages<true>(res, res == bufa ? bufb : bufa, twl, p, nc);
      cf* dst = hand2 + (n * S + s) * (long)pw * det + x0;
      for (int idx = threadIdx.x; idx < (pw << logL); idx += GEN_NT) {
        const int l = idx & (L - 1), r = idx >> logL;
        if (l < nc) dst[(long)r * det + l] = back[l * p.ls + mix_pad(pad + r)];
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------- K3
// mpu: S x pw x pw complex as float pairs; with `part` != nullptr (deterministic
// mode) chunk c leaves its sums in part[c] instead of adding them atomically.
// The S rows of the NEXT position are requested (GEN_K3_PRE elements per
// thread, (mode, e) decoded with e padded to a power of two: no division)
// before the butterflies of the position in hand.
#define GEN_K3_PRE 8
__global__ __launch_bounds__(GEN_NT) void gen_inv_rows_gradients_kernel(
    const cf* __restrict__ hand2, const cf* __restrict__ patches, const TkProbe probe,
    cf* __restrict__ objproj, cf* __restrict__ chi0, float* __restrict__ mpu, float mpu_scale,
    float* __restrict__ part, MixPlan p, const cf* __restrict__ twg, int nscan, int S, int pw,
    int det, int logD, int chunk, float inv_scale) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  cf* twl = reinterpret_cast<cf*>(lds_raw);
  cf* bufa = twl + det;
  cf* bufb = bufa + S * p.ls;
  cf* acc = bufb + S * p.ls;  // S x pw
  for (int k = threadIdx.x; k < det; k += GEN_NT) twl[k] = twg[k];
  const int pad = (det - pw) / 2;
  const int nchunk = (nscan + chunk - 1) / chunk;
  const long nitem = (long)pw * nchunk;
  const bool grad = mpu != nullptr || part != nullptr;
  const bool plain = probe.weights == nullptr;
  const int D = 1 << logD;  // >= det
  const int mode_stride = pw * det;  // elements between the rows y of two modes
  const int slots = S << logD;       // (mode, e) slots of one position
  for (unsigned item = blockIdx.x; item < (unsigned)nitem; item += gridDim.x) {
    // (rows fastest: the workgroups of one chunk run together and share the
    // chunk's probe / patch lines in L2)
    const long c = item / (unsigned)pw;
    const int y = (int)(item - (unsigned)c * (unsigned)pw);
    const long n0 = c * chunk, n1 = n0 + chunk < nscan ? n0 + chunk : nscan;
    if (grad)
      for (int idx = threadIdx.x; idx < S * pw; idx += GEN_NT) acc[idx] = mk(0.f, 0.f);
    cf pre[GEN_K3_PRE];
    auto request = [&](long n) {
      const cf* src = hand2 + (n * S * pw + y) * (long)det;
#pragma unroll
      for (int u = 0; u < GEN_K3_PRE; ++u) {
        int idx = threadIdx.x + u * GEN_NT;
        asm volatile("" : "+v"(idx));  // (not hoisted out of the position loop)
        const int sm = idx >> logD, e = idx & (D - 1);
        pre[u] = src[idx < slots && e < det ? sm * mode_stride + e : 0];
      }
    };
    request(n0);
    for (long n = n0; n < n1; ++n) {
      // ---- row y of every mode of position n into LDS
#pragma unroll
      for (int u = 0; u < GEN_K3_PRE; ++u) asm volatile("" : "+v"(pre[u].x));
#pragma unroll
      for (int u = 0; u < GEN_K3_PRE; ++u) {
        int idx = threadIdx.x + u * GEN_NT;
        asm volatile("" : "+v"(idx));
        const int sm = idx >> logD, e = idx & (D - 1);
        if (idx < slots && e < det) bufa[sm * p.ls + mix_pad(e)] = pre[u];
      }
      if (slots > GEN_K3_PRE * GEN_NT) {  // (more modes than the registers take)
        const cf* src = hand2 + (n * S * pw + y) * (long)det;
        for (int idx = threadIdx.x + GEN_K3_PRE * GEN_NT; idx < slots; idx += GEN_NT) {
          const int sm = idx >> logD, e = idx & (D - 1);
          if (e < det) bufa[sm * p.ls + mix_pad(e)] = src[sm * mode_stride + e];
        }
      }
      // the object patch row of this position (needed after the butterflies)
      const cf* orow = patches + (n * pw + y) * (long)pw;
      // (requested before the butterflies, used after them)
      const cf O0 = conjf(orow[threadIdx.x < pw ? threadIdx.x : 0]);
      __syncthreads();
      if (n + 1 < n1) request(n + 1);
      const cf* res = mix_stages<true>(bufa, bufb, twl, p, S);
      // ---- chi = crop(res) * scale: both products, mode 0
      cf* oproj = objproj ? objproj + (n * pw + y) * (long)pw : nullptr;
      cf* c0 = chi0 ? chi0 + (n * pw + y) * (long)pw : nullptr;
      for (int x = threadIdx.x; x < pw; x += GEN_NT) {
        const cf Ox = x == (int)threadIdx.x ? O0 : conjf(orow[x]);
        const cf* q = res + mix_pad(pad + x);
        cf op = mk(0.f, 0.f);
        for (int s0 = 0; s0 < S; s0 += 4) {
          cf P[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int sm = s0 + u < S ? s0 + u : S - 1;
            P[u] = plain ? gen_probe_row(probe, n, sm, y)[x]
                         : probe.at(n, sm, (long)y * pw + x);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int sm = s0 + u;
            if (sm < S) {
              const cf chi = q[sm * p.ls] * inv_scale;
              op = op + conjf(P[u]) * chi;
              if (grad) acc[sm * pw + x] = acc[sm * pw + x] + Ox * chi;
              if (sm == 0 && c0) c0[x] = chi;
            }
          }
        }
        if (oproj) oproj[x] = op;
      }
      __syncthreads();
    }
    if (grad) {
      for (int sm = 0; sm < S; ++sm) {
        const long o = 2 * (((long)sm * pw + y) * pw);
        float* dst = part ? part + c * 2L * S * pw * pw + o : mpu + o;
        for (int x = threadIdx.x; x < pw; x += GEN_NT) {
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
      __syncthreads();
    }
  }
}

// ------------------------------------------------- K2, one sweep (resident)
// Where all S modes of a column group fit LDS beside each other (S x L lines,
// two buffers): hand1 is read ONCE -- the forward column transforms of every
// mode are in LDS when the intensity is formed, the factor is applied in
// place and the inverse runs on the same lines.  One 1024-thread workgroup
// per CU; the columns of the NEXT work item are requested into registers
// before the butterflies of the one in hand (GEN_PF elements per thread), the
// counts of the item before its forward transform.
#define GEN_BIG 1024
#ifndef GEN_K2_NT
#define GEN_K2_NT 512
#endif

__device__ __forceinline__ float gen_block_sum(float v, float* red) {
  v = tk_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

template <int MODEL, int NT>
__global__ __launch_bounds__(NT) void gen_cols_resident_kernel(
    const cf* __restrict__ hand1, const float* __restrict__ data,
    const unsigned char* __restrict__ mask, const TkCostSink costs, cf* __restrict__ hand2,
    MixPlan p, const cf* __restrict__ twg, int nscan, int S, int pw, int det, int L, int logL,
    int logQ, float fwd_scale, float unmeasured_scaling, float inv_nmeasured) {
  constexpr int PF = 8192 / NT, DV = 4096 / NT;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  cf* twl = reinterpret_cast<cf*>(lds_raw);
  cf* bufa = twl + det;
  cf* bufb = bufa + (long)S * L * p.ls;
  __shared__ float red[16];
  for (int k = threadIdx.x; k < det; k += NT) twl[k] = twg[k];
  const int pad = (det - pw) / 2;
  const int ngrp = (det + L - 1) / L;
  const long nitem = (long)nscan * ngrp;
  const int nl = S * L;
  const float rcp_pad = 1.0f / (float)(det - pw > 0 ? det - pw : 1);
  // element idx of an item: column l of row r of mode s -> (line, r, offset in
  // hand1 / hand2 or -1 for the columns a short last group does not have)
  // element slot idx of an item -> (mode s, row r, column l): the (row,
  // column) pairs of a mode are padded to Q = 2^logQ slots, so the decode is
  // shifts and masks (a float-reciprocal division and 64-bit offsets per
  // element cost a quarter of the kernel's vector instructions).  Returns the
  // 32-bit offset from the item's base in hand1 / hand2, or -1 for a slot
  // without an element.  (idx is made opaque at every use: (line, r) of a
  // slot do not depend on the item, and hoisted out of the item loop -- for
  // every prefetch slot -- they would stay alive across the butterflies: 256
  // VGPRs and 344 B/lane of scratch instead of 194 and none.)
  const int Q = 1 << logQ, mode_stride = pw * det;
  auto where = [&](int x0, int idx, int& line, int& r) -> int {
    asm volatile("" : "+v"(idx));
    const int s = idx >> logQ, q2 = idx & (Q - 1);
    const int l = q2 & (L - 1);
    r = q2 >> logL;
    line = s * L + l;
    return s < S && r < pw && x0 + l < det ? s * mode_stride + r * det + l : -1;
  };
  const int slots = S << logQ;
  cf pre[PF];
  auto request = [&](unsigned item) {
    const long n = item / (unsigned)ngrp;
    const int x0 = (int)(item - (unsigned)n * (unsigned)ngrp) * L;
    const cf* src = hand1 + n * S * (long)mode_stride + x0;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      int line, r;
      const int off = where(x0, threadIdx.x + u * NT, line, r);
      pre[u] = src[off >= 0 ? off : 0];
    }
  };
  if (blockIdx.x < nitem) request(blockIdx.x);
  for (unsigned item = blockIdx.x; item < (unsigned)nitem; item += gridDim.x) {
    const long n = item / (unsigned)ngrp;
    const int grp = (int)(item - (unsigned)n * (unsigned)ngrp), x0 = grp * L;
#pragma unroll
    for (int u = 0; u < PF; ++u) asm volatile("" : "+v"(pre[u].x));
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = threadIdx.x + u * NT;
      int line, r;
      const int off = where(x0, idx, line, r);
      // (columns the group does not have: zeros, so that the arithmetic on
      // their lines stays finite)
      if (idx < slots && r < pw)
        bufa[line * p.ls + mix_pad(pad + r)] = off >= 0 ? pre[u] : mk(0.f, 0.f);
    }
    for (int idx = threadIdx.x; idx < nl * (det - pw); idx += NT) {
      const int line = mix_div(idx, rcp_pad), q = idx - line * (det - pw);
      bufa[line * p.ls + mix_pad(q < pad ? q : q + pw)] = mk(0.f, 0.f);
    }
    // the counts of the item's det x L pixels (requested with everything else,
    // never behind the mask test; NaN at unmeasured pixels is selected away)
    float dv[DV];
    unsigned measured = 0;
#pragma unroll
    for (int j = 0; j < DV; ++j) {
      const int idx = threadIdx.x + j * NT;
      const int l = idx & (L - 1), k = idx >> logL;
      const bool ok = k < det && x0 + l < det;
      const long pix = ok ? (long)k * det + x0 + l : 0L;
      dv[j] = data[n * (long)det * det + pix];
      measured |= (mask ? mask[pix] != 0 : true) ? 1u << j : 0u;
    }
    __syncthreads();
    if (item + gridDim.x < nitem) request(item + gridDim.x);
    cf* res = mix_stages<false>(bufa, bufb, twl, p, nl);
    float cost = 0.f;
#pragma unroll
    for (int j = 0; j < DV; ++j) {
      const int idx = threadIdx.x + j * NT;
      const int l = idx & (L - 1), k = idx >> logL;
      if (!(k < det && x0 + l < det)) continue;
      const bool m = (measured >> j) & 1u;
      cf* q = res + l * p.ls + mix_pad(k);
      float I = 0.f;
      for (int s = 0; s < S; ++s) I += norm2(q[s * L * p.ls] * fwd_scale);
      float g, term;
      if (MODEL == 0) {
        const float sI = sqrtf(I), sd = sqrtf(dv[j]);
        const float diff = sI - sd;
        term = diff * diff;
        g = -(1.0f - sd / (sI + 1e-9f));
      } else {
        term = I - dv[j] * logf(I + 1e-9f);
        g = -(1.0f - dv[j] / (I + 1e-9f));
      }
      cost += m ? term : 0.f;
      const float gg = (m ? g : unmeasured_scaling - 1.0f) * fwd_scale;
      for (int s = 0; s < S; ++s) q[s * L * p.ls] = q[s * L * p.ls] * gg;
    }
    if (costs.costs) {
      cost = gen_block_sum(cost, red);
      if (threadIdx.x == 0) tk_cost_add(costs, n, grp, cost * inv_nmeasured);
    }
    __syncthreads();
    if (hand2) {
      const cf* back = mix_stages<true>(res, res == bufa ? bufb : bufa, twl, p, nl);
      cf* dst = hand2 + n * S * (long)mode_stride + x0;
      for (int idx = threadIdx.x; idx < slots; idx += NT) {
        int line, r;
        const int off = where(x0, idx, line, r);
        if (off >= 0) dst[off] = back[line * p.ls + mix_pad(pad + r)];
      }
    }
    __syncthreads();
  }
}

// columns per item of the resident K2 (0: it does not fit, two sweeps)
static int gen_cols_resident(const MixPlan& p, int S, int pw) {
  for (int L = 8; L >= 2; L /= 2) {
    const size_t need = sizeof(cf) * ((size_t)p.n + 2 * (size_t)p.ls * S * L);
    // (the prefetch registers take S x Q slots, Q = pow2 >= pw * L)
    long Q = 1;
    while (Q < (long)pw * L) Q *= 2;
    if (need <= gen_lds_limit() && (long)p.n * L <= 4096 && Q * S <= 8192) return L;
  }
  return 0;
}

// ---------------------------------------------------------------- host side
static int log2_ceil(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static int log2_floor(int v) {
  int l = 0;
  while ((2 << l) <= v) ++l;
  return l;
}

// LDS of K1: twiddles, the S lines of a probe row twice, the patch row
static size_t gen_k1_lds(const MixPlan& p, int S, int pw) {
  return sizeof(cf) * ((size_t)p.n + 2 * (size_t)p.ls * S + (size_t)pw);
}
// column groups of the two-sweep K2 for a shape (0: no fit)
static int gen_cols_per_item(const MixPlan& p) {
  const size_t tw = sizeof(cf) * (size_t)p.n;
  for (int L = 8; L >= 1; L /= 2) {
    const size_t need = tw + (2 * sizeof(cf) * (size_t)p.ls + sizeof(float) * (size_t)p.n) * L;
    if (need <= (L > 2 ? 50 * 1024 : gen_lds_limit())) return L;
  }
  return 0;
}
static size_t gen_k3_lds(const MixPlan& p, int S, int pw) {
  return sizeof(cf) * ((size_t)p.n + 2 * (size_t)p.ls * S + (size_t)S * pw);
}

extern "C" int tike_gen_supported(int S, int pw, int det) {
  if (S < 1 || pw < 1 || det < pw || det > TK_MIX_MAX_N) return 0;
  MixPlan p;
  if (!mix_make_plan(det, &p)) return 0;  // (Bluestein sizes: the unfused path)
  return gen_k1_lds(p, S, pw) <= gen_lds_limit() && gen_cols_per_item(p) > 0 &&
                 gen_k3_lds(p, S, pw) <= gen_lds_limit()
             ? 1
             : 0;
}

template <class K>
static int gen_lds_attr(K kern, size_t lds) {
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  return TK_OK;
}

extern "C" int tike_gen_fwd_rows(const void* psi, const float* scan, const void* probe,
                                 int probe_per_scan, const void* unique,
                                 const void* eigen_probe, const float* eigen_weights,
                                 int num_eigen, int eigen_modes, void* hand1, void* patches,
                                 int nscan, int S, int pw, int det, int H, int W, void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 1 && det >= pw && H >= 1 && W >= 1);
  TK_CHECK_ARG(!(eigen_weights && probe_per_scan));
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(psi && scan && probe && hand1);
  if (!tike_gen_supported(S, pw, det)) return TK_ERR_UNSUPPORTED;
  const MixTables* t = tk_mix_tables(det);
  if (!t || t->bluestein) return TK_ERR_UNSUPPORTED;
  const TkProbe P = tk_make_probe(probe, probe_per_scan, eigen_probe, eigen_weights, num_eigen,
                                  eigen_modes, S, pw, unique);
  const size_t lds = gen_k1_lds(t->plan, S, pw);
  int rc = gen_lds_attr(gen_fwd_rows_kernel, lds);
  if (rc) return rc;
  const long nitem = (long)nscan * pw;
  if (nitem >= (1L << 31)) return TK_ERR_ARG;
  hipLaunchKernelGGL(gen_fwd_rows_kernel, dim3(tk_grid(nitem, 8)), dim3(GEN_NT), lds, stream,
                     (const cf*)psi, scan, P, (cf*)hand1, (cf*)patches, t->plan, t->tw, nscan, S,
                     pw, det, H, W);
  TK_LAUNCH_CHECK();
  return TK_OK;
}

extern "C" int tike_gen_cols_gradient(const void* hand1, const float* data,
                                      const unsigned char* measured, float* costs, void* hand2,
                                      int nscan, int S, int pw, int det, float fwd_scale,
                                      int model, float unmeasured_scaling, long num_measured,
                                      void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 1 && det >= pw);
  TK_CHECK_ARG((model == 0 || model == 1) && num_measured > 0);
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(hand1 && data);
  if (!tike_gen_supported(S, pw, det)) return TK_ERR_UNSUPPORTED;
  const MixTables* t = tk_mix_tables(det);
  if (!t || t->bluestein) return TK_ERR_UNSUPPORTED;
  const int LR = gen_cols_resident(t->plan, S, pw);
  const int L = LR ? LR : gen_cols_per_item(t->plan), logL = log2_floor(L);
  const int ngrp = (det + L - 1) / L;
  TkCostSink sink;
  int rc = tk_cost_sink(costs, nscan, ngrp, stream, &sink);
  if (rc) return rc;
  const size_t lds =
      LR ? sizeof(cf) * ((size_t)det + 2 * (size_t)t->plan.ls * S * L)
         : sizeof(cf) * (size_t)det +
               (2 * sizeof(cf) * (size_t)t->plan.ls + sizeof(float) * (size_t)det) * L;
  const long nitem = (long)nscan * ngrp;
  if (nitem >= (1L << 31)) return TK_ERR_ARG;
  const float inv = 1.0f / (float)num_measured;
  const int logQ = log2_ceil(pw * L);
#define COMMA ,
#define TK_GEN_K2(KERN, NT, PER_CU, EXTRA)                                                             \
  do {                                                                                          \
    rc = gen_lds_attr(KERN, lds);                                                               \
    if (rc) return rc;                                                                          \
    hipLaunchKernelGGL(KERN, dim3(tk_grid(nitem, PER_CU)), dim3(NT), lds, stream,               \
                       (const cf*)hand1, data, measured, sink, (cf*)hand2, t->plan, t->tw,      \
                       nscan, S, pw, det, L, logL, EXTRA fwd_scale, unmeasured_scaling, inv);   \
  } while (0)
  if (LR && model == 0)
    TK_GEN_K2((gen_cols_resident_kernel<0, GEN_K2_NT>), GEN_K2_NT, 1, logQ COMMA);
  else if (LR)
    TK_GEN_K2((gen_cols_resident_kernel<1, GEN_K2_NT>), GEN_K2_NT, 1, logQ COMMA);
  else if (model == 0)
    TK_GEN_K2(gen_cols_gradient_kernel<0>, GEN_NT, 8, );
  else
    TK_GEN_K2(gen_cols_gradient_kernel<1>, GEN_NT, 8, );
#undef TK_GEN_K2
#undef COMMA
  TK_LAUNCH_CHECK();
  return tk_cost_finish(sink, nscan, stream);
}

extern "C" int tike_gen_inv_rows_gradients(const void* hand2, const void* patches,
                                           const void* probe, int probe_per_scan,
                                           const void* unique, const void* eigen_probe,
                                           const float* eigen_weights, int num_eigen,
                                           int eigen_modes, void* objproj, void* chi0,
                                           void* m_probe_update, float probe_update_scale,
                                           int nscan, int S, int pw, int det, float inv_scale,
                                           void* stream_) {
  TK_ENTER();
  hipStream_t stream = (hipStream_t)stream_;
  TK_CHECK_ARG(nscan >= 0 && S >= 1 && pw >= 1 && det >= pw);
  TK_CHECK_ARG(!(eigen_weights && probe_per_scan));
  if (nscan == 0) return TK_OK;
  TK_CHECK_ARG(hand2 && patches && probe);
  if (!tike_gen_supported(S, pw, det)) return TK_ERR_UNSUPPORTED;
  const MixTables* t = tk_mix_tables(det);
  if (!t || t->bluestein) return TK_ERR_UNSUPPORTED;
  const TkProbe P = tk_make_probe(probe, probe_per_scan, eigen_probe, eigen_weights, num_eigen,
                                  eigen_modes, S, pw, unique);
  const size_t lds = gen_k3_lds(t->plan, S, pw);
  int rc = gen_lds_attr(gen_inv_rows_gradients_kernel, lds);
  if (rc) return rc;
  // chunks: pw x nchunk work items, about eight per CU; at least 8 positions
  // per chunk (one atomic per pixel, mode and chunk)
  int nchunk = (int)((2048 + pw - 1) / pw);
  if (nchunk > (nscan + 7) / 8) nchunk = (nscan + 7) / 8;
  if (nchunk < 1) nchunk = 1;
  int chunk = (nscan + nchunk - 1) / nchunk;
  nchunk = (nscan + chunk - 1) / chunk;
  if ((long)pw * nchunk >= (1L << 31)) return TK_ERR_ARG;
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
  hipLaunchKernelGGL(gen_inv_rows_gradients_kernel, dim3(tk_grid((long)pw * nchunk, 8)),
                     dim3(GEN_NT), lds, stream, (const cf*)hand2, (const cf*)patches, P,
                     (cf*)objproj, (cf*)chi0, (float*)m_probe_update, probe_update_scale, part,
                     t->plan, t->tw, nscan, S, pw, det, log2_ceil(det), chunk, inv_scale);
  TK_LAUNCH_CHECK();
  if (part)
    return tk_ordered_sum((float*)m_probe_update, part, nmpu, nchunk, true, stream);
  return TK_OK;
}

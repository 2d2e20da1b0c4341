// Internal (non-ABI) entry points shared between translation units.
#pragma once
#include "common.h"

int tk_fft2(const cf* in, cf* out, long ntile, int n, int inverse, float scale,
            hipStream_t stream);
// csrc/fft_mixed.hip: the shape-general engine (any n it supports; l_rows /
// l_cols = lines per workgroup group, 0 = planner's choice)
int tk_fft2_general(const cf* in, cf* out, long ntile, int n, int inverse, float scale,
                    int l_rows, int l_cols, hipStream_t stream);
int tk_conv_fwd(const cf* psi, const float* scan, const TkProbe& probe, cf* nearplane, int nscan,
                int S, int pw, int det, int H, int W, hipStream_t stream);
int tk_conv_adj(const cf* nearplane, const float* scan, const TkProbe& probe, cf* psi, int nscan,
                int S, int pw, int det, int H, int W, hipStream_t stream);
int tk_conv_adj_probe(const cf* nearplane, const float* scan, const cf* psi, cf* probe_adj,
                      int nscan, int S, int pw, int det, int H, int W, hipStream_t stream);

// csrc/adjoint.hip: pass 1 of the two-pass transform on plain tiles (out != in),
// and inverse pass 2 in place fused with the adjoint products
int tk_fft2_pass1(const cf* in, cf* out, long ntile, int det, bool inverse, bool keep,
                  hipStream_t stream);
int tk_ifft2_pass2_products(cf* work, const cf* psi, const float* scan, const cf* probe,
                            int probe_per_scan, cf* objproj, float* pnum, float pnum_scale,
                            cf* chi0, int out, int nscan, int S, int det, int H, int W,
                            float inv_scale, hipStream_t stream);

// ---- deterministic mode (tike_set_deterministic, fft2.hip).  Off: sums that
// several workgroups contribute to are float atomics (the reference's scheme,
// operators/cupy/convolution.cu:51-66): their order, and with it the last bits
// of every result, change from run to run.  On: every such sum has ONE
// contributor per address, or its partial sums go to the caller's scratch
// buffer and are added in a fixed order.
bool tk_deterministic();
// `bytes` of the caller's scratch buffer (nullptr when it is too small or the
// mode is off); one user at a time: launches are stream ordered
float* tk_det_scratch(size_t bytes);

// Per-pattern cost sink: costs[n] += v by one atomic per workgroup, or -- in
// deterministic mode -- part[n * nslots + slot] = v, summed in slot order by
// tk_cost_finish after the launch.
struct TkCostSink {
  float* costs;
  float* part;
  int nslots;
};
#if defined(__HIPCC__)
__device__ __forceinline__ void tk_cost_add(const TkCostSink& k, long n, int slot, float v) {
  if (k.part != nullptr)
    k.part[n * k.nslots + slot] = v;
  else
    unsafeAtomicAdd(&k.costs[n], v);
}
#endif
// sink for `nslots` contributors per pattern (zeroes costs when atomics are used)
int tk_cost_sink(float* costs, long nscan, int nslots, hipStream_t stream, TkCostSink* sink);
int tk_cost_finish(const TkCostSink& sink, long nscan, hipStream_t stream);
// out[i] (+)= sum_c part[c * n + i], c ascending
int tk_ordered_sum(float* out, const float* part, long n, int nparts, bool accumulate,
                   hipStream_t stream, int out_stride = 1);

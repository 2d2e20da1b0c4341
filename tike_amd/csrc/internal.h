// Internal (non-ABI) entry points shared between translation units.
#pragma once
#include "common.h"

int tk_fft2(const cf* in, cf* out, long ntile, int n, int inverse, float scale,
            hipStream_t stream);
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

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

// prefilter_common.h -- per-row scaling and error budget of the fp16 candidate pass (shared by the stand-alone
// preparation kernel in encode_topk.hip and the fused prologue of sweep_xstat_f16.h).
#pragma once

#include "common.h"

namespace qsae {

__device__ __forceinline__ float pow2_scale_for(float maxabs) {
    // power of two s with maxabs * s in [64, 128); 1 for zero, 0 (unusable) for non-finite input
    if (!(maxabs == maxabs) || maxabs == __builtin_huge_valf()) return 0.f;
    if (maxabs == 0.f) return 1.f;
    int e;
    (void)frexpf(maxabs, &e);                   // maxabs = m * 2^e, m in [0.5, 1)
    return ldexpf(1.0f, 7 - e);
}

// From a row's max |x| and sum x^2 (fp32 sums, any order): the power-of-two scale s_x of its fp16 copy,
// inv = 1 / (s_x s_w) and margin = 2 eps_b.  eps_b bounds |approximate latent - exact fp32 chain| for every hidden
// unit; meta = {s_w, max_h ||W_h||_2, max |bias|}.
__device__ __forceinline__ void pref_row_params(float mx, float ss, int D, float sw, float wn, float bmax, float& sx,
                                                float& inv, float& margin) {
    sx = pow2_scale_for(mx);
    const float nrm = sqrtf(ss) * 1.0001f;           // covers the rounding of the fp32 sum of squares (<= D 2^-24 relative)
    float eps = __builtin_huge_valf();               // non-finite row or weights: everything is a candidate -> flagged
    inv = 0.f;
    if (sx > 0.f && sw > 0.f && nrm < 3.0e38f && wn < 3.0e38f) {
        const float sd = sqrtf(static_cast<float>(D));
        const float u = 5.9604645e-8f;                              // 2^-24
        const float c1 = 9.78e-4f + 5.0f * D * u;                   // fp16 input roundings + 4x fp32 accumulation + exact chain
        eps = c1 * nrm * wn                                         // relative to sum_k |x_k||w_k| <= ||x|| ||w||
              + (D + 8.0f) * u * bmax                               // every chain step (and the final fma) also rounds at the bias' magnitude
              + 8.0f * u * nrm * wn
              + 6.0e-8f * sd * (wn / sx + nrm / sw)                 // fp16 subnormal flushing of tiny elements
              + 4.0e-6f * (nrm * wn + bmax);                        // candidate records carry the latent truncated by < 2^-18
        eps *= 1.0001f;
        inv = (1.0f / sx) * (1.0f / sw);                            // powers of two: exact
    }
    margin = 2.0f * eps;
}

}  // namespace qsae

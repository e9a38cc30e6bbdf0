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

// What a value contributes to the input error of the fp16 pass: |v - v^| with v^ = the fp16 copy as the matrix core
// reads it.  fp16 subnormals may or may not be flushed to zero on the way in, so for a subnormal copy the larger of the
// two distances counts.  `scaled` = v * s (s a power of two: exact), `back` = 1 / s.  The difference of two fp32 numbers
// this close is exact.
__device__ __forceinline__ float fp16_input_error(float v, float scaled, float back) {
    const _Float16 q = static_cast<_Float16>(scaled);
    const float kept = static_cast<float>(q) * back;
    float e = fabsf(v - kept);
    const float aq = fabsf(static_cast<float>(q));
    if (aq < 6.103515625e-5f) e = fmaxf(e, fabsf(v));          // below 2^-14: possibly read as zero
    return e;
}

// From a row's max |x|, sum x^2 and sum e^2 (e = fp16_input_error of its elements under the scale pow2_scale_for(max);
// fp32 sums, any order; ee < 0: not measured, the worst case 2^-11 |x| is assumed): the power-of-two scale s_x of its
// fp16 copy, inv = 1 / (s_x s_w) and margin = 2 eps_b.  eps_b bounds |approximate latent - exact fp32 chain| for every
// hidden unit; meta = {s_w, max_h ||W_h||_2, max |bias|, max_h ||W_h - W^_h||_2}.
//   sum x w - sum x^ w^ = sum (x - x^) w + sum x^ (w - w^)   =>   |.| <= ||e|| ||W_h|| + (||x|| + ||e||) ||f_h||
// (Cauchy-Schwarz; the measured norms are ~0.42 of the worst case 2^-11 (||x|| ||w|| + ...) for Gaussian rows.)
__device__ __forceinline__ void pref_row_params(float mx, float ss, float ee, int D, float sw, float wn, float bmax,
                                                float fn, float& sx, float& inv, float& margin) {
    sx = pow2_scale_for(mx);
    const float nrm = sqrtf(ss) * 1.0001f;           // covers the rounding of the fp32 sum of squares (<= D 2^-24 relative)
    float eps = __builtin_huge_valf();               // non-finite row or weights: everything is a candidate -> flagged
    inv = 0.f;
    // (bmax sx sw: the sweep starts its chains from bias * s_x s_w -- a bias that overflows there makes the row unservable)
    if (sx > 0.f && sw > 0.f && nrm < 3.0e38f && wn < 3.0e38f && fn < 3.0e38f && !(ee != ee) && ee < 3.0e38f &&
        bmax * sx * sw < 3.0e38f) {
        const float sd = sqrtf(static_cast<float>(D));
        const float u = 5.9604645e-8f;                              // 2^-24
        // ||e||: measured, or 2^-11 ||x|| plus what subnormal flushing can add
        const float en = ee >= 0.f ? sqrtf(ee) * 1.0001f : 4.8829e-4f * nrm + 6.2e-5f * sd / sx;
        eps = en * wn + (nrm + en) * fn                             // fp16 copies of x and W (see above)
              + 5.0f * D * u * nrm * wn                             // 4x fp32 accumulation in the matrix core + the exact chain's roundings
              + (D + 8.0f) * u * bmax                               // every chain step (and the final fma) also rounds at the bias' magnitude
              + 4.0f * D * u * bmax                                 // ... and so does the approximate chain, which starts from the bias
              + 8.0f * u * nrm * wn
              + 8.0e-6f * (nrm * wn + bmax);                        // candidate records carry the latent truncated by < 2^-17
        eps *= 1.0001f;
        inv = (1.0f / sx) * (1.0f / sw);                            // powers of two: exact
    }
    margin = 2.0f * eps;
}

}  // namespace qsae

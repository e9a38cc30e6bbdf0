// encode.hip -- encoder contraction entry points (dense latent, z-bit latent).
//
// Replaces nn.Linear(+ReLU/+Sigmoid) of SparseAutoencoder.encode (reference sae/base.py:16-19,
// sae/binary.py:82-84, sae/baseline.py:8-10, sae/ternary.py:95-98,
// sae/quantized_matryoshka.py:206-209) with the exact-fp32 MFMA contraction of
// gemm_mfma_f32.h.  Orientation: R = x (batch rows -> accumulator registers), Cm = W_enc
// (hidden units -> lanes), so one accumulator register across a half-wave is 32 consecutive
// hidden units of one batch row = one 128-byte store segment.
#include "gemm_mfma_f32.h"

namespace qsae {

// tuning switches: variables (behind qsae_debug_* setters) in the debug library only, constants in the product library
#ifdef QSAE_DEBUG_BUILD
int g_sweep_override = 0;       // 0 = heuristic (pick_sweep)
int g_stagger = 0;              // start delay of the second co-resident block, x1024 cycles
#endif

// ---- epilogues ------------------------------------------------------------------------
template <int MT, int NT, int WTM, int WTN>
struct EpiBase {
    // seed every accumulator of output column j with bias[j]
    __device__ __forceinline__ static void seed_bias(const float* bias, f32x16 (&acc)[MT][NT],
                                                     const TileCtx& c) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = c.n0 + c.wn * WTN + nt * 32 + c.lane_col;
            const float b = (bias != nullptr && col < c.N) ? bias[col] : 0.0f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = b;
        }
    }
};

template <int ACT, int BM, int BN>
struct EpiDense : EpiBase<BM / 64, BN / 64, BM / 2, BN / 2> {
    static constexpr int MT = BM / 64, NT = BN / 64, WTM = BM / 2, WTN = BN / 2;
    static constexpr int kCheckpoints = 0;
    static constexpr int kLdsFloats = 0;
    static constexpr int kStoresPerFinish = 0;   // conservative: the staged-load wait then also covers them
    template <class A> __device__ __forceinline__ void begin(const A&, const TileCtx&) {}
    template <class A> __device__ __forceinline__ void end(const A&, const TileCtx&) {}
    struct Args {
        const float* bias;
        float* out;
        int64_t ld;
    };
    __device__ __forceinline__ void init(const Args& a, f32x16 (&acc)[MT][NT], const TileCtx& c) {
        this->seed_bias(a.bias, acc, c);
    }
    __device__ __forceinline__ void checkpoint(const Args&, f32x16 (&)[MT][NT], const TileCtx&, int) {}
    __device__ __forceinline__ void finish(const Args& a, f32x16 (&acc)[MT][NT], const TileCtx& c) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c.m0 + c.wm * WTM + mt * 32 + mfma_row(r, c.lane_half);
                if (row >= c.M) continue;
                float* orow = a.out + static_cast<int64_t>(row) * a.ld;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = c.n0 + c.wn * WTN + nt * 32 + c.lane_col;
                    float v = acc[mt][nt][r];
                    if (ACT == QSAE_ACT_RELU) v = v > 0.0f ? v : 0.0f;
                    if (ACT == QSAE_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                    if (col < c.N) orow[col] = v;
                }
            }
        }
    }
};

template <int BM, int BN>
struct EpiBits : EpiBase<BM / 64, BN / 64, BM / 2, BN / 2> {
    static constexpr int MT = BM / 64, NT = BN / 64, WTM = BM / 2, WTN = BN / 2;
    static constexpr int kCheckpoints = 0;
    static constexpr int kLdsFloats = 0;
    static constexpr int kStoresPerFinish = 0;   // conservative: the staged-load wait then also covers them
    template <class A> __device__ __forceinline__ void begin(const A&, const TileCtx&) {}
    template <class A> __device__ __forceinline__ void end(const A&, const TileCtx&) {}
    struct Args {
        const float* bias;
        uint32_t* zbits;
        int64_t words_ld;
    };
    __device__ __forceinline__ void init(const Args& a, f32x16 (&acc)[MT][NT], const TileCtx& c) {
        this->seed_bias(a.bias, acc, c);
    }
    __device__ __forceinline__ void checkpoint(const Args&, f32x16 (&)[MT][NT], const TileCtx&, int) {}
    __device__ __forceinline__ void finish(const Args& a, f32x16 (&acc)[MT][NT], const TileCtx& c) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col0 = c.n0 + c.wn * WTN + nt * 32;
                const bool col_ok = (col0 + c.lane_col) < c.N;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // lanes 0-31 carry output row mfma_row(r,0), lanes 32-63 row mfma_row(r,1)
                    const unsigned long long m = __ballot(col_ok && sig_gt_half(acc[mt][nt][r]));
                    const int row = c.m0 + c.wm * WTM + mt * 32 + mfma_row(r, c.lane_half);
                    if (c.lane_col == 0 && row < c.M && col0 < c.N) {
                        const uint32_t w = c.lane_half ? static_cast<uint32_t>(m >> 32) : static_cast<uint32_t>(m);
                        a.zbits[static_cast<int64_t>(row) * a.words_ld + (col0 >> 5)] = w;
                    }
                }
            }
        }
    }
};

template <class Epi, int BM, int BN, int BK>
static int run_encoder(const float* x, const float* W, int B, int D, int H, const typename Epi::Args& ea,
                       hipStream_t s, bool kperm = false) {
    if (kperm) {     // operands stored K-interleaved: direct 16-byte LDS writes (D % BK == 0 checked by caller)
        using LA = LoaderF32<BM, BK, false, true, true>;
        using LB = LoaderF32<BN, BK, false, true, true>;
        typename LA::Args la{x, D, B};
        typename LB::Args lb{W, D, H};
        return launch_gemm<LA, LB, Epi, BM, BN, BK>(la, lb, ea, B, H, D, pick_sweep<BM, BN>(B, H, D), s);
    }
    if (D % BK == 0) {
        constexpr bool kAsm = (BM == 128 && BN == 128);     // audited spill-free instantiations only
        using LA = LoaderF32<BM, BK, false, kAsm>;
        using LB = LoaderF32<BN, BK, false, kAsm>;
        typename LA::Args la{x, D, B};
        typename LB::Args lb{W, D, H};
        return launch_gemm<LA, LB, Epi, BM, BN, BK>(la, lb, ea, B, H, D, pick_sweep<BM, BN>(B, H, D), s);
    }
    using LA = LoaderF32<BM, BK, true>;
    using LB = LoaderF32<BN, BK, true>;
    typename LA::Args la{x, D, B};
    typename LB::Args lb{W, D, H};
    return launch_gemm<LA, LB, Epi, BM, BN, BK>(la, lb, ea, B, H, D, pick_sweep<BM, BN>(B, H, D), s);
}

template <int ACT, int BM, int BN, int BK>
static int run_dense(const float* x, const float* W, const float* bias, int B, int D, int H, float* out,
                     int64_t ld, hipStream_t s, bool kperm = false) {
    using Epi = EpiDense<ACT, BM, BN>;
    typename Epi::Args ea{bias, out, ld};
    return run_encoder<Epi, BM, BN, BK>(x, W, B, D, H, ea, s, kperm);
}

template <int BM, int BN, int BK>
static int run_bits(const float* x, const float* W, const float* bias, int B, int D, int H, uint32_t* zbits,
                    int64_t words_ld, hipStream_t s) {
    using Epi = EpiBits<BM, BN>;
    typename Epi::Args ea{bias, zbits, words_ld};
    return run_encoder<Epi, BM, BN, BK>(x, W, B, D, H, ea, s);
}

// One tile shape: 128 x 128 x 32, two workgroups per CU (LDS 2 x 74 KB, <= 256 registers per lane).
// 256 x 256 tiles (one wave per SIMD) were measured slower and, with the two-deep staging sets, no
// longer fit the register file (profiles/r01_*; DESIGN.md "what was tried").

template <int ACT>
static int dispatch_dense(const float* x, const float* W, const float* bias, int B, int D, int H, float* out,
                          int64_t ld, hipStream_t s, bool kperm = false) {
    // serving-sized batches: a 64-row tile does half of the (mostly padded) matrix work of the 128-row one; same chains
    if (B <= 64) return run_dense<ACT, 64, 128, 32>(x, W, bias, B, D, H, out, ld, s, kperm);
    return run_dense<ACT, 128, 128, 32>(x, W, bias, B, D, H, out, ld, s, kperm);
}

}  // namespace qsae

using namespace qsae;

#ifdef QSAE_DEBUG_BUILD
extern "C" int qsae_debug_set_gemm_config(int) { return QSAE_OK; }   // one tile shape is built; kept for old tools

// Diagnosis only: the encoder contraction with parts of the pipeline removed (results are wrong).
extern "C" int qsae_debug_encode_ablate(const float* x, const float* W, int B, int D, int H, float* out, int cfg,
                                        int ablate, qsae_stream_t stream) {
    hipStream_t s = as_stream(stream);
#define QSAE_ABL(BM, BN, BK, AB)                                                                   \
    {                                                                                              \
        /* asm-staged loads only in the complete pipeline: an ablated build never waits for them */ \
        using LA = LoaderF32<BM, BK, false, (AB == 0)>;                                            \
        using LB = LoaderF32<BN, BK, false, (AB == 0)>;                                            \
        using Epi = EpiDense<QSAE_ACT_NONE, BM, BN>;                                               \
        typename LA::Args la{x, D, B};                                                             \
        typename LB::Args lb{W, D, H};                                                             \
        typename Epi::Args ea{nullptr, out, H};                                                    \
        return launch_gemm<LA, LB, Epi, BM, BN, BK, AB>(la, lb, ea, B, H, D, pick_sweep<BM, BN>(B, H, D), s); \
    }
    (void)cfg;
    if (ablate == 1) QSAE_ABL(128, 128, 32, 1)
    if (ablate == 2) QSAE_ABL(128, 128, 32, 2)
    QSAE_ABL(128, 128, 32, 0)
#undef QSAE_ABL
}

extern "C" int qsae_debug_set_stagger(int units) {
    g_stagger = units;
    return QSAE_OK;
}

extern "C" int qsae_debug_set_sweep(int sweep) {
    g_sweep_override = sweep;
    return QSAE_OK;
}
#endif  // QSAE_DEBUG_BUILD

static int encode_dense_impl(const float* x, const float* W, const float* bias, int B, int D, int H, int act,
                             float* out, int64_t out_ld, qsae_stream_t stream, bool kperm) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(x && W && out, "null pointer");
    QSAE_CHECK_ARG(out_ld >= H, "out_ld < H");
    QSAE_CHECK_ARG(act >= QSAE_ACT_NONE && act <= QSAE_ACT_SIGMOID, "unknown activation");
    QSAE_CHECK_SUPPORTED(D % 4 == 0, "D must be a multiple of 4");
    if (kperm) QSAE_CHECK_SUPPORTED(D % 32 == 0, "K-interleaved operands need D % 32 == 0");
    QSAE_CHECK_ARG(aligned16(x) && aligned16(W), "x and W must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    switch (act) {
        case QSAE_ACT_RELU: return dispatch_dense<QSAE_ACT_RELU>(x, W, bias, B, D, H, out, out_ld, s, kperm);
        case QSAE_ACT_SIGMOID: return dispatch_dense<QSAE_ACT_SIGMOID>(x, W, bias, B, D, H, out, out_ld, s, kperm);
        default: return dispatch_dense<QSAE_ACT_NONE>(x, W, bias, B, D, H, out, out_ld, s, kperm);
    }
}

extern "C" int qsae_encode_dense(const float* x, const float* W, const float* bias, int B, int D, int H,
                                 int act, float* out, int64_t out_ld, qsae_stream_t stream) {
    return encode_dense_impl(x, W, bias, B, D, H, act, out, out_ld, stream, false);
}

extern "C" int qsae_encode_dense_kperm(const float* xp, const float* Wp, const float* bias, int B, int D, int H,
                                       int act, float* out, int64_t out_ld, qsae_stream_t stream) {
    return encode_dense_impl(xp, Wp, bias, B, D, H, act, out, out_ld, stream, true);
}

extern "C" int qsae_encode_bits(const float* x, const float* W, const float* bias, int B, int D, int H,
                                uint32_t* zbits, int64_t words_ld, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(x && W && zbits, "null pointer");
    QSAE_CHECK_ARG(words_ld >= (H + 31) / 32, "words_ld < ceil(H/32)");
    QSAE_CHECK_SUPPORTED(D % 4 == 0, "D must be a multiple of 4");
    QSAE_CHECK_ARG(aligned16(x) && aligned16(W), "x and W must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    // bits beyond H inside the last word are written as zero by the epilogue; words the tiles
    // never touch (words_ld > ceil(H/32)) are cleared here.
    const int64_t used = (H + 31) / 32;
    if (words_ld > used)
        QSAE_HIP(hipMemset2DAsync(zbits + used, words_ld * 4, 0, (words_ld - used) * 4, B, s));
    return run_bits<128, 128, 32>(x, W, bias, B, D, H, zbits, words_ld, s);
}

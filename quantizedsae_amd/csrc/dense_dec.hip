// dense_dec.hip -- dense quantized-dictionary decoders (ternary, matryoshka).
//
// Ternary (reference sae/ternary.py:41-52): recon = h @ hard^T with hard = sign(w)*(|w|>=0.5)
// in {-1,0,+1}; the reference materialises three [D,H] fp32 temporaries per call.  Here the
// dictionary is packed once to 2-bit fields (4 MiB at 512x32768) and expanded to fp32 inside
// the LDS staging of the exact-fp32 MFMA contraction (gemm_mfma_f32.h).
//
// Matryoshka (reference sae/quantized_matryoshka.py:47-143): per nested level i,
// recon += (scale * z) @ S with z in {0,1}, S = Bsign + Bsign_mirror in {-2,0,2} and a per-row
// scale.  Packed as 2-bit fields of S/2 (transposed, hidden index contiguous) plus fp32
// 2*scale; all levels run in ONE K-walk whose accumulator is written out at every level
// boundary (the reference launches one GEMM per level and clones the running sum).
#include "gemm_mfma_f32.h"
#include "split_dec_bf16.h"

namespace qsae {

constexpr int kMaxLevels = 8;

// ---- ternary --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pack_ternary_kernel(const float* __restrict__ w, int D, int H, int words, uint32_t* __restrict__ codes) {
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= static_cast<long long>(D) * words) return;
    const int d = static_cast<int>(gid / words), wi = static_cast<int>(gid % words);
    const float* row = w + static_cast<long long>(d) * H;
    uint32_t word = 0;
    for (int f = 0; f < 16; ++f) {
        const int h = wi * 16 + f;
        if (h >= H) break;
        const float x = row[h];
        const uint32_t code = (fabsf(x) >= 0.5f) ? (x > 0.0f ? 1u : 3u) : 0u;   // ternary.py:47-49
        word |= code << (2 * f);
    }
    codes[gid] = word;
}

template <int BM, int BN>
struct EpiStore {
    static constexpr int MT = BM / 64, NT = BN / 64, WTM = BM / 2, WTN = BN / 2;
    static constexpr int kCheckpoints = 0;
    static constexpr int kLdsFloats = 0;
    static constexpr int kStoresPerFinish = 0;   // conservative: the staged-load wait then also covers them
    template <class A> __device__ __forceinline__ void begin(const A&, const TileCtx&) {}
    template <class A> __device__ __forceinline__ void end(const A&, const TileCtx&) {}
    struct Args {
        float* out;
        int64_t ld;
    };
    __device__ __forceinline__ void init(const Args&, f32x16 (&acc)[MT][NT], const TileCtx&) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    }
    __device__ __forceinline__ void checkpoint(const Args&, f32x16 (&)[MT][NT], const TileCtx&, int) {}
    __device__ __forceinline__ void finish(const Args& a, f32x16 (&acc)[MT][NT], const TileCtx& c) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c.m0 + c.wm * WTM + mt * 32 + mfma_row(r, c.lane_half);
                if (row >= c.M) continue;
                float* orow = a.out + static_cast<int64_t>(row) * a.ld;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = c.n0 + c.wn * WTN + nt * 32 + c.lane_col;
                    if (col < c.N) orow[col] = acc[mt][nt][r];
                }
            }
    }
};

// ---- matryoshka -----------------------------------------------------------------------------
struct LevelTable {
    int n;
    int end[kMaxLevels];     // exclusive end (hidden index) of each level
    float factor[kMaxLevels];  // 2^(n-i-2) * quant_step
};

// codes2t[d][h/16] field = S/2 in two's complement, S = sgn(sig(w)>=.5) + sgn(sig(wm)>=.5)
__global__ void __launch_bounds__(256)
pack_matryoshka_codes_kernel(const float* __restrict__ w, const float* __restrict__ wm, int H, int D, int words,
                             uint32_t* __restrict__ codes) {
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= static_cast<long long>(D) * words) return;
    const int d = static_cast<int>(gid / words), wi = static_cast<int>(gid % words);
    uint32_t word = 0;
    for (int f = 0; f < 16; ++f) {
        const int h = wi * 16 + f;
        if (h >= H) break;
        const long long o = static_cast<long long>(h) * D + d;
        const int half = (sig_ge_half(w[o]) ? 1 : -1) + (sig_ge_half(wm[o]) ? 1 : -1);   // -2, 0, 2
        const uint32_t code = half == 0 ? 0u : (half > 0 ? 1u : 3u);
        word |= code << (2 * f);
    }
    codes[gid] = word;
}

// scale[j] = reciprocal(sqrt(4*nz_j) + 1e-8) * factor(level(j)); one wave per hidden row.
__global__ void __launch_bounds__(256)
pack_matryoshka_scale_kernel(const float* __restrict__ w, const float* __restrict__ wm, int H, int D,
                             LevelTable lv, float* __restrict__ scale) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= H) return;
    int nz = 0;
    for (int d = lane; d < D; d += 64) {
        const long long o = static_cast<long long>(j) * D + d;
        nz += (sig_ge_half(w[o]) != sig_ge_half(wm[o])) ? 0 : 1;
    }
    for (int off = 32; off > 0; off >>= 1) nz += __shfl_down(nz, off, 64);
    if (lane == 0) {
        int level = 0;
        while (level < lv.n - 1 && j >= lv.end[level]) ++level;
        const float norm = sqrtf(static_cast<float>(4 * nz));
        const float denom = norm + 1e-8f;
        const float rcp = 1.0f / denom;
        scale[j] = rcp * lv.factor[level];
    }
}

template <int BM, int BN>
struct EpiLevels {
    static constexpr int MT = BM / 64, NT = BN / 64, WTM = BM / 2, WTN = BN / 2;
    static constexpr int kCheckpoints = 1;
    static constexpr int kLdsFloats = 0;
    static constexpr int kStoresPerFinish = 0;   // conservative: the staged-load wait then also covers them
    template <class A> __device__ __forceinline__ void begin(const A&, const TileCtx&) {}
    template <class A> __device__ __forceinline__ void end(const A&, const TileCtx&) {}
    struct Args {
        LevelTable lv;
        const float* bias;     // nullptr when allow_bias == 0
        float* levels;         // [n][B][D]
        int64_t level_stride;  // B * D
    };
    __device__ __forceinline__ void init(const Args&, f32x16 (&acc)[MT][NT], const TileCtx&) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    }
    __device__ __forceinline__ void write_level(const Args& a, f32x16 (&acc)[MT][NT], const TileCtx& c,
                                                int level) const {
        float* base = a.levels + static_cast<int64_t>(level) * a.level_stride;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = c.n0 + c.wn * WTN + nt * 32 + c.lane_col;
            if (col >= c.N) continue;
            const float b = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = c.m0 + c.wm * WTM + mt * 32 + mfma_row(r, c.lane_half);
                    if (row < c.M) base[static_cast<int64_t>(row) * c.N + col] = acc[mt][nt][r] + b;
                }
        }
    }
    __device__ __forceinline__ void checkpoint(const Args& a, f32x16 (&acc)[MT][NT], const TileCtx& c,
                                               int k_done) const {
        // level boundaries are multiples of BK (checked on the host), so every boundary coincides
        // with the end of exactly one K slice; empty levels share a boundary and are all written.
        for (int i = 0; i < a.lv.n; ++i)
            if (a.lv.end[i] == k_done) write_level(a, acc, c, i);
    }
    __device__ __forceinline__ void finish(const Args&, f32x16 (&)[MT][NT], const TileCtx&) {}
};

// l0_counts[level] += popcount of the level's z bits (level boundaries multiples of 32).  A workgroup walks whole
// rows (no division per word), a lane keeps its level's running count while its word index stays in that level.
__global__ void __launch_bounds__(256)
count_bits_kernel(const uint32_t* __restrict__ zbits, int64_t words_ld, int B, int words, LevelTable lv,
                  unsigned long long* __restrict__ counts) {
    unsigned long long local[kMaxLevels] = {0};
    for (int wi = threadIdx.x; wi < words; wi += 256) {
        int level = 0;
        while (level < lv.n - 1 && wi * 32 >= lv.end[level]) ++level;
        unsigned long long c = 0;
        for (int b = blockIdx.x; b < B; b += gridDim.x) c += __popc(zbits[static_cast<int64_t>(b) * words_ld + wi]);
#pragma unroll
        for (int l = 0; l < kMaxLevels; ++l) local[l] += (l == level) ? c : 0ull;
    }
    __shared__ unsigned long long part[4][kMaxLevels];
    for (int l = 0; l < lv.n; ++l) {
        unsigned long long v = local[l];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][l] = v;
    }
    __syncthreads();
    if (threadIdx.x < lv.n) {
        const unsigned long long v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (v) atomicAdd(&counts[threadIdx.x], v);
    }
}

// ---- matryoshka, sparse activations --------------------------------------------------------------
// With few active units per row (a trained dictionary: tens to hundreds of 32768) the dense contraction spends
// 2 H D FLOP per row on zeros.  The sparse form walks the row's z bits in ascending hidden order and adds
// scale_j * S_j for the active units only: the same fmaf chain as the MFMA kernel above, whose terms with
// z = 0 leave the accumulator unchanged, so the results are bit-identical.  Dictionary in hidden-major order
// (codes_rows[j][D/16]: one 128-byte line per unit at D = 512), a wave per activation row, every lane owns
// D/64 consecutive output columns; level outputs are written whenever the walk crosses a level boundary.
constexpr int kSpWaves = 4;
constexpr int kSpRound = 64 * 32;          // hidden units whose bits one round of the walk loads (one word per lane)

__global__ void __launch_bounds__(256)
pack_matryoshka_rows_kernel(const float* __restrict__ w, const float* __restrict__ wm, int H, int D, int row_words,
                            uint32_t* __restrict__ codes_rows) {
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= static_cast<long long>(H) * row_words) return;
    const int j = static_cast<int>(gid / row_words), wi = static_cast<int>(gid % row_words);
    uint32_t word = 0;
    for (int f = 0; f < 8; ++f) {
        const int d = wi * 8 + f;
        if (d >= D) break;
        const long long o = static_cast<long long>(j) * D + d;
        const int half = (sig_ge_half(w[o]) ? 1 : -1) + (sig_ge_half(wm[o]) ? 1 : -1);   // -2, 0, 2
        const uint32_t code = half == 0 ? 0u : (half > 0 ? 1u : 0xFu);                    // S / 2 as a 4-bit two's-complement field
        word |= code << (4 * f);
    }
    codes_rows[gid] = word;
}

// low nibble of byte `byte` of w as a signed 4-bit integer / 16, in one instruction (decode_row.h has the same helper)
__device__ __forceinline__ float sp_cvt_off_nibble(uint32_t w, int byte) {
    float r;
    if (byte == 0) asm("v_cvt_off_f32_i4_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(r) : "v"(w));
    else if (byte == 1) asm("v_cvt_off_f32_i4_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(r) : "v"(w));
    else if (byte == 2) asm("v_cvt_off_f32_i4_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(r) : "v"(w));
    else asm("v_cvt_off_f32_i4_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(r) : "v"(w));
    return r;
}
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));

// Round 3: the walk is VALU-bound at ~200 active units per row (bit-field extract + convert + fma per unit and column, 24
// instructions per 8 columns), so the dictionary rows are 4-bit fields now: v_cvt_off_f32_i4 converts a nibble in one
// instruction (to S / 32: the factor goes to the scale, 32 scale_j is exact), two columns share a v_pk_fma_f32, and up to
// eight units are in flight instead of four (a group ends at the level boundary, not before it).
template <int FPL>   // output columns per lane: D = 64 * FPL
__global__ void __launch_bounds__(64 * kSpWaves)
decode_matryoshka_sparse_kernel(const uint32_t* __restrict__ zbits, int64_t words_ld, int B, int H,
                                const uint32_t* __restrict__ codes_rows, const float* __restrict__ scale, LevelTable lv,
                                const float* __restrict__ bias, float* __restrict__ levels, int64_t level_stride) {
    constexpr int D = 64 * FPL;
    constexpr int ROW_WORDS = D / 8;
    constexpr int NW = FPL >= 8 ? FPL / 8 : 1;             // dictionary words per lane and unit
    constexpr int NP = FPL >= 2 ? FPL / 2 : 1;             // column pairs per lane
    constexpr int kUnits = 8;                              // units in flight
    __shared__ uint16_t lists[kSpWaves][kSpRound];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x * kSpWaves + wave;
    if (b >= B) return;
    uint16_t* list = lists[wave];
    typedef const __attribute__((address_space(4))) float* cflt_t;
    sp_f32x2 acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = sp_f32x2{0.0f, 0.0f};
    float bcol[FPL];
#pragma unroll
    for (int f = 0; f < FPL; ++f) bcol[f] = bias ? bias[lane * FPL + f] : 0.0f;
    // this lane's FPL four-bit fields: from word (lane*FPL)/8 of a dictionary row, bit 4*((lane*FPL)%8) on
    const int wsel = (lane * FPL) / 8;
    const int wshift = 4 * ((lane * FPL) % 8);
    int level = 0;
    auto write_level = [&](int lvl) {
        float* out = levels + static_cast<int64_t>(lvl) * level_stride + static_cast<int64_t>(b) * D + lane * FPL;
#pragma unroll
        for (int f = 0; f < FPL; ++f) out[f] = acc[f >> 1][f & 1] + bcol[f];
    };
    // the dense kernel's term is fmaf(scale_j, S, acc) with S in {-2, 0, 2}; 32 scale_j is exact, so
    // fmaf(32 scale_j, S / 32, acc) rounds the same real number (S / 32 = the field / 16)
    auto add_unit = [&](const uint32_t (&cw)[NW], float a) {
        const float t = 32.0f * a;
        const sp_f32x2 t2 = {t, t};
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const uint32_t w = cw[q] >> (FPL >= 8 ? 0 : wshift);
            const uint32_t odd = w >> 4;
            if (FPL == 1) {
                acc[0][0] = fmaf(t, sp_cvt_off_nibble(w, 0), acc[0][0]);
            } else {
#pragma unroll
                for (int p = 0; p < (FPL >= 8 ? 4 : FPL / 2); ++p)
                    acc[4 * q + p] = __builtin_elementwise_fma(t2, sp_f32x2{sp_cvt_off_nibble(w, p), sp_cvt_off_nibble(odd, p)},
                                                               acc[4 * q + p]);
            }
        }
    };
    const int words = H / 32;
    const uint32_t* zrow = zbits + static_cast<int64_t>(b) * words_ld;
    for (int w0 = 0; w0 < words; w0 += 64) {
        uint32_t word = (w0 + lane < words) ? zrow[w0 + lane] : 0u;
        // ascending list of the round's active units: exclusive prefix of the per-lane popcounts
        const int mine = __popc(word);
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total == 0) continue;
        int pos = incl - mine;
        while (word) {
            const int bit = __builtin_ctz(word);
            list[pos++] = static_cast<uint16_t>(lane * 32 + bit);
            word &= word - 1u;
        }
        asm volatile("" ::: "memory");                       // one wave's LDS operations execute in order
        const int base = w0 * 32;
        int t = 0;
        while (t < total) {
            // the next up to kUnits units (lane u holds unit t + u); the group ends where the level does
            const int jl = (lane < kUnits && t + lane < total) ? base + list[t + lane] : 0x7FFFFFFF;
            const int first = __builtin_amdgcn_readfirstlane(jl);
            while (level < lv.n - 1 && first >= lv.end[level]) write_level(level++);
            const bool inside = jl != 0x7FFFFFFF && (level == lv.n - 1 || jl < lv.end[level]);
            const int n = __popcll(__ballot(inside));        // >= 1: the list is ascending, the units inside are a prefix
            uint32_t cw[kUnits][NW];
            float a[kUnits];
#pragma unroll
            for (int u = 0; u < kUnits; ++u)
                if (u < n) {
                    const int j = __builtin_amdgcn_readlane(jl, u);
#pragma unroll
                    for (int q = 0; q < NW; ++q) cw[u][q] = codes_rows[static_cast<int64_t>(j) * ROW_WORDS + wsel + q];
                    a[u] = ((cflt_t)scale)[j];
                }
#pragma unroll
            for (int u = 0; u < kUnits; ++u)
                if (u < n) add_unit(cw[u], a[u]);
            t += n;
        }
        asm volatile("" ::: "memory");
    }
    while (level < lv.n) write_level(level++);
}

static int make_levels(int H, int n_bits, float abs_range, const int32_t* level_sizes, LevelTable& lv) {
    int32_t sizes[kMaxLevels];
    if (level_sizes) {
        long long sum = 0;
        for (int i = 0; i < n_bits; ++i) {
            if (level_sizes[i] < 0) return fail(QSAE_ERR_INVALID_ARG, "%s: negative level size", __func__);
            sizes[i] = level_sizes[i];
            sum += sizes[i];
        }
        if (sum != H) return fail(QSAE_ERR_INVALID_ARG, "%s: level sizes do not sum to H", __func__);
    } else {
        const int rc = qsae_matryoshka_sizes(H, n_bits, sizes);
        if (rc != QSAE_OK) return rc;
    }
    lv.n = n_bits;
    int acc = 0;
    const double quant_step = static_cast<double>(abs_range) / static_cast<double>(1u << (n_bits - 1));
    for (int i = 0; i < n_bits; ++i) {
        acc += sizes[i];
        lv.end[i] = acc;
        const int e = n_bits - i - 2;
        const double p = e >= 0 ? static_cast<double>(1u << e) : 1.0 / static_cast<double>(1u << (-e));
        lv.factor[i] = static_cast<float>(p * quant_step);
    }
    for (int i = n_bits; i < kMaxLevels; ++i) { lv.end[i] = acc; lv.factor[i] = 0.0f; }
    return QSAE_OK;
}

}  // namespace qsae

using namespace qsae;

extern "C" int qsae_matryoshka_sizes(int H, int n_bits, int32_t* sizes) {
    QSAE_CHECK_ARG(H > 0 && n_bits >= 1 && n_bits <= kMaxLevels && sizes, "H > 0, 1 <= n_bits <= 8, sizes != NULL");
    long long sum = 0;
    for (int i = 0; i < n_bits; ++i) { sizes[i] = (i < 2) ? 1 : (1 << (i - 1)); sum += sizes[i]; }
    if (sum != H) {
        const double sf = static_cast<double>(H) / static_cast<double>(sum);
        long long acc = 0;
        for (int i = 0; i < n_bits; ++i) {
            int s = static_cast<int>(static_cast<double>(sizes[i]) * sf);
            sizes[i] = s < 1 ? 1 : s;
        }
        for (int i = 0; i < n_bits - 1; ++i) acc += sizes[i];
        sizes[n_bits - 1] = static_cast<int32_t>(H - acc);
    }
    return QSAE_OK;
}

extern "C" int qsae_pack_ternary(const float* w, int D, int H, uint32_t* codes2, qsae_stream_t stream) {
    QSAE_CHECK_ARG(D > 0 && H > 0 && w && codes2, "D > 0, H > 0, non-null pointers");
    const int words = (H + 15) / 16;
    const long long total = static_cast<long long>(D) * words;
    hipLaunchKernelGGL(pack_ternary_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w, D, H, words, codes2);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

template <int BM, int BN, int BK>
static int run_ternary(const float* h, int64_t ld, int B, int H, const uint32_t* codes, int D, float* recon,
                       hipStream_t s) {
    using Epi = EpiStore<BM, BN>;
    typename Epi::Args ea{recon, D};
    const int64_t words = (H + 15) / 16;
    if (H % BK == 0) {
        using LA = LoaderF32<BM, BK, false>;
        using LB = LoaderCode2<BN, BK, false>;
        typename LA::Args la{h, ld, B};
        typename LB::Args lb{codes, words, D};
        return launch_gemm<LA, LB, Epi, BM, BN, BK>(la, lb, ea, B, D, H, pick_sweep<BM, BN>(B, D, H), s);
    }
    using LA = LoaderF32<BM, BK, true>;
    using LB = LoaderCode2<BN, BK, true>;
    typename LA::Args la{h, ld, B};
    typename LB::Args lb{codes, words, D};
    return launch_gemm<LA, LB, Epi, BM, BN, BK>(la, lb, ea, B, D, H, pick_sweep<BM, BN>(B, D, H), s);
}

extern "C" int qsae_decode_ternary_dense(const float* h, int64_t ld, int B, int H, const uint32_t* codes2, int D,
                                         float* recon, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && H > 0 && D > 0, "B >= 0, H > 0, D > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(h && codes2 && recon, "null pointer");
    QSAE_CHECK_ARG(ld >= H, "ld < H");
    QSAE_CHECK_SUPPORTED(H % 4 == 0 && ld % 4 == 0, "H and ld must be multiples of 4");
    QSAE_CHECK_ARG(aligned16(h), "h must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    return run_ternary<128, 128, 32>(h, ld, B, H, codes2, D, recon, s);
}

extern "C" int qsae_pack_matryoshka(const float* w, const float* wm, int H, int D, int n_bits, float abs_range,
                                    const int32_t* level_sizes, uint32_t* codes2t, float* scale,
                                    qsae_stream_t stream) {
    QSAE_CHECK_ARG(H > 0 && D > 0 && w && wm && codes2t && scale, "H > 0, D > 0, non-null pointers");
    QSAE_CHECK_ARG(n_bits >= 1 && n_bits <= kMaxLevels, "1 <= n_bits <= 8 required");
    LevelTable lv;
    const int rc = make_levels(H, n_bits, abs_range, level_sizes, lv);
    if (rc != QSAE_OK) return rc;
    hipStream_t s = as_stream(stream);
    const int words = (H + 15) / 16;
    const long long total = static_cast<long long>(D) * words;
    hipLaunchKernelGGL(pack_matryoshka_codes_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                       s, w, wm, H, D, words, codes2t);
    QSAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(pack_matryoshka_scale_kernel, dim3((H + 3) / 4), dim3(256), 0, s, w, wm, H, D, lv, scale);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

template <int BM, int BN, int BK>
static int run_matryoshka(const uint32_t* zbits, int64_t words_ld, int B, int H, int D, const uint32_t* codes,
                          const float* scale2, const typename EpiLevels<BM, BN>::Args& ea, hipStream_t s) {
    using Epi = EpiLevels<BM, BN>;
    using LA = LoaderBitsScale<BM, BK, false>;
    using LB = LoaderCode2<BN, BK, false, 2>;   // fields hold S/2
    typename LA::Args la{zbits, words_ld, B, scale2};
    typename LB::Args lb{codes, (H + 15) / 16, D};
    return launch_gemm<LA, LB, Epi, BM, BN, BK>(la, lb, ea, B, D, H, pick_sweep<BM, BN>(B, D, H), s);
}

extern "C" int qsae_decode_matryoshka(const uint32_t* zbits, int64_t words_ld, int B, int H, int D, int n_bits,
                                      const int32_t* level_sizes, const uint32_t* codes2t, const float* scale,
                                      const float* bias,
                                      int allow_bias, float* levels, unsigned long long* l0_counts,
                                      qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && H > 0 && D > 0, "B >= 0, H > 0, D > 0 required");
    QSAE_CHECK_ARG(n_bits >= 1 && n_bits <= kMaxLevels, "1 <= n_bits <= 8 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(zbits && codes2t && scale && levels, "null pointer");
    QSAE_CHECK_ARG(words_ld >= (H + 31) / 32, "words_ld < ceil(H/32)");
    QSAE_CHECK_ARG(aligned16(scale), "scale must be 16-byte aligned");
    LevelTable lv;
    const int rc = make_levels(H, n_bits, 1.0f, level_sizes, lv);
    if (rc != QSAE_OK) return rc;
    QSAE_CHECK_SUPPORTED(H % 32 == 0, "H must be a multiple of 32 (pad the dictionary)");
    for (int i = 0; i < n_bits; ++i)
        QSAE_CHECK_SUPPORTED(lv.end[i] % 32 == 0, "level boundaries must be multiples of 32 (pad the levels)");
    hipStream_t s = as_stream(stream);
    if (l0_counts) {
        QSAE_HIP(hipMemsetAsync(l0_counts, 0, sizeof(unsigned long long) * n_bits, s));
        const int words = H / 32;
        long long blocks = B < 2048 ? B : 2048;
        hipLaunchKernelGGL(count_bits_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, zbits, words_ld, B,
                           words, lv, l0_counts);
        QSAE_LAUNCH_CHECK();
    }
    typename EpiLevels<128, 128>::Args ea{lv, allow_bias ? bias : nullptr, levels, static_cast<int64_t>(B) * D};
    return run_matryoshka<128, 128, 32>(zbits, words_ld, B, H, D, codes2t, scale, ea, s);
}

extern "C" int qsae_pack_matryoshka_rows(const float* w, const float* wm, int H, int D, uint32_t* codes_rows,
                                         qsae_stream_t stream) {
    QSAE_CHECK_ARG(H > 0 && D > 0 && w && wm && codes_rows, "H > 0, D > 0, non-null pointers");
    const int row_words = (D + 7) / 8;
    const long long total = static_cast<long long>(H) * row_words;
    hipLaunchKernelGGL(pack_matryoshka_rows_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w, wm, H, D, row_words, codes_rows);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

extern "C" int qsae_decode_matryoshka_sparse(const uint32_t* zbits, int64_t words_ld, int B, int H, int D, int n_bits,
                                             const int32_t* level_sizes, const uint32_t* codes_rows, const float* scale,
                                             const float* bias, int allow_bias, float* levels,
                                             unsigned long long* l0_counts, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && H > 0 && D > 0, "B >= 0, H > 0, D > 0 required");
    QSAE_CHECK_ARG(n_bits >= 1 && n_bits <= kMaxLevels, "1 <= n_bits <= 8 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(zbits && codes_rows && scale && levels, "null pointer");
    QSAE_CHECK_ARG(words_ld >= (H + 31) / 32, "words_ld < ceil(H/32)");
    QSAE_CHECK_SUPPORTED(H % 32 == 0, "H must be a multiple of 32 (pad the dictionary)");
    QSAE_CHECK_SUPPORTED(D == 64 || D == 128 || D == 256 || D == 512 || D == 1024, "D must be 64, 128, 256, 512 or 1024");
    LevelTable lv;
    const int rc = make_levels(H, n_bits, 1.0f, level_sizes, lv);
    if (rc != QSAE_OK) return rc;
    hipStream_t s = as_stream(stream);
    if (l0_counts) {
        QSAE_HIP(hipMemsetAsync(l0_counts, 0, sizeof(unsigned long long) * n_bits, s));
        const int words = H / 32;
        long long blocks = B < 2048 ? B : 2048;
        hipLaunchKernelGGL(count_bits_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, zbits, words_ld, B,
                           words, lv, l0_counts);
        QSAE_LAUNCH_CHECK();
    }
    const dim3 grid((B + kSpWaves - 1) / kSpWaves), block(64 * kSpWaves);
    const float* bp = allow_bias ? bias : nullptr;
    const int64_t ls = static_cast<int64_t>(B) * D;
#define QSAE_SP_LAUNCH(F) hipLaunchKernelGGL(decode_matryoshka_sparse_kernel<F>, grid, block, 0, s, zbits, words_ld, B, H, \
                                             codes_rows, scale, lv, bp, levels, ls)
    switch (D / 64) {
        case 1: QSAE_SP_LAUNCH(1); break;
        case 2: QSAE_SP_LAUNCH(2); break;
        case 4: QSAE_SP_LAUNCH(4); break;
        case 8: QSAE_SP_LAUNCH(8); break;
        default: QSAE_SP_LAUNCH(16); break;
    }
#undef QSAE_SP_LAUNCH
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

// ---- the same decoders on the bf16 matrix pipe (split_dec_bf16.h) ---------------------------------------------------------
extern "C" int qsae_split_dec_supported(int B, int H, int D) { return split_dec_shape_ok(B, H, D) ? 1 : 0; }

extern "C" size_t qsae_expand_codes_bf16_bytes(int D, int H) {
    if (D != kSdBN || H <= 0 || H % (2 * kSdBK) != 0) return 0;
    return static_cast<size_t>(H) * static_cast<size_t>(D) * 2;
}

extern "C" int qsae_expand_codes_bf16(const uint32_t* codes2, int D, int H, void* tq, qsae_stream_t stream) {
    QSAE_CHECK_ARG(D > 0 && H > 0 && codes2 && tq, "D > 0, H > 0, non-null pointers");
    QSAE_CHECK_SUPPORTED(D == kSdBN && H % (2 * kSdBK) == 0, "D must be 512 and H a multiple of 64");
    QSAE_CHECK_ARG(aligned16(tq), "tq must be 16-byte aligned");
    const long long total = static_cast<long long>(H / kSdBK) * D * 4;
    hipLaunchKernelGGL(expand_codes_bf16_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), codes2, D, H, static_cast<sd_u32x4*>(tq));
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

extern "C" int qsae_decode_ternary_dense_split(const float* h, int64_t ld, int B, int H, const void* tq, int D, float* recon,
                                               qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && H > 0 && D > 0, "B >= 0, H > 0, D > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(h && tq && recon, "null pointer");
    QSAE_CHECK_SUPPORTED(split_dec_shape_ok(B, H, D), "D must be 512 and H a multiple of 64 (use qsae_decode_ternary_dense)");
    QSAE_CHECK_ARG(ld >= H && ld % 4 == 0, "ld >= H and ld a multiple of 4 required");
    QSAE_CHECK_ARG(aligned16(h) && aligned16(tq), "h and tq must be 16-byte aligned");
    SdArgs a{};
    a.h = h; a.ld = ld; a.tq = static_cast<const __bf16*>(tq); a.B = B; a.H = H; a.out = recon;
    return launch_split_dec<0>(a, as_stream(stream));
}

extern "C" int qsae_split_scale_bf16(const float* scale, int H, void* s3, qsae_stream_t stream) {
    QSAE_CHECK_ARG(H > 0 && scale && s3, "H > 0, non-null pointers");
    QSAE_CHECK_ARG(aligned16(s3) && H % 8 == 0, "s3 must be 16-byte aligned and H a multiple of 8");
    hipLaunchKernelGGL(split_scale_bf16_kernel, dim3((H + 255) / 256), dim3(256), 0, as_stream(stream), scale, H,
                       static_cast<__bf16*>(s3));
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

extern "C" int qsae_decode_matryoshka_split(const uint32_t* zbits, int64_t words_ld, int B, int H, int D, int n_bits,
                                            const int32_t* level_sizes, const void* tq, const void* s3, const float* bias,
                                            int allow_bias, float* levels, unsigned long long* l0_counts,
                                            qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && H > 0 && D > 0, "B >= 0, H > 0, D > 0 required");
    QSAE_CHECK_ARG(n_bits >= 1 && n_bits <= kMaxLevels, "1 <= n_bits <= 8 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(zbits && tq && s3 && levels, "null pointer");
    QSAE_CHECK_ARG(words_ld >= (H + 31) / 32, "words_ld < ceil(H/32)");
    QSAE_CHECK_SUPPORTED(split_dec_shape_ok(B, H, D), "D must be 512 and H a multiple of 64 (use qsae_decode_matryoshka)");
    QSAE_CHECK_ARG(aligned16(tq) && aligned16(s3), "tq and s3 must be 16-byte aligned");
    QSAE_CHECK_SUPPORTED(static_cast<unsigned long long>(B) * static_cast<unsigned long long>(words_ld) * 4ull < (1ull << 32),
                         "z bits of one call must stay below 4 GiB");
    LevelTable lv;
    const int rc = make_levels(H, n_bits, 1.0f, level_sizes, lv);
    if (rc != QSAE_OK) return rc;
    for (int i = 0; i < n_bits; ++i)
        QSAE_CHECK_SUPPORTED(lv.end[i] % 64 == 0, "level boundaries must be multiples of 64 (use qsae_decode_matryoshka)");
    hipStream_t s = as_stream(stream);
    if (l0_counts) {
        QSAE_HIP(hipMemsetAsync(l0_counts, 0, sizeof(unsigned long long) * n_bits, s));
        long long blocks = B < 2048 ? B : 2048;
        hipLaunchKernelGGL(count_bits_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, zbits, words_ld, B,
                           H / 32, lv, l0_counts);
        QSAE_LAUNCH_CHECK();
    }
    SdArgs a{};
    a.zbits = zbits; a.words_ld = words_ld; a.s3 = static_cast<const __bf16*>(s3); a.tq = static_cast<const __bf16*>(tq);
    a.B = B; a.H = H; a.out = levels; a.bias = allow_bias ? bias : nullptr; a.n_levels = n_bits;
    for (int i = 0; i < kSdMaxLevels; ++i) a.level_end[i] = i < n_bits ? lv.end[i] : H;
    return launch_split_dec<1>(a, s);
}

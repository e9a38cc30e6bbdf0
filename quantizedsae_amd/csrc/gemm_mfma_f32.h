// gemm_mfma_f32.h -- exact-fp32 "NT" contraction on the CDNA4 matrix cores.
//
//   C[i][j] = epilogue( init(i,j) (+) sum_k R[i][k] * Cm[j][k] )        k ascending
//
// Both operands are K-contiguous (activations [rows][K], dictionaries [units][K]), which is
// the layout of every contraction on the quantized-SAE forward path:
//   encoder      R = x [B][D],          Cm = W_enc [H][D]            (sae/base.py:16-19)
//   ternary dec  R = h [B][H],          Cm = codes [D][H]            (sae/ternary.py:52)
//   matryoshka   R = z*scale [B][H],    Cm = codes^T [D][H]          (sae/quantized_matryoshka.py:121)
//
// v_mfma_f32_32x32x2_f32 is bit-for-bit an fp32 fmaf chain in k order (one rounding per
// product, no wider accumulation), so walking K in ascending order makes the result equal
// to oracle/qsae_oracle.c's chain.  The MFMA takes k = 2s from lanes 0-31 and k = 2s+1 from
// lanes 32-63; to feed it with one ds_read_b128 per four MFMAs the LDS image stores each
// group of 8 consecutive k as [k0 k2 k4 k6 | k1 k3 k5 k7]: lane half h reads the 16 bytes at
// float offset 4h and register t then holds k = 8g + 2t + h, i.e. natural order.
//
// Tile: BM x BN outputs per 256-thread workgroup (2 x 2 waves, each (BM/2) x (BN/2) =
// MT x NT MFMA tiles of 32x32), K staged through LDS in BK-deep slices, double buffered,
// global -> registers -> LDS (the register hop applies the k permutation and lets packed
// 2-bit / 1-bit operands be expanded to fp32 on the way).  LDS rows are padded by 4 floats:
// row stride BK+4 makes the ds_read_b128 fragment reads and the ds_write_b64 staging writes
// bank-conflict free (bank = dword address mod 64 / mod 32).
#pragma once

#include "common.h"

namespace qsae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kGemmThreads = 256;

template <int BK>
struct TileGeom {
    static constexpr int LDS_STRIDE = BK + 4;                 // floats per LDS row
    static constexpr int CHUNKS = BK / 4;                     // 16-byte chunks per row slice
    static constexpr int ROWS_PER_PASS = kGemmThreads / CHUNKS;
};

// float offset inside an LDS row of the pair (k0,k2) of chunk c (k = 4c..4c+3); (k1,k3) is +4.
__device__ __forceinline__ int perm_pair_offset(int c) { return 8 * (c >> 1) + 2 * (c & 1); }

__device__ __forceinline__ void lds_store_chunk(float* row, int c, f32x4 v) {
    float* p = row + perm_pair_offset(c);
    *reinterpret_cast<f32x2*>(p) = f32x2{v[0], v[2]};
    *reinterpret_cast<f32x2*>(p + 4) = f32x2{v[1], v[3]};
}

// -------------------------------------------------------------------------------------
// Operand loaders.  Each stages ROWS x BK of one operand: load(kt) issues the global loads of
// K-slice kt into registers, store(tile) expands/permutes them into the LDS image.

// fp32 rows [nrows][ld], rows >= nrows are clamped (their outputs are never stored),
// columns >= K read as zero (fma(0,0,acc) == acc keeps the chain exact).
template <int ROWS, int BK, bool KTAIL = false>
struct LoaderF32 {
    using G = TileGeom<BK>;
    static constexpr int PASSES = ROWS / G::ROWS_PER_PASS;
    static_assert(ROWS % G::ROWS_PER_PASS == 0, "tile rows must be a multiple of rows per pass");
    struct Args {
        const float* p;
        int64_t ld;
        int nrows;
    };
    const float* rowp[PASSES];
    f32x4 r[PASSES];
    int K, c, lrow;

    __device__ __forceinline__ void init(const Args& a, int row0, int K_, int tid) {
        K = K_;
        c = tid % G::CHUNKS;
        lrow = tid / G::CHUNKS;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            int row = row0 + i * G::ROWS_PER_PASS + lrow;
            row = row < a.nrows ? row : a.nrows - 1;
            rowp[i] = a.p + static_cast<int64_t>(row) * a.ld + 4 * c;
        }
    }
    // KTAIL = false (K % BK == 0, the hot configuration): plain loads, nothing consumes them
    // until store(), so their latency hides under the MFMAs of the current slice.
    // KTAIL = true: chunks at or beyond K re-read the row start and are zeroed.
    __device__ __forceinline__ void load(int kt) {
        const int k0 = kt * BK;
        if (KTAIL) {
            const bool ok = (k0 + 4 * c) < K;
            const int koff = ok ? k0 : -4 * c;
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(rowp[i] + koff);
                r[i] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int i = 0; i < PASSES; ++i) r[i] = *reinterpret_cast<const f32x4*>(rowp[i] + k0);
        }
    }
    __device__ __forceinline__ void store(float* tile) const {
#pragma unroll
        for (int i = 0; i < PASSES; ++i)
            lds_store_chunk(tile + (i * G::ROWS_PER_PASS + lrow) * G::LDS_STRIDE, c, r[i]);
    }
};

// 2-bit two's-complement codes {0, +1, -1} (times MUL): codes [nrows][words_ld] uint32, 16 fields
// per word, field k of a row at bit 2*(k%16) of word k/16.  One thread expands 4 fields (one chunk).
template <int ROWS, int BK, bool KTAIL = false, int MUL = 1>
struct LoaderCode2 {
    using G = TileGeom<BK>;
    static constexpr int PASSES = ROWS / G::ROWS_PER_PASS;
    struct Args {
        const uint32_t* p;
        int64_t words_ld;
        int nrows;
    };
    const uint32_t* rowp[PASSES];
    uint32_t r[PASSES];
    int K, c, lrow;

    __device__ __forceinline__ void init(const Args& a, int row0, int K_, int tid) {
        K = K_;
        c = tid % G::CHUNKS;
        lrow = tid / G::CHUNKS;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            int row = row0 + i * G::ROWS_PER_PASS + lrow;
            row = row < a.nrows ? row : a.nrows - 1;
            rowp[i] = a.p + static_cast<int64_t>(row) * a.words_ld;
        }
    }
    __device__ __forceinline__ void load(int kt) {
        const int k = kt * BK + 4 * c;            // first field of this thread's chunk
        if (KTAIL) {
            const bool ok = k < K;
            const int w = ok ? (k >> 4) : 0;
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const uint32_t t = rowp[i][w];
                r[i] = ok ? t : 0u;
            }
        } else {
#pragma unroll
            for (int i = 0; i < PASSES; ++i) r[i] = rowp[i][k >> 4];
        }
    }
    __device__ __forceinline__ void store(float* tile) const {
        const int sh = 2 * ((4 * c) & 15);
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int w = static_cast<int>(r[i] >> sh);
            f32x4 v;
            v[0] = static_cast<float>(MUL * sbfe_i32(w, 0, 2));
            v[1] = static_cast<float>(MUL * sbfe_i32(w, 2, 2));
            v[2] = static_cast<float>(MUL * sbfe_i32(w, 4, 2));
            v[3] = static_cast<float>(MUL * sbfe_i32(w, 6, 2));
            lds_store_chunk(tile + (i * G::ROWS_PER_PASS + lrow) * G::LDS_STRIDE, c, v);
        }
    }
};

// Binary activations times a per-k scale: a[row][k] = bit(row,k) ? scale[k] : 0.
// bits [nrows][words_ld] uint32 (bit k%32 of word k/32), scale [K].
template <int ROWS, int BK, bool KTAIL = false>
struct LoaderBitsScale {
    using G = TileGeom<BK>;
    static constexpr int PASSES = ROWS / G::ROWS_PER_PASS;
    struct Args {
        const uint32_t* bits;
        int64_t words_ld;
        int nrows;
        const float* scale;
    };
    const uint32_t* rowp[PASSES];
    uint32_t r[PASSES];
    f32x4 s;
    const float* scale;
    int K, c, lrow;

    __device__ __forceinline__ void init(const Args& a, int row0, int K_, int tid) {
        K = K_;
        c = tid % G::CHUNKS;
        lrow = tid / G::CHUNKS;
        scale = a.scale;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            int row = row0 + i * G::ROWS_PER_PASS + lrow;
            row = row < a.nrows ? row : a.nrows - 1;
            rowp[i] = a.bits + static_cast<int64_t>(row) * a.words_ld;
        }
    }
    __device__ __forceinline__ void load(int kt) {
        const int k = kt * BK + 4 * c;
        if (KTAIL) {
            const bool ok = k < K;
            const int kk = ok ? k : 0;
            const f32x4 t = *reinterpret_cast<const f32x4*>(scale + kk);
            s = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const uint32_t w = rowp[i][kk >> 5];
                r[i] = ok ? w : 0u;
            }
        } else {
            s = *reinterpret_cast<const f32x4*>(scale + k);
#pragma unroll
            for (int i = 0; i < PASSES; ++i) r[i] = rowp[i][k >> 5];
        }
    }
    __device__ __forceinline__ void store(float* tile) const {
        const int sh = (4 * c) & 31;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const uint32_t w = r[i] >> sh;
            f32x4 v;
            v[0] = (w & 1u) ? s[0] : 0.f;
            v[1] = (w & 2u) ? s[1] : 0.f;
            v[2] = (w & 4u) ? s[2] : 0.f;
            v[3] = (w & 8u) ? s[3] : 0.f;
            lds_store_chunk(tile + (i * G::ROWS_PER_PASS + lrow) * G::LDS_STRIDE, c, v);
        }
    }
};

// -------------------------------------------------------------------------------------
// Workgroup -> tile mapping.  Blocks are dealt round-robin to the 8 XCDs (b % 8 labels the
// blocks that share an L2); give each XCD a contiguous run of tiles walked in groups of
// GROUP_M row-tiles so that the ~32 tiles in flight on one XCD share 8 R-panels and 4
// Cm-panels in its L2.  Bijective for any tile count.  Speed only, never correctness.
struct TileMap {
    int tiles_m, tiles_n;
    __device__ __forceinline__ void locate(int bid, int nblocks, int& tm, int& tn) const {
        constexpr int NXCD = 8, GROUP_M = 8;
        const int q = nblocks / NXCD, rem = nblocks % NXCD;
        const int xcd = bid % NXCD, slot = bid / NXCD;
        const int vid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
        const int per_group = GROUP_M * tiles_n;
        const int group = vid / per_group, in_group = vid % per_group;
        const int first_m = group * GROUP_M;
        const int gsize = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
        tm = first_m + in_group % gsize;
        tn = in_group / gsize;
    }
};

// -------------------------------------------------------------------------------------
// The kernel.  Epi provides:
//   struct Args;
//   static constexpr int kCheckpoints (0 = single epilogue at the end)
//   __device__ void init(acc, ctx)            -- accumulator seed (bias)
//   __device__ void finish(acc, ctx)          -- consume the finished tile
// ctx carries tile origin, wave/lane coordinates and problem sizes.
struct TileCtx {
    int m0, n0;        // tile origin (row of R, row of Cm)
    int wm, wn;        // wave coordinates in the 2x2 wave grid
    int lane_col;      // lane & 31  -> output column (Cm row) inside a 32x32 tile
    int lane_half;     // lane >> 5  -> +4 on the output row
    int M, N, K;
    int tid;
};

// Output row inside a 32x32 MFMA tile held by accumulator register `reg` of this lane.
__device__ __forceinline__ int mfma_row(int reg, int lane_half) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * lane_half;
}

template <class LA, class LB, class Epi, int BM, int BN, int BK>
__global__ void __launch_bounds__(kGemmThreads)
gemm_nt_f32_kernel(typename LA::Args la, typename LB::Args lb, typename Epi::Args ea, int M, int N,
                   int K, TileMap map) {
    using G = TileGeom<BK>;
    constexpr int WTM = BM / 2, WTN = BN / 2;      // per-wave tile
    constexpr int MT = WTM / 32, NT = WTN / 32;    // MFMA tiles per wave
    constexpr int TILE_A = BM * G::LDS_STRIDE, TILE_B = BN * G::LDS_STRIDE;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][TILE_A + TILE_B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TileCtx ctx;
    int tm, tn;
    map.locate(blockIdx.x, gridDim.x, tm, tn);
    ctx.m0 = tm * BM;
    ctx.n0 = tn * BN;
    ctx.wm = wave >> 1;
    ctx.wn = wave & 1;
    ctx.lane_col = lane & 31;
    ctx.lane_half = lane >> 5;
    ctx.M = M; ctx.N = N; ctx.K = K; ctx.tid = tid;

    LA a;
    LB b;
    a.init(la, ctx.m0, K, tid);
    b.init(lb, ctx.n0, K, tid);

    f32x16 acc[MT][NT];
    Epi epi;
    epi.init(ea, acc, ctx);

    const int nk = (K + BK - 1) / BK;
    a.load(0);
    b.load(0);
    a.store(smem);
    b.store(smem + TILE_A);
    __syncthreads();

    const int arow = (ctx.wm * WTM + ctx.lane_col) * G::LDS_STRIDE + 4 * ctx.lane_half;
    const int brow = (ctx.wn * WTN + ctx.lane_col) * G::LDS_STRIDE + 4 * ctx.lane_half;

#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        const float* sA = smem + (kt & 1) * (TILE_A + TILE_B);
        const float* sB = sA + TILE_A;
        float* nA = smem + ((kt + 1) & 1) * (TILE_A + TILE_B);
        const bool more = (kt + 1) < nk;
        if (more) {
            a.load(kt + 1);
            b.load(kt + 1);
        }
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            f32x4 af[MT], bf[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                af[mt] = *reinterpret_cast<const f32x4*>(sA + arow + mt * 32 * G::LDS_STRIDE + 8 * g);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                bf[nt] = *reinterpret_cast<const f32x4*>(sB + brow + nt * 32 * G::LDS_STRIDE + 8 * g);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][t], bf[nt][t],
                                                                           acc[mt][nt], 0, 0, 0);
        }
        if (Epi::kCheckpoints) epi.checkpoint(ea, acc, ctx, (kt + 1) * BK);
        if (more) {
            a.store(nA);
            b.store(nA + TILE_A);
        }
        __syncthreads();
    }
    epi.finish(ea, acc, ctx, smem);
}

template <int BM, int BN, int BK>
constexpr size_t gemm_lds_bytes() {
    return 2ull * (BM + BN) * (BK + 4) * sizeof(float);
}

// Host-side launcher.
template <class LA, class LB, class Epi, int BM, int BN, int BK>
inline int launch_gemm(const typename LA::Args& la, const typename LB::Args& lb,
                       const typename Epi::Args& ea, int M, int N, int K, hipStream_t stream) {
    auto kern = gemm_nt_f32_kernel<LA, LB, Epi, BM, BN, BK>;
    constexpr size_t lds = gemm_lds_bytes<BM, BN, BK>();
    static bool configured = false;   // per instantiation
    if (!configured) {
        QSAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        configured = true;
    }
    TileMap map;
    map.tiles_m = (M + BM - 1) / BM;
    map.tiles_n = (N + BN - 1) / BN;
    const long long nblocks = static_cast<long long>(map.tiles_m) * map.tiles_n;
    if (nblocks <= 0 || nblocks > 0x7FFFFFFFll) return fail(QSAE_ERR_UNSUPPORTED, "%s: tile count out of range", __func__);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(nblocks)), dim3(kGemmThreads), lds, stream, la, lb,
                       ea, M, N, K, map);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

}  // namespace qsae

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

#include <type_traits>

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
// KPERM: the operand is stored K-interleaved in memory (every 8 consecutive k as
// [k0 k2 k4 k6 k1 k3 k5 k7], qsae_kperm_rows): a loaded 16-byte chunk is then already an LDS
// chunk and goes out as ONE ds_write_b128 -- no register shuffling.
template <int ROWS, int BK, bool KTAIL = false, bool ASM = false, bool KPERM = false>
struct LoaderF32 {
    using G = TileGeom<BK>;
    static constexpr int PASSES = ROWS / G::ROWS_PER_PASS;
    static_assert(ROWS % G::ROWS_PER_PASS == 0, "tile rows must be a multiple of rows per pass");
    struct Args {
        const float* p;
        int64_t ld;
        int nrows;
    };
    // The hot (K % BK == 0) variant issues its loads from inline asm: hipcc cannot tell the two
    // staging sets apart across loop iterations and would drain the newer set with vmcnt(0);
    // the kernel waits for them itself with a counted s_waitcnt.  Opt-in (ASM) and only for
    // register-lean tiles: an asm load's destination must never be spilled or copied before the
    // wait (cdna guide 5.7 item 1), which is audited per instantiation (no scratch).  Addressing is
    // SGPR base (operand + k offset, advanced with scalar adds) + one 32-bit VGPR byte offset per
    // pass, so a K step costs no vector address arithmetic.
    static constexpr bool kAsmLoads = ASM && !KTAIL;
    static constexpr int kLoadsPerStep = PASSES;
    const float* rowp[PASSES];       // compiler-load variants
    uint32_t voff[PASSES];           // asm variant: byte offset of (row, chunk) from the operand base
    f32x4 r[2][PASSES];              // two staging sets: slices are fetched two steps ahead
    Args args;
    int K, c, lrow;

    __device__ __forceinline__ void init(const Args& a, int K_, int tid) {
        args = a;
        K = K_;
        c = tid % G::CHUNKS;
        lrow = tid / G::CHUNKS;
    }
    // point the loader at the tile whose first row is row0
    __device__ __forceinline__ void set_rows(int row0) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            int row = row0 + i * G::ROWS_PER_PASS + lrow;
            row = row < args.nrows ? row : args.nrows - 1;
            if (kAsmLoads) voff[i] = static_cast<uint32_t>((static_cast<int64_t>(row) * args.ld + 4 * c) * 4);
            else rowp[i] = args.p + static_cast<int64_t>(row) * args.ld + 4 * c;
        }
    }
    // KTAIL = false (K % BK == 0, the hot configuration): plain loads, nothing consumes them
    // until store(), so their latency hides under the MFMAs of the current slice.
    // KTAIL = true: chunks at or beyond K re-read the row start and are zeroed.
    template <int P>
    __device__ __forceinline__ void load(int kt) {
        const int k0 = kt * BK;
        if (KTAIL) {
            const bool ok = (k0 + 4 * c) < K;
            const int koff = ok ? k0 : -4 * c;
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(rowp[i] + koff);
                r[P][i] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else if (ASM) {
            const float* base = args.p + k0;      // wave-uniform: lives in an SGPR pair
#pragma unroll
            for (int i = 0; i < PASSES; ++i)
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(r[P][i]) : "v"(voff[i]), "s"(base) : "memory");
        } else {
#pragma unroll
            for (int i = 0; i < PASSES; ++i) r[P][i] = *reinterpret_cast<const f32x4*>(rowp[i] + k0);
        }
    }
    // order every register of set P behind the caller's s_waitcnt (cdna guide 5.7, form ii)
    template <int P>
    __device__ __forceinline__ void pin() {
        if (kAsmLoads) {
#pragma unroll
            for (int i = 0; i < PASSES; ++i) asm volatile("" : "+v"(r[P][i]));
        }
    }
    template <int P>
    __device__ __forceinline__ void store(float* tile) const {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            float* row = tile + (i * G::ROWS_PER_PASS + lrow) * G::LDS_STRIDE;
            if (KPERM) *reinterpret_cast<f32x4*>(row + 4 * c) = r[P][i];
            else lds_store_chunk(row, c, r[P][i]);
        }
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
    static constexpr bool kAsmLoads = false;
    static constexpr int kLoadsPerStep = PASSES;
    template <int P> __device__ __forceinline__ void pin() {}
    const uint32_t* rowp[PASSES];
    uint32_t r[2][PASSES];
    Args args;
    int K, c, lrow;

    __device__ __forceinline__ void init(const Args& a, int K_, int tid) {
        args = a;
        K = K_;
        c = tid % G::CHUNKS;
        lrow = tid / G::CHUNKS;
    }
    __device__ __forceinline__ void set_rows(int row0) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            int row = row0 + i * G::ROWS_PER_PASS + lrow;
            row = row < args.nrows ? row : args.nrows - 1;
            rowp[i] = args.p + static_cast<int64_t>(row) * args.words_ld;
        }
    }
    template <int P>
    __device__ __forceinline__ void load(int kt) {
        const int k = kt * BK + 4 * c;            // first field of this thread's chunk
        if (KTAIL) {
            const bool ok = k < K;
            const int w = ok ? (k >> 4) : 0;
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const uint32_t t = rowp[i][w];
                r[P][i] = ok ? t : 0u;
            }
        } else {
#pragma unroll
            for (int i = 0; i < PASSES; ++i) r[P][i] = rowp[i][k >> 4];
        }
    }
    template <int P>
    __device__ __forceinline__ void store(float* tile) const {
        const int sh = 2 * ((4 * c) & 15);
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int w = static_cast<int>(r[P][i] >> sh);
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
    static constexpr bool kAsmLoads = false;
    static constexpr int kLoadsPerStep = PASSES + 1;
    template <int P> __device__ __forceinline__ void pin() {}
    const uint32_t* rowp[PASSES];
    uint32_t r[2][PASSES];
    f32x4 s[2];
    const float* scale;
    Args args;
    int K, c, lrow;

    __device__ __forceinline__ void init(const Args& a, int K_, int tid) {
        args = a;
        K = K_;
        c = tid % G::CHUNKS;
        lrow = tid / G::CHUNKS;
        scale = a.scale;
    }
    __device__ __forceinline__ void set_rows(int row0) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            int row = row0 + i * G::ROWS_PER_PASS + lrow;
            row = row < args.nrows ? row : args.nrows - 1;
            rowp[i] = args.bits + static_cast<int64_t>(row) * args.words_ld;
        }
    }
    template <int P>
    __device__ __forceinline__ void load(int kt) {
        const int k = kt * BK + 4 * c;
        if (KTAIL) {
            const bool ok = k < K;
            const int kk = ok ? k : 0;
            const f32x4 t = *reinterpret_cast<const f32x4*>(scale + kk);
            s[P] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const uint32_t w = rowp[i][kk >> 5];
                r[P][i] = ok ? w : 0u;
            }
        } else {
            s[P] = *reinterpret_cast<const f32x4*>(scale + k);
#pragma unroll
            for (int i = 0; i < PASSES; ++i) r[P][i] = rowp[i][k >> 5];
        }
    }
    template <int P>
    __device__ __forceinline__ void store(float* tile) const {
        const int sh = (4 * c) & 31;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const uint32_t w = r[P][i] >> sh;
            f32x4 v;
            v[0] = (w & 1u) ? s[P][0] : 0.f;
            v[1] = (w & 2u) ? s[P][1] : 0.f;
            v[2] = (w & 4u) ? s[P][2] : 0.f;
            v[3] = (w & 8u) ? s[P][3] : 0.f;
            lds_store_chunk(tile + (i * G::ROWS_PER_PASS + lrow) * G::LDS_STRIDE, c, v);
        }
    }
};

// -------------------------------------------------------------------------------------
// Workgroup -> work mapping.  A workgroup owns ONE Cm panel (BN rows of the K-contiguous
// "column" operand) and sweeps `sweep` consecutive R tiles (BM rows each) against it, keeping
// the global->LDS pipeline running across tile boundaries (no per-tile prologue).  The
// encoder+top-k path sets sweep = all R tiles so that a workgroup sees whole latent rows.
// Blocks are dealt round-robin to the 8 XCDs (b % 8 labels the blocks that share an L2): each
// XCD gets a contiguous run of work items, ordered so that the blocks resident together on one
// XCD share Cm panels.  Bijective for any block count.  Speed only, never correctness.
struct SweepMap {
    int tiles_m, tiles_n, sweep, msplit;    // msplit = ceil(tiles_m / sweep)
    int stagger;                            // units of 1024 cycles of start delay for the 2nd co-resident block
    __device__ __forceinline__ void locate(int bid, int nblocks, int& tn, int& m_first, int& m_last) const {
        constexpr int NXCD = 8;
        const int q = nblocks / NXCD, rem = nblocks % NXCD;
        const int xcd = bid % NXCD, slot = bid / NXCD;
        const int vid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
        tn = vid / msplit;
        const int ms = vid % msplit;
        m_first = ms * sweep;
        m_last = (m_first + sweep) < tiles_m ? (m_first + sweep) : tiles_m;
    }
};

// -------------------------------------------------------------------------------------
// The kernel.  Epi provides:
//   struct Args;  static constexpr int kCheckpoints;
//   __device__ void begin(args, ctx, smem_epi)   -- once per workgroup, before the sweep
//   __device__ void init(args, acc, ctx)         -- accumulator seed for the tile at ctx.m0
//   __device__ void checkpoint(args, acc, ctx, k_done)   (only if kCheckpoints)
//   __device__ void finish(args, acc, ctx)       -- consume the finished tile
//   __device__ void end(args, ctx)               -- once per workgroup, after the sweep
//   static constexpr int kLdsFloats               -- extra LDS the epilogue wants
// ctx carries tile origin, wave/lane coordinates and problem sizes.
struct TileCtx {
    int m0, n0;        // tile origin (row of R, row of Cm)
    int wm, wn;        // wave coordinates in the 2x2 wave grid
    int lane_col;      // lane & 31  -> output column (Cm row) inside a 32x32 tile
    int lane_half;     // lane >> 5  -> +4 on the output row
    int M, N, K;
    int tid;
    int part;          // which of the map's msplit slices of the R range this workgroup sweeps (0 when it sweeps all of it)
    float* lds_epi;    // epilogue scratch (kLdsFloats floats), after the staging buffers
};

// Output row inside a 32x32 MFMA tile held by accumulator register `reg` of this lane.
__device__ __forceinline__ int mfma_row(int reg, int lane_half) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * lane_half;
}

// 128x128 tiles are sized for two workgroups per CU (LDS 2 x 74 KB, <= 256 registers per lane):
// the second launch-bounds argument (waves per SIMD) makes the register allocator honour that.
// ABLATE (diagnosis builds only, wrong results): 1 = no global loads / LDS writes inside the K loop,
// 2 = additionally no LDS fragment reads (operands stay in registers): isolates the MFMA stream.
template <class LA, class LB, class Epi, int BM, int BN, int BK, int ABLATE = 0>
__global__ void __launch_bounds__(kGemmThreads, (BM * BN <= 128 * 128) ? 2 : 1)
gemm_nt_f32_kernel(typename LA::Args la, typename LB::Args lb, typename Epi::Args ea, int M, int N,
                   int K, SweepMap map) {
    using G = TileGeom<BK>;
    constexpr int WTM = BM / 2, WTN = BN / 2;      // per-wave tile
    constexpr int MT = WTM / 32, NT = WTN / 32;    // MFMA tiles per wave
    constexpr int TILE_A = BM * G::LDS_STRIDE, TILE_B = BN * G::LDS_STRIDE;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][TILE_A + TILE_B] + epilogue

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TileCtx ctx;
    int tn, m_first, m_last;
    map.locate(blockIdx.x, gridDim.x, tn, m_first, m_last);
    ctx.n0 = tn * BN;
    ctx.wm = wave >> 1;
    ctx.wn = wave & 1;
    ctx.lane_col = lane & 31;
    ctx.lane_half = lane >> 5;
    ctx.M = M; ctx.N = N; ctx.K = K; ctx.tid = tid;
    ctx.lds_epi = smem + 2 * (TILE_A + TILE_B);
    ctx.m0 = m_first * BM;
    ctx.part = m_first / map.sweep;

    // Two workgroups share a CU and run the same program: launched together they reach their
    // barriers and LDS-write phases together and leave the matrix pipe idle together.  Blocks are
    // dealt round-robin over 8 XCDs x 32 CUs, so bit 8 of the block id tells the first from the
    // second block of a CU in the first dispatch wave; delaying the second one by about half a
    // K step de-phases the pair (cdna microarch guide, "two waves per SIMD", item 9).  Speed only.
    if (map.stagger > 0 && ((blockIdx.x >> 8) & 1)) {
        for (int i = 0; i < map.stagger; ++i) __builtin_amdgcn_s_sleep(16);
    }

    LA a;
    LB b;
    a.init(la, K, tid);
    b.init(lb, K, tid);
    // diagnosis (stagger == -1/-2, wrong results): every workgroup streams the SAME operand rows, i.e.
    // all staging loads hit L2 -- separates "memory supply" from "instruction stream" losses
    const bool same_a = map.stagger < 0, same_b = map.stagger == -1;
    a.set_rows(same_a ? 0 : ctx.m0);
    b.set_rows(same_b ? 0 : ctx.n0);

    Epi epi;
    epi.begin(ea, ctx);

    // Flat pipeline over steps s = (tile, kt): the LDS image of step s lives in buffer s & 1, the
    // global loads of step s + 2 are issued at the top of step s into staging set s & 1 and are
    // written to LDS at the end of step s + 1 -- two full MFMA phases to cover the L2/HBM latency.
    const int nk = (K + BK - 1) / BK;
    const int nsteps = (m_last - m_first) * nk;
    int ld_tile = m_first, ld_kt = 0;           // position of the load stream
    auto issue_loads = [&](auto pc) {
        constexpr int P = decltype(pc)::value;
        a.template load<P>(ld_kt);
        b.template load<P>(ld_kt);
        if (++ld_kt == nk) {
            ld_kt = 0;
            ++ld_tile;
            a.set_rows(same_a ? 0 : ld_tile * BM);
        }
    };
    issue_loads(std::integral_constant<int, 0>{});
    if (nsteps > 1) issue_loads(std::integral_constant<int, 1>{});
    if (LA::kAsmLoads || LB::kAsmLoads) {
        constexpr int kYounger = LA::kLoadsPerStep + LB::kLoadsPerStep;
        if (nsteps > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kYounger) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        a.template pin<0>();
        b.template pin<0>();
    }
    a.template store<0>(smem);
    b.template store<0>(smem + TILE_A);
    __syncthreads();

    const int arow = (ctx.wm * WTM + ctx.lane_col) * G::LDS_STRIDE + 4 * ctx.lane_half;
    const int brow = (ctx.wn * WTN + ctx.lane_col) * G::LDS_STRIDE + 4 * ctx.lane_half;
    int tile = m_first, kt = 0;
    f32x16 acc[MT][NT];
    constexpr int kStoreAfterG = (BK / 8 >= 2) ? (BK / 8) / 2 - 1 : 0;   // LDS writes after the first half

    auto step = [&](auto pc, int s_idx) {
        constexpr int P = decltype(pc)::value;      // parity of this step: LDS buffer and staging set
        const float* sA = smem + P * (TILE_A + TILE_B);
        const float* sB = sA + TILE_A;
        float* nA = smem + (P ^ 1) * (TILE_A + TILE_B);
        if (kt == 0) {
            ctx.m0 = tile * BM;
            epi.init(ea, acc, ctx);
        }
        // staging set P held step s (already in LDS): refill it with step s + 2
        if (ABLATE == 0 && s_idx + 2 < nsteps) issue_loads(pc);
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            f32x4 af[MT], bf[NT];
            if (ABLATE >= 2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) { af[mt] = f32x4{1.f, 2.f, 3.f, 4.f}; asm volatile("" : "+v"(af[mt])); }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { bf[nt] = f32x4{1.f, 2.f, 3.f, 4.f}; asm volatile("" : "+v"(bf[nt])); }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    af[mt] = *reinterpret_cast<const f32x4*>(sA + arow + mt * 32 * G::LDS_STRIDE + 8 * g);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bf[nt] = *reinterpret_cast<const f32x4*>(sB + brow + nt * 32 * G::LDS_STRIDE + 8 * g);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][t], bf[nt][t], acc[mt][nt], 0, 0, 0);
            // Mid-step: step s + 1 (fetched during step s - 1) goes to the other LDS buffer.  Nobody
            // reads that buffer during step s (its last readers passed the barrier that ended step
            // s - 1), so the LDS writes ride under this wave's remaining MFMAs instead of forming a
            // serial tail in front of the barrier.
            if (g == kStoreAfterG && ABLATE == 0 && s_idx + 1 < nsteps) {
                __builtin_amdgcn_sched_barrier(0);
                if (LA::kAsmLoads || LB::kAsmLoads) {
                    // Every VMEM op younger than set P^1's loads may stay in flight: the loads of set
                    // P issued at the top of this step (if any) -- anything else (epilogue stores, bias
                    // loads) only makes the wait stricter than necessary, never looser.
                    constexpr int kYounger = LA::kLoadsPerStep + LB::kLoadsPerStep;
                    // first step after an epilogue: its fire-and-forget stores sit between the two sets
                    constexpr int kAfterEpi = (kYounger + Epi::kStoresPerFinish) < 63 ? (kYounger + Epi::kStoresPerFinish) : 63;
                    if (s_idx + 2 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (kt == 0 && Epi::kStoresPerFinish > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kAfterEpi) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kYounger) : "memory");
                    a.template pin<P ^ 1>();
                    b.template pin<P ^ 1>();
                }
                a.template store<P ^ 1>(nA);
                b.template store<P ^ 1>(nA + TILE_A);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (Epi::kCheckpoints) epi.checkpoint(ea, acc, ctx, (kt + 1) * BK);
        __syncthreads();
        if (++kt == nk) {
            epi.finish(ea, acc, ctx);
            kt = 0;
            ++tile;
        }
    };
#pragma unroll 1
    for (int s_idx = 0; s_idx < nsteps; s_idx += 2) {
        step(std::integral_constant<int, 0>{}, s_idx);
        if (s_idx + 1 < nsteps) step(std::integral_constant<int, 1>{}, s_idx + 1);
    }
    epi.end(ea, ctx);
}

// Sweep length heuristic for kernels without row ownership: long enough to amortise the pipeline
// fill, short enough to leave >= ~2048 workgroups for load balance.  g_sweep_override > 0 forces it.
#ifdef QSAE_DEBUG_BUILD
extern int g_sweep_override;
extern int g_stagger;
#else
constexpr int g_sweep_override = 0;
constexpr int g_stagger = 0;
#endif
template <int BM, int BN>
inline int pick_sweep(int M, int N, int /*K*/) {
    if (g_sweep_override > 0) return g_sweep_override;
    const long long tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    long long s = tiles_m * tiles_n / 2048;
    if (s < 1) s = 1;
    if (s > 16) s = 16;
    return static_cast<int>(s);
}

template <int BM, int BN, int BK>
constexpr size_t gemm_lds_bytes(int epi_floats) {
    return (2ull * (BM + BN) * (BK + 4) + static_cast<size_t>(epi_floats)) * sizeof(float);
}

// Host-side launcher.  sweep = number of consecutive R tiles per workgroup (<= 0: all of them).
template <class LA, class LB, class Epi, int BM, int BN, int BK, int ABLATE = 0>
inline int launch_gemm(const typename LA::Args& la, const typename LB::Args& lb,
                       const typename Epi::Args& ea, int M, int N, int K, int sweep, hipStream_t stream) {
    auto kern = gemm_nt_f32_kernel<LA, LB, Epi, BM, BN, BK, ABLATE>;
    constexpr size_t lds = gemm_lds_bytes<BM, BN, BK>(Epi::kLdsFloats);
    QSAE_SET_MAX_LDS_ONCE(kern, lds);     // per instantiation and device
    SweepMap map;
    map.tiles_m = (M + BM - 1) / BM;
    map.tiles_n = (N + BN - 1) / BN;
    map.sweep = (sweep <= 0 || sweep > map.tiles_m) ? map.tiles_m : sweep;
    map.msplit = (map.tiles_m + map.sweep - 1) / map.sweep;
    map.stagger = g_stagger;
    const long long nblocks = static_cast<long long>(map.tiles_n) * map.msplit;
    if (nblocks <= 0 || nblocks > 0x7FFFFFFFll) return fail(QSAE_ERR_UNSUPPORTED, "%s: tile count out of range", __func__);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(nblocks)), dim3(kGemmThreads), lds, stream, la, lb,
                       ea, M, N, K, map);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

}  // namespace qsae

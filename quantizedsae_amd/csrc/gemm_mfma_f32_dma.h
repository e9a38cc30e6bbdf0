// gemm_mfma_f32_dma.h -- the exact-fp32 MFMA contraction with LDS-DMA staging.
//
// Same arithmetic as gemm_mfma_f32.h (v_mfma_f32_32x32x2_f32 walked in ascending k: bit-for-bit the
// oracle's fmaf chain), different data movement: operands that are stored K-interleaved in memory
// (qsae_kperm_rows) need no register hop on their way to LDS, so the staging is done by
// `global_load_lds_dwordx4` (LDS-DMA): no staging VGPRs, no ds_write instructions, no register
// shuffles -- the instruction stream of a K step is fragment reads + MFMAs + 6 DMA issues per wave.
//
//   * workgroup: 512 threads = 8 waves as 4 (R) x 2 (Cm), each wave a 64 x 64 block (2 x 2 MFMA
//     tiles), tile BM x BN = 256 x 128, BK = 32: 42.7 FLOP per staged byte (128 x 128: 32).
//   * LDS: 3 stages x (256 + 128) rows x 128 B = 144 KiB, one workgroup per CU, two waves per SIMD.
//     DMA for step s + 2 is issued at the top of step s; a counted s_waitcnt vmcnt(6) + one raw
//     s_barrier per step retire the stage needed next (cdna guide: "read a staged buffer one phase
//     after the wait that retires it") and protect the stage about to be overwritten.
//   * LDS image: row r = 32 floats (128 B, unpadded -- an LDS-DMA instruction writes 1 KiB
//     lane-linearly); the 16-byte chunk j of row r sits at position j ^ (r & 7).  The XOR is applied
//     on the per-lane SOURCE address of the DMA and again on the fragment read (cdna guide rule 21),
//     which leaves the ds_read_b128 fragment reads 2-way conflicted instead of 8-way.
//
// HALF = true instantiates the same pipeline over fp16 operands ([rows][K] _Float16, natural k order)
// with v_mfma_f32_32x32x16_f16: a K step is still 128 bytes per row (64 k), the image and the swizzle
// are unchanged, one 16-byte fragment read feeds one MFMA instead of four.  It is used only as the
// order-preserving *prefilter* of the encoder+top-k path (encode_topk.hip), never for returned values.
#pragma once

#include "gemm_mfma_f32.h"

namespace qsae {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kDmaThreads = 512;
constexpr int kDmaBK = 32;          // fp32 elements per step (128 bytes per row); fp16: 64 elements
constexpr int kDmaStages = 3;         // default; a 256 x 256 tile fits only two stages (1-deep prefetch)

template <class Epi, int BM, int BN, bool HALF = false, int STAGES = kDmaStages>
__global__ void __launch_bounds__(kDmaThreads, 2)
gemm_nt_f32_dma_kernel(const float* __restrict__ Rp, int M, const float* __restrict__ Cp, int N, int K,
                       typename Epi::Args ea, SweepMap map) {
    // K and the operand pointers are expressed in 4-byte words: an fp16 operand with K16 elements per row
    // is passed as K = K16 / 2 words per row (so that all addressing below is type-agnostic).
    constexpr int WMW = 4, WNW = 2;                      // wave grid
    constexpr int WTM = BM / WMW, WTN = BN / WNW;        // 64 x 64 per wave
    constexpr int MT = WTM / 32, NT = WTN / 32;
    constexpr int ROWS = BM + BN;
    constexpr int STAGE = ROWS * kDmaBK;                 // floats per stage
    constexpr int GROUPS = ROWS / 8;                     // 1 KiB DMA pieces (8 rows) per stage
    constexpr int PER_WAVE = GROUPS / 8;                 // pieces per wave per step
    constexpr int A_PIECES = (BM / 8) / 8;               // of which the first A_PIECES belong to R
    static_assert(GROUPS % 8 == 0 && (BM / 8) % 8 == 0, "tile rows must split evenly over 8 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [3][ROWS][32] + epilogue scratch

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
    ctx.lds_epi = smem + STAGES * STAGE;
    ctx.m0 = m_first * BM;
    ctx.part = m_first / map.sweep;

    Epi epi;
    epi.begin(ea, ctx);

    // ---- DMA source addressing: piece p of this wave covers tile rows 8*(wave + 8p) .. +7 ----------
    const int sub_row = lane >> 3;                       // row inside the 8-row piece
    const int src_chunk = (lane & 7) ^ sub_row;          // XOR swizzle on the source (row & 7 == sub_row)
    const float* pb[PER_WAVE - A_PIECES];                // Cm rows never change during the sweep
#pragma unroll
    for (int p = A_PIECES; p < PER_WAVE; ++p) {
        int row = ctx.n0 + 8 * (wave + 8 * p) - BM + sub_row;
        row = row < N ? row : N - 1;
        pb[p - A_PIECES] = Cp + static_cast<int64_t>(row) * K + 4 * src_chunk;
    }
    const float* pa[A_PIECES];
    auto set_a_rows = [&](int m0) {
#pragma unroll
        for (int p = 0; p < A_PIECES; ++p) {
            int row = m0 + 8 * (wave + 8 * p) + sub_row;
            row = row < M ? row : M - 1;
            pa[p] = Rp + static_cast<int64_t>(row) * K + 4 * src_chunk;
        }
    };
    set_a_rows(ctx.m0);

    const int nk = K / kDmaBK;
    const int nsteps = (m_last - m_first) * nk;
    int ld_tile = m_first, ld_kt = 0, ld_stage = 0;
    auto issue_dma = [&]() {
        float* stage = smem + ld_stage * STAGE;
        const int k0 = ld_kt * kDmaBK;
#pragma unroll
        for (int p = 0; p < PER_WAVE; ++p) {
            const float* src = (p < A_PIECES ? pa[p] : pb[p - A_PIECES]) + k0;
            float* dst = stage + (wave + 8 * p) * (8 * kDmaBK);      // wave-uniform; lanes land at +16 B each
            typedef const __attribute__((address_space(1))) void* gptr_t;
            typedef __attribute__((address_space(3))) void* lptr_t;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        }
        ld_stage = (ld_stage + 1 == STAGES) ? 0 : ld_stage + 1;
        if (++ld_kt == nk) {
            ld_kt = 0;
            ++ld_tile;
            set_a_rows(ld_tile * BM);
        }
    };
    // prefetch distance AHEAD = STAGES - 1 steps: 3 stages -> DMA two steps ahead, 2 stages -> one step ahead
    constexpr int AHEAD = STAGES - 1;
    issue_dma();
    if (AHEAD == 2 && nsteps > 1) {
        issue_dma();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    // ---- fragment read addressing -----------------------------------------------------------------
    const int swz = ctx.lane_col & 7;
    int choff[kDmaBK / 8];                               // float offset of chunk (2g + half) after the XOR
#pragma unroll
    for (int g = 0; g < kDmaBK / 8; ++g) choff[g] = 4 * ((2 * g + ctx.lane_half) ^ swz);
    const int arow = (ctx.wm * WTM + ctx.lane_col) * kDmaBK;
    const int brow = (BM + ctx.wn * WTN + ctx.lane_col) * kDmaBK;

    int tile = m_first, kt = 0, st = 0;
    f32x16 acc[MT][NT];
#pragma unroll 1
    for (int s_idx = 0; s_idx < nsteps; ++s_idx) {
        const float* sbase = smem + st * STAGE;
        if (kt == 0) {
            ctx.m0 = tile * BM;
            epi.init(ea, acc, ctx);
        }
        const bool prefetch = (s_idx + AHEAD) < nsteps;
        if (prefetch) issue_dma();                        // step s+AHEAD -> the stage last read in step s-1
#pragma unroll
        for (int g = 0; g < kDmaBK / 8; ++g) {
            f32x4 af[MT], bf[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                af[mt] = *reinterpret_cast<const f32x4*>(sbase + arow + mt * 32 * kDmaBK + choff[g]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                bf[nt] = *reinterpret_cast<const f32x4*>(sbase + brow + nt * 32 * kDmaBK + choff[g]);
            if (HALF) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(f16x8, af[mt]), __builtin_bit_cast(f16x8, bf[nt]), acc[mt][nt], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][t], bf[nt][t], acc[mt][nt], 0, 0, 0);
            }
        }
        // retire the stage of step s+1 (its DMA was issued during step s-1): everything but the pieces
        // issued at the top of this step must have landed, for every wave, before anyone reads it
        if (prefetch && AHEAD == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        st = (st + 1 == STAGES) ? 0 : st + 1;
        if (++kt == nk) {
            epi.finish(ea, acc, ctx);
            kt = 0;
            ++tile;
        }
    }
    epi.end(ea, ctx);
}

// K is in 4-byte words per row (fp16 operands: elements / 2).  sweep <= 0: a workgroup sweeps every R tile.
template <class Epi, int BM, int BN, bool HALF = false, int STAGES = kDmaStages>
inline int launch_gemm_dma(const float* Rp, int M, const float* Cp, int N, int K, const typename Epi::Args& ea,
                           hipStream_t stream, int sweep = 0) {
    auto kern = gemm_nt_f32_dma_kernel<Epi, BM, BN, HALF, STAGES>;
    constexpr size_t lds = (static_cast<size_t>(STAGES) * (BM + BN) * kDmaBK + Epi::kLdsFloats) * sizeof(float);
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    QSAE_SET_MAX_LDS_ONCE(kern, lds);     // per instantiation and device
    if (K % kDmaBK != 0) return fail(QSAE_ERR_UNSUPPORTED, "%s: K must be a multiple of 32 words", __func__);
    SweepMap map;
    map.tiles_m = (M + BM - 1) / BM;
    map.tiles_n = (N + BN - 1) / BN;
    map.sweep = (sweep <= 0 || sweep > map.tiles_m) ? map.tiles_m : sweep;
    map.msplit = (map.tiles_m + map.sweep - 1) / map.sweep;
    map.stagger = 0;
    const long long nblocks = static_cast<long long>(map.tiles_n) * map.msplit;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(nblocks)), dim3(kDmaThreads), lds, stream, Rp, M, Cp, N,
                       K, ea, map);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

}  // namespace qsae

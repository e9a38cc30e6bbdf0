// sweep_xstat_f16.h -- the fp16 candidate sweep with the activation panel stationary in registers.
//
// Role: the candidate pass of the prefilter pipeline (encode_topk.hip).  For every activation row b and every
// hidden unit h it forms the approximate latent  v = bias[h] + inv[b] sum_k xq[b,k] wq[h,k]  with
// v_mfma_f32_32x32x16_f16 (the chain of a row tile starts from bias / inv, so the accumulators are v / inv and are
// compared with (tau[b] - margin[b]) / inv directly) and appends (v, h) to the row's candidate list when
// !(v < tau[b] - margin[b]).
// It returns no values of its own: the refine step recomputes every survivor with the exact fp32 chain.
// The same launch (i) derives tau from a pilot pass over a stratified H/16 sample of the hidden units
// (pilot_stages > 0), (ii) writes the zeros of the dense [B, H] latent, whose HBM traffic it hides.
//
// Data movement (the generic LDS-DMA GEMM stages BOTH operands for every tile and is bound by ~20 GB/s/CU of
// DMA, 3.0 ms on the headline shape):
//   * a workgroup owns 256 activation rows for the whole launch: wave w keeps the B-operand fragments of
//     rows 32w..32w+31 for all of K in registers (D = 512: 32 k-blocks x 4 VGPRs = 128 VGPRs), loaded once;
//   * only the fp16 weights stream: 64 hidden rows (D*2 bytes each) per iteration, 2 buffers in LDS, written by
//     global_load_lds_dwordx4 one iteration ahead, one 1-KiB piece between two MFMA groups; every workgroup
//     streams the same rows in the same order, so after the first toucher in an XCD the stream is L2 hits;
//   * staged bytes per MFMA FLOP are 1/256 B (256 x 256 tile: 1/128, 256 x 128: 1/85).
// LDS image of a buffer: row r = CPR 16-byte chunks; chunk j sits at position j ^ (r & 15), applied on the
// DMA source address and on the fragment read: the four 16-lane groups of a ds_read_b128 (MI355X guide,
// LDS table) each see 16 rows that are distinct mod 16, i.e. 16 distinct bank groups.
// One iteration (in-sweep-fill build): flush of the previous records -> [32 MFMAs of row tile 0 | DMA pieces of the
// next block | filter of the previous row tile 1] -> [32 MFMAs of row tile 1 | fill stores | filter of row tile 0] ->
// s_waitcnt vmcnt(#fill stores) -> s_barrier.  vmcnt retires loads, stores and LDS-DMA in issue order, and
// the fill stores are the youngest operations of an iteration: the wait retires the DMA without waiting for
// the stores.  Nothing else uses the vector memory counter inside the loops: the block's bias is DMA'd into
// LDS with the weights, hits collect in per-lane LDS record slots and are flushed (list lengths in registers)
// at the top of the next iteration, before its DMA.
// The build without fill code (ABL 9, the product path at D = 512: the zeros come from a co-resident kernel) gives
// the two waves of a SIMD different jobs: waves 0-3 flush at the top, issue ALL DMA pieces of the next block during
// their first pass (two per MFMA group); waves 4-7 issue none and flush between their two passes -- see `iteration`.
#pragma once

#include "gemm_mfma_f32_dma.h"
#include "prefilter_common.h"

namespace qsae {

constexpr int kXsWaves = 8;
constexpr int kXsRows = 32 * kXsWaves;     // activation rows per workgroup
constexpr int kXsHT = 64;                  // hidden rows per stage (two MFMA row tiles)
constexpr int kXsStages = 2;
constexpr int kXsRingSlots = 14;           // 4-byte record slots per lane ([slot][lane] image, 256 B per slot and wave)
constexpr int kXsClamp = 10;               // a lane's write position is clamped to this slot after every 4 filtered values
constexpr int kXsSlots = kXsClamp - 1;     // records a lane can hold between flushes (position kXsClamp = "overflowed")

struct XsArgs {
    const _Float16* __restrict__ xq;      // [B][D] scaled fp16 activations
    const _Float16* __restrict__ wq;      // [Hs][D] fp16 weights of the swept hidden units
    const float* __restrict__ bias;       // [Hs] or nullptr
    const float* __restrict__ tau;        // [B] (read only when pilot_stages == 0)
    const float* __restrict__ margin;     // [B]
    const float* __restrict__ inv;        // [B]
    uint2* __restrict__ cand;             // [B][cap]
    int* __restrict__ cnt;                // [B] in: seeds already present, out: total
    int B, Hs, cap, hidden_offset;
    int rot_mul;             // row rotation of the DMA order per workgroup (see issue())
    unsigned long long* stamps;   // ABL == 5: per-wave phase cycle totals [wg][wave][8]
    float* dense;            // optional [B][dense_ld] dense latent: zero-filled here, under the MFMA-bound sweep
    long long dense_ld;
    int H;                   // columns of the dense latent (all of them are filled, not only the swept ones)
    int fill_cw;             // 1-KiB pieces (256 columns of one row) a wave fills per stage
    int pilot_stages;        // > 0: the kernel derives tau itself from pilot_stages evenly spaced 64-unit blocks (see below)
    int pilot_rank;          //   tau = pilot_rank-th largest of the row's 32 group maxima
    float* tau_out;          //   [B] tau, for the refine step's validity check
    const float* x32;        // non-null: fp32 activations [B][D]; the kernel scales / converts them itself (no xq) and
    const float* meta;       //   writes inv / margin for the refine step; meta = {s_w, max ||W_h||, max |bias|}
    float* inv_out;
    float* margin_out;
    int parts;               // hidden-range split (gridDim.y): part p sweeps stages [p*n/parts, (p+1)*n/parts),
    int* cnt_parts;          //   appends to the row's list segment [p*cap/parts, ...) and counts in cnt_parts[(p-1)*B + row]
};

template <int KB, int ABL = 0>   // k-blocks of 16 halves: D = 16 * KB; ABL: timing ablations (results wrong)
__global__ void __launch_bounds__(64 * kXsWaves)
sweep_xstat_f16_kernel(XsArgs a) {
    constexpr int D = 16 * KB;
    constexpr int CPR = 2 * KB;                              // 16-byte chunks per row
    constexpr int SW = (CPR < 16 ? CPR : 16) - 1;
    constexpr int STAGE_BYTES = kXsHT * D * 2;
    constexpr int IPW = (kXsHT * CPR / 64) / kXsWaves;       // DMA instructions per wave per stage
    constexpr int MT = kXsHT / 32;
    constexpr int BIAS_BYTES = kXsHT * 4;                    // one stage's bias; three copies in rotation
    constexpr int BIAS_BASE = kXsStages * STAGE_BYTES;
    constexpr int RING_BASE = BIAS_BASE + 3 * BIAS_BYTES;
    constexpr int RING_WAVE = kXsRingSlots * 256;            // [slot][lane] records of 4 bytes
    static_assert(kXsClamp + 4 <= kXsRingSlots, "four unclamped writes past the clamp position must stay inside the ring");
    constexpr int LDS_END = RING_BASE + kXsWaves * RING_WAVE;
    static_assert(IPW >= 1 && (kXsHT * CPR) % (64 * kXsWaves) == 0, "stage must split evenly over the waves");
    static_assert(LDS_END <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char xs_smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform for the compiler, too
    const int lane_col = lane & 31, lane_half = lane >> 5;
    const int row = blockIdx.x * kXsRows + wave * 32 + lane_col;
    const bool row_ok = row < a.B;
    const int crow = row_ok ? row : a.B - 1;

    float margin_row, inv;
    // ---- stationary operand: this lane's 8 halves of every k-block of its activation row -----------
    f16x8 xf[KB];
    if (a.x32) {
        // fused preparation: the row's max |x| and sum x^2 (this lane holds half of the row, the partner lane the
        // other half), its power-of-two scale, error budget, and the fp16 fragments straight into registers
        const float* xr = a.x32 + static_cast<int64_t>(crow) * D + 8 * lane_half;
        float mx = 0.f, ss = 0.f;
#pragma unroll 4
        for (int kb = 0; kb < KB; ++kb) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(xr + 16 * kb), p1 = *reinterpret_cast<const f32x4*>(xr + 16 * kb + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = e < 4 ? p0[e] : p1[e - 4];
                const float av = fabsf(v);
                mx = (av > mx || av != av) ? av : mx;       // NaN propagates
                ss = fmaf(v, v, ss);
            }
        }
        {
            const auto sm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            const float mo = __uint_as_float(lane_half ? sm[0] : sm[1]);
            mx = (mo > mx || mo != mo) ? mo : mx;
            const auto sq = __builtin_amdgcn_permlane32_swap(__float_as_uint(ss), __float_as_uint(ss), false, false);
            const float so = __uint_as_float(lane_half ? sq[0] : sq[1]);
            ss = lane_half ? so + ss : ss + so;              // same operand order in both lanes: identical sums
        }
        float sx;
        pref_row_params(mx, ss, -1.0f, D, a.meta[0], a.meta[1], a.meta[2], a.meta[3], sx, inv, margin_row);   // (row error not measured here: worst case)
        if (blockIdx.y == 0 && row_ok && lane_half == 0) {
            a.inv_out[row] = inv;
            a.margin_out[row] = margin_row;
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(xr + 16 * kb), p1 = *reinterpret_cast<const f32x4*>(xr + 16 * kb + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) xf[kb][e] = static_cast<_Float16>((e < 4 ? p0[e] : p1[e - 4]) * sx);
        }
    } else {
        margin_row = a.margin[crow];
        inv = a.inv[crow];
        const _Float16* xr = a.xq + static_cast<int64_t>(crow) * D + 8 * lane_half;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) xf[kb] = *reinterpret_cast<const f16x8*>(xr + 16 * kb);
    }
    const float thr = a.pilot_stages > 0 ? 0.0f : a.tau[crow] - margin_row;      // in-kernel pilot: set after the pilot pass

    // ---- DMA addressing: a piece = 64 consecutive 16-byte chunks (1 KiB) of the stage ---------------------
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // Workgroups of one XCD stream the same rows in lockstep; started on the same row they would all queue
    // on the same L2 channel at the same moment.  Each workgroup therefore walks the 1-KiB pieces of a
    // stage in its own rotation (pieces = 64-chunk groups; the LDS image is unchanged).
    constexpr int PIECES = kXsHT * CPR / 64;
    const int rot = ((blockIdx.x >> 3) * a.rot_mul) % PIECES;
    char* bias_lds = xs_smem + BIAS_BASE;
    char* my_ring = xs_smem + RING_BASE + wave * RING_WAVE + lane * 4;     // this lane's slot 0
    // no bias: the three copies stay zero.  With a bias, copies 1 and 2 start as zeros too (copy 0 arrives with the first
    // block): the first sweep iteration filters a row tile of -inf against the copy "of the iteration before", which
    // has to hold finite numbers even when no iteration has written it yet.
    if (tid < 3 * kXsHT && (!a.bias || tid >= kXsHT)) reinterpret_cast<float*>(bias_lds)[tid] = 0.f;
    // Small batches have fewer 256-row panels than the chip has CUs: the hidden range is then split over
    // gridDim.y parts, each with its own segment of every row's candidate list and its own counter.
    const int part = blockIdx.y;
    const int all_stages = a.Hs / kXsHT;
    const int s_begin = static_cast<int>(static_cast<long long>(part) * all_stages / a.parts);
    const int nstages = static_cast<int>(static_cast<long long>(part + 1) * all_stages / a.parts);   // end (absolute)
    const int cap_part = a.cap / a.parts;
    int* my_cnt = part == 0 ? a.cnt : a.cnt_parts + static_cast<size_t>(part - 1) * a.B;
    // Iterations and blocks.  A part runs pilot_stages pilot iterations over evenly spaced hidden blocks (64 hidden
    // units each), then its share [s_begin, nstages) of all blocks; iteration `it` uses LDS buffer it % 2 and bias copy
    // it % 3, block(it) gives the hidden block it works on.
    const int ps = a.pilot_stages;
    const int n_iter = ps + (nstages - s_begin);
    // the pilot blocks are spread evenly over the hidden range (a stratified sample: robust against checkpoints
    // whose hidden units are ordered, e.g. dead units first)
    const int pilot_stride = ps > 0 ? all_stages / ps : 1;
    auto block_of = [&](int it) { return it < ps ? it * pilot_stride : s_begin + (it - ps); };
    int ld = 0;                                              // next iteration to issue
    // One 1-KiB piece of iteration `ld` per call (i is a compile-time constant at every call site); wave 0 adds the
    // block's bias (kXsHT floats, lanes 0..kXsHT/4-1, 16 bytes each) to piece 0.  Three bias copies in rotation:
    // while iteration it+1 lands, the filter still reads copy it and copy it-1.
    // Where iteration `ld` comes from and goes to: computed once per iteration (next_block), in SGPRs -- worked out per
    // piece (a compare and two branches for block_of, 64-bit address arithmetic) it cost a dozen scalar instructions
    // and three branches in front of every one of the eight DMA issues of a stage.  Past the last iteration the last
    // block is fetched once more (into the buffer nobody reads any more): no branch around the issues either.
    const char* nx_src = nullptr;                            // block base, opaque scalar pair: SGPR base + 32-bit lane offset
    const char* nx_bias = nullptr;                           // addressing (otherwise the compiler hoists `wq + voff[i]`
    char* nx_dst = nullptr;                                  // as eight 64-bit VGPR pairs out of the loops and spills them)
    char* nx_bias_dst = nullptr;
    auto scalar_ptr = [](const void* p) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(u));
        const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(u >> 32));
        return reinterpret_cast<const char*>((static_cast<unsigned long long>(hi) << 32) | lo);
    };
    auto next_block = [&]() {
        const int blk = block_of(ld < n_iter ? ld : n_iter - 1);
        nx_src = scalar_ptr(reinterpret_cast<const char*>(a.wq) + static_cast<int64_t>(blk) * STAGE_BYTES);
        nx_dst = xs_smem + (ld % kXsStages) * STAGE_BYTES;
        if (a.bias) {
            nx_bias = scalar_ptr(a.bias + static_cast<int64_t>(blk) * kXsHT);
            nx_bias_dst = bias_lds + (ld % 3) * BIAS_BYTES;
        }
    };
    // Waves 0-3 issue all the DMA pieces of a stage (two per MFMA group of their first row-tile pass), waves 4-7 none:
    // the younger wave of a SIMD (w + 4) loses the issue arbitration against its partner all stage long and is the one
    // everybody waits for at the barrier (waves 0-3 idle ~1000 cycles there); the ~80 cycles an LDS-DMA issue costs are
    // better spent by the wave that has them to spare (sweep + fill 2.40 -> 2.35 ms).  Lane offsets on the fly (an
    // XOR and a shift-add per piece) instead of a register per piece.
    // (No-fill builds only: with the fill inside the sweep the extra registers of two issues per group spill.)
    constexpr bool DMAE = ABL == 9 || ABL == 10;
    constexpr int IPD = DMAE ? 2 * IPW : IPW;                // pieces per issuing wave
    const bool dma_wave = !DMAE || wave < kXsWaves / 2;      // wave-uniform
    auto issue_piece = [&](int i) {
        if (!dma_wave) return;
        const int pc = (wave * IPD + i + rot) % PIECES;      // scalar
        unsigned vo;
        if (CPR == 64) {                                      // one piece = one hidden row: row scalar, chunk = lane
            vo = (static_cast<unsigned>(lane ^ (pc & SW)) << 4) + static_cast<unsigned>(pc * 1024);
        } else {
            const int c = pc * 64 + lane;
            const int r = c / CPR, pos = c % CPR;
            vo = static_cast<unsigned>((r * CPR + (pos ^ (r & SW))) * 16);
        }
        asm volatile("" : "+v"(vo));
        __builtin_amdgcn_global_load_lds((gptr_t)(nx_src + vo), (lptr_t)(nx_dst + pc * 1024), 16, 0, 0);
        if (i == 0 && a.bias && wave == 0 && lane < kXsHT / 4)
            __builtin_amdgcn_global_load_lds((gptr_t)(nx_bias + static_cast<unsigned>(16 * lane)), (lptr_t)nx_bias_dst, 16, 0, 0);
    };
    auto issue = [&]() {
#pragma unroll
        for (int i = 0; i < IPD; ++i) issue_piece(i);
        ++ld;
    };

    if (n_iter > 0) {
        next_block();
        issue();
    }
    // Make the compiler retire its own loads (x fragments, threshold, scale) HERE: it cannot see the asm
    // waits below, and a load still pending in its model at the loop header costs a vmcnt(0) per stage.
    float thr_r = thr, inv_r = inv;
    asm volatile("" : "+v"(thr_r), "+v"(inv_r));
    // The sweep compares in the accumulator's own scale: the MFMA chain of a row tile starts from bias * rinv instead of
    // zero (rinv = s_x s_w, a power of two: exact), so an accumulator IS (latent * rinv) and the filter needs no v_fma
    // per value -- it compares with thr * rinv, and a record is scaled back by inv (exact) when it is flushed.  The
    // roundings of the chain now happen at the bias' magnitude, too: eps_b has a term for that (prefilter_common.h).
    const float rinv_r = inv_r > 0.f ? 1.0f / inv_r : 0.f;   // (inv = 0: row not servable, flagged whatever it lists)
    float thr_s = thr_r * rinv_r;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) asm volatile("" : "+v"(xf[kb]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- fragment read addressing: chunk (2kb + half) of row lane_col, kb = 8g + j -------------------
    int off[8];
    {
        const int q = (lane_col & SW) ^ lane_half;
#pragma unroll
        for (int j = 0; j < 8; ++j) off[j] = lane_col * (CPR * 16) + 16 * (((2 * j) & SW) ^ q) + 16 * ((2 * j) & ~SW);
    }

    // ---- filter + candidate records ------------------------------------------------------------------
    // Measured with s_memtime stamps (tools/prof_xstat_phases.py): a filter made of per-value wave
    // ballots and scalar branches takes ~5000 cycles per stage -- the VALU -> SALU -> branch round trip of
    // every value is exposed -- more than the stage's MFMAs (2900).  This one is straight-line: per value
    // a v_fma, the record's tag and slot address, then v_cmpx (EXEC = the lanes at or above their
    // threshold), a ds_write_b64 of {v, h | row << 27} into the lane's own record slots, EXEC restored and
    // an add-with-carry on the lane's record count.  No branch, no LDS traffic without a hit, and no
    // vmcnt(0) in front of the LDS write (the compiler would put one there: it cannot know that the slots
    // never alias the DMA'd stages).  [slot][lane] image, slot = min(count, kXsSlots): the last slot only
    // absorbs overflow.  The pilot threshold lets ~300 values per row through, ~0.3 per lane and stage.
    // The records move to the global lists at the top of the next stage, before that stage's DMA is
    // issued (the vmcnt(0) at the end of a stage then never waits for the acknowledgement of a store).
    // A row belongs to lanes l and l+32 of one wave: its list length lives in a register of both, the
    // partner's record count comes over a ds_bpermute, the lower lane appends first.  A lane that
    // overflowed its slots pushes the row's length past cap: the row takes the exact fallback like any
    // other overflowing row.
    f32x16 acc[MT];
    int count = (row_ok && part == 0 && ps == 0) ? a.cnt[crow] : 0;   // this part's segment length (equal in both lanes)
    uint2* list = a.cand + static_cast<int64_t>(crow) * a.cap + part * cap_part;
    typedef __attribute__((address_space(3))) char* lds_char_t;
    const unsigned ring_addr = static_cast<unsigned>(reinterpret_cast<size_t>((lds_char_t)my_ring));   // LDS byte address
    const unsigned ring_limit = ring_addr + kXsClamp * 256u;
    unsigned waddr = ring_addr;                              // next free record slot of this lane
    const int lane_h = a.hidden_offset + 4 * lane_half;
    // Records are 4 bytes: the approximate latent (in the accumulator's scale) with its low 6 mantissa bits replaced by
    // the value's place inside its 64-unit block (32 x row tile + 8 x accumulator quad + register in the quad; the
    // half-wave's + 4 comes from the lane); the block itself is known at flush time (tile 0 records belong to block
    // blk0, tile 1 records to blk1).  The truncation (< 2^-17 relative, towards zero) is part
    // of the error budget (prefilter_common.h).
    auto flush = [&](int blk0, int blk1) {
        if (__builtin_amdgcn_ballot_w64(waddr != ring_addr) != 0ull) {
            asm volatile("" ::: "memory");
            const int nrec = static_cast<int>((waddr - ring_addr) >> 8);
            // partner lane's count: v_permlane32_swap exchanges the half-waves in one VALU instruction
            const auto sw = __builtin_amdgcn_permlane32_swap(nrec, nrec, false, false);
            const int n_other = lane_half ? static_cast<int>(sw[0]) : static_cast<int>(sw[1]);
            int pos = count + (lane_half ? (n_other < kXsSlots ? n_other : kXsSlots) : 0);
            const int mine = (nrec < kXsSlots ? nrec : kXsSlots) * ((row_ok && ABL != 3) ? 1 : 0);
            // record slots by hand-written ds_reads: in front of a compiler-generated LDS read the compiler
            // drains the vector-memory queue (it cannot know the slots never alias an LDS-DMA in flight),
            // which here would wait for the acknowledgement of the previous stage's fill stores
            unsigned rr[4];
            asm volatile("ds_read_b32 %0, %4\n\t"
                         "ds_read_b32 %1, %4 offset:256\n\t"
                         "ds_read_b32 %2, %4 offset:512\n\t"
                         "ds_read_b32 %3, %4 offset:768\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(rr[0]), "=&v"(rr[1]), "=&v"(rr[2]), "=&v"(rr[3])
                         : "v"(ring_addr)
                         : "memory");
            // a record's low six bits are the value's place inside its 64-unit block (see filter_value): hidden index =
            // block base + place; one limit for "mine" and "room in the list", one pointer for the lane's run of entries
            const int hb0 = blk0 * kXsHT + lane_h, hb1 = blk1 * kXsHT + lane_h;
            const int room = cap_part - pos;
            const int lim = mine < room ? mine : room;
            uint2* wp = list + pos;
            auto put = [&](int j, unsigned rec) {
                if (j < lim) {
                    const int h = ((rec & 32u) ? hb1 : hb0) + static_cast<int>(rec & 63u);
                    wp[j] = make_uint2(__float_as_uint(__uint_as_float(rec & 0xFFFFFFC0u) * inv_r), static_cast<unsigned>(h));
                }
            };
            // slot j holds a record only for lanes with more than j of them: stop at the first empty level
            // (typically two or three)
            put(0, rr[0]);
            if (__builtin_amdgcn_ballot_w64(mine > 1) != 0ull) {
                put(1, rr[1]);
                if (__builtin_amdgcn_ballot_w64(mine > 2) != 0ull) {
                    put(2, rr[2]);
                    put(3, rr[3]);
                    if (__builtin_amdgcn_ballot_w64(mine > 4) != 0ull) {
                        static_assert(kXsSlots == 9, "the asm below reads slots 4..8");
                        unsigned r2[5];
                        asm volatile("ds_read_b32 %0, %5 offset:1024\n\t"
                                     "ds_read_b32 %1, %5 offset:1280\n\t"
                                     "ds_read_b32 %2, %5 offset:1536\n\t"
                                     "ds_read_b32 %3, %5 offset:1792\n\t"
                                     "ds_read_b32 %4, %5 offset:2048\n\t"
                                     "s_waitcnt lgkmcnt(0)"
                                     : "=&v"(r2[0]), "=&v"(r2[1]), "=&v"(r2[2]), "=&v"(r2[3]), "=&v"(r2[4])
                                     : "v"(ring_addr)
                                     : "memory");
#pragma unroll
                        for (int j = 0; j < 5; ++j) put(4 + j, r2[j]);
                    }
                }
            }
            count += (nrec < kXsSlots ? nrec : kXsSlots) + (n_other < kXsSlots ? n_other : kXsSlots);
            if (nrec > kXsSlots || n_other > kXsSlots) count = cap_part + 1;  // lost records: exact fallback
            asm volatile("" ::: "memory");
            waddr = ring_addr;
        }
    };
    // One value of the filter: accumulator register q of row tile mt (bias quad bq = rows 8g..8g+3 [+4 for the upper
    // half-wave] of that tile, g = q / 4).  EXEC <- !(v < thr) (at or above the threshold, or NaN); the hit lanes write
    // their record and advance their write position; EXEC <- all.  After every fourth value the position is clamped.
    auto filter_value = [&](int mt, int q) {
        const float v = acc[mt][q];                          // latent * rinv (the chain started from bias * rinv)
        // the low six mantissa bits make room for the value's place inside the block: row tile, accumulator quad, register
        const unsigned rec = (__float_as_uint(v) & 0xFFFFFFC0u) | static_cast<unsigned>(mt * 32 + 8 * (q >> 2) + (q & 3));
        const float cmp = (ABL == 1 || ABL == 7) ? __builtin_huge_valf() : thr_s;
        asm volatile("v_cmpx_nlt_f32_e32 vcc, %[v], %[thr]\n\t"
                     "ds_write_b32 %[addr], %[rec]\n\t"
                     "v_add_u32_e32 %[addr], 0x100, %[addr]\n\t"
                     "s_mov_b64 exec, -1"
                     : [addr] "+v"(waddr)
                     : [v] "v"(v), [thr] "v"(cmp), [rec] "v"(rec)
                     : "vcc", "memory");
        if ((q & 3) == 3) waddr = waddr < ring_limit ? waddr : ring_limit;
    };
    auto load_bias = [&](int it, int mt, f32x4 (&bq)[4]) {
        const char* bb = bias_lds + (it % 3) * BIAS_BYTES + 16 * lane_half + mt * 128;
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(bb + 32 * g);
    };

    // ---- dense latent zero-fill -----------------------------------------------------------------------
    // The reference returns latent * mask as a dense [B, H] tensor (sae/binary.py:96-99): 99.8 % zeros, 8 GiB
    // per batch.  This kernel is bound by instruction issue and leaves HBM idle, so the zeros are written
    // here: at the top of every stage (before the DMA, so the stage's vmcnt(0) also retires them a whole
    // stage later) each wave writes `fill_quota` pieces of 1 KiB -- 256 consecutive columns of one of its
    // 32 rows per store instruction, walking row after row (fully coalesced lines, one sequential stream
    // per wave).  The k survivors are scattered in by the caller afterwards.
    // Contiguous latent (ld == H): the workgroup's 256 rows are one block of memory and the eight waves
    // write adjacent runs of it every stage (one sequential stream per workgroup); otherwise each wave walks
    // its own 32 rows.
    const int fill_ppr = a.H / 256;                          // pieces per row
    const bool fill_linear = a.dense_ld == a.H;
    const long long wg_row0 = static_cast<long long>(blockIdx.x) * kXsRows;
    const long long fill_row0 = wg_row0 + (fill_linear ? 0 : wave * 32);
    const long long rows_here = a.B - fill_row0 < (fill_linear ? kXsRows : 32) ? a.B - fill_row0 : (fill_linear ? kXsRows : 32);
    // pieces this wave's sequence covers; with a split hidden range every part takes an equal share of them
    const long long fill_all = (rows_here > 0 ? rows_here : 0) * fill_ppr;
    const long long fill_begin = fill_all * part / a.parts, fill_total = fill_all * (part + 1) / a.parts;
    long long fill_next = fill_begin + (fill_linear ? static_cast<long long>(wave) * a.fill_cw : 0);   // next piece (wave-uniform)
    const long long fill_step = fill_linear ? static_cast<long long>(kXsWaves - 1) * a.fill_cw : 0;
    // One piece per call, at most a.fill_cw per stage; the calls sit between the MFMA groups of the stage
    // (a burst of nine 1-KiB stores at the top of a stage overruns the write queues and stalls the wave).
    constexpr bool FILL = ABL != 9 && ABL != 10;             // ABL 9: no fill code at all (the zeros come from a co-resident fill kernel); 10 = 9 + stamps
    int fill_left = 0, nfill = 0;                            // budget / store instructions issued this stage
    int fill_r = 0, fill_c = 0;
    auto fill_begin_stage = [&]() {
        if (FILL && a.dense) {
            fill_left = a.fill_cw;
            nfill = 0;
            fill_r = static_cast<int>(fill_next) / fill_ppr;
            fill_c = static_cast<int>(fill_next) - fill_r * fill_ppr;
        }
    };
    auto fill_one = [&]() {
        if (FILL && fill_left > 0) {                         // wave-uniform
            if (fill_next < fill_total) {
                float* base = a.dense + (fill_row0 + fill_r) * a.dense_ld + fill_c * 256;      // wave-uniform
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                __builtin_nontemporal_store(z, reinterpret_cast<f32x4*>(base) + lane);   // streaming: keep W in L2
                ++nfill;
            }
            ++fill_next;
            if (++fill_c == fill_ppr) { fill_c = 0; ++fill_r; }
            if (--fill_left == 0) fill_next += fill_step;    // linear mode: skip the other waves' runs
        }
    };
    // vmcnt retires loads, stores and LDS-DMA together in issue order (MI355X guide, "s_waitcnt vmcnt(N)"):
    // the fill stores are the youngest operations of a stage, so waiting for all but them retires the DMA
    // without waiting for the stores' acknowledgement (they get the whole next stage for that).
    auto wait_all_but = [&](int n) {
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };

    // ---- main loop ---------------------------------------------------------------------------------------
    // A wave hides its own epilogue: the 32 MFMAs of row tile 0 of stage s carry the filter of row tile 1 of
    // stage s-1 in their shadows (one value = 8 instructions per two MFMAs), the 32 MFMAs of row tile 1 carry
    // the filter of row tile 0 of stage s.  (With the filter after the MFMAs of a stage, even with the two
    // waves of a SIMD run out of phase, a stage took 8300 cycles: each wave's MFMA phase (2900) and filter
    // (2800) are serial, and the other wave's MFMAs slow the filter down to ~10 cycles per instruction.)
    // The sched_barriers pin the interleave and keep the fragment reads one group ahead of their MFMAs.
    // kb = 8g + j: the chunk positions repeat every 8 k-blocks, 256 bytes apart; row tile mt is 32 rows =
    // 32 * CPR * 16 bytes further.
    static_assert(MT == 2 && KB % 4 == 0, "the interleave below is written for two row tiles");
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int which) {
        if (ABL >= 5 && ABL != 9) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            tacc[which] += t - tprev;
            tprev = t;
        }
    };
    if (ABL >= 5 && ABL != 9) tprev = __builtin_amdgcn_s_memtime();
    constexpr bool FILTER = ABL != 2 && ABL != 8;
    constexpr int VPG = 32 / KB;                             // filter values per pair of MFMAs (KB = 32: 1)
    static_assert(VPG >= 1 && VPG * (KB / 2) == 16, "16 values spread over KB/2 MFMA pairs");
    // phase 0 (first pass of a stage): one DMA piece of the next stage per MFMA group -- an LDS-DMA issue costs
    // 60-180 cycles, eight of them at the top of a stage were 5-10 % of it; phase 1: the stage's fill stores, which
    // must stay younger than every DMA piece for the counted wait at the end of the stage.
    // Pilot pass (in-kernel threshold): over the first pilot_stages blocks every lane keeps, per accumulator
    // register position, the running maximum of its approximate latents -- 16 group maxima per lane, 32 per
    // row, each over 64 * pilot_stages / 32 hidden units.  tau = the pilot_rank-th largest of the row's 32
    // maxima: at least pilot_rank pilot values reach it, and it sits at about the same quantile as the 20th
    // largest of the whole pilot block (1 - F^64 = 14/32 -> 0.9 % tail, ~300 values per row above it).  It
    // replaces a separate pilot GEMM + per-row selection (0.53 ms) by pilot_stages extra iterations here.
    float gmax[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) gmax[q] = -__builtin_huge_valf();
    auto track_value = [&](int mt, int q, const f32x4& bq) {
        const float v = fmaf(acc[mt][q], inv_r, bq[q & 3]);
        gmax[q] = fmaxf(gmax[q], v);                         // v_max_f32: a NaN operand is ignored
    };
    // with_filter: in the MFMA shadows, filter acc[fmt].  cit >= 0: the chain of row tile mt starts from the bias of
    // iteration cit's block (copy cit % 3) times rinv -- the sweep; cit < 0: from zero -- the pilot pass.
    auto tile_pass = [&](const char* sbase, int mt, bool with_filter, int fmt, int cit, int phase,
                         bool dma) __attribute__((always_inline)) {
        // MFMAs of row tile mt over all of K; in their shadows the epilogue of acc[fmt]
        auto rd = [&](int kb) {
            return *reinterpret_cast<const f16x8*>(sbase + off[kb & 7] + 256 * (kb >> 3) + mt * (32 * CPR * 16));
        };
        auto epilogue = [&](int qv) {
            if (FILTER && with_filter) filter_value(fmt, qv);
        };
        f16x8 w0 = rd(0), w1 = rd(1), w2, w3;
        f32x16 c;
        if (cit >= 0) {
            f32x4 bqc[4];
            load_bias(cit, mt, bqc);
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 r2 = {rinv_r, rinv_r};
#pragma unroll
            for (int g = 0; g < 4; ++g) {                    // (two packed multiplies per accumulator quad)
                const f32x2 lo = f32x2{bqc[g][0], bqc[g][1]} * r2, hi = f32x2{bqc[g][2], bqc[g][3]} * r2;
                c[4 * g + 0] = lo[0]; c[4 * g + 1] = lo[1]; c[4 * g + 2] = hi[0]; c[4 * g + 3] = hi[1];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = 0.f;
        }
#pragma unroll
        for (int kb = 0; kb < KB; kb += 4) {
            w2 = rd(kb + 2); w3 = rd(kb + 3);
            __builtin_amdgcn_sched_barrier(0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, xf[kb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, xf[kb + 1], c, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < VPG; ++u) epilogue((kb / 2) * VPG + u);
            if (phase == 0) {
                if (dma && kb / 4 < IPW) {                     // (dma: a compile-time constant at every call site)
                    if (DMAE) { issue_piece(2 * (kb / 4)); issue_piece(2 * (kb / 4) + 1); }
                    else issue_piece(kb / 4);
                }
            } else {
                fill_one();
                fill_one();
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kb + 4 < KB) { w0 = rd(kb + 4); w1 = rd(kb + 5); }
            __builtin_amdgcn_sched_barrier(0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, xf[kb + 2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3, xf[kb + 3], c, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < VPG; ++u) epilogue((kb / 2 + 1) * VPG + u);
            __builtin_amdgcn_sched_barrier(0);
        }
        acc[mt] = c;
    };
    // tau from the group maxima: MSB-first bisection on the monotone keys of the lane pair's 32 values
    auto select_tau = [&]() {
        uint32_t key[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) key[q] = mono_key(gmax[q]);
        uint32_t T = 0u;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = T | (1u << bit);
            int c = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) c += key[q] >= trial ? 1 : 0;
            const auto sw = __builtin_amdgcn_permlane32_swap(c, c, false, false);
            c += lane_half ? static_cast<int>(sw[0]) : static_cast<int>(sw[1]);
            if (c >= a.pilot_rank) T = trial;
        }
        // key -> value (inverse of mono_key; the maxima are never NaN)
        return __uint_as_float((T & 0x80000000u) ? (T & 0x7FFFFFFFu) : ~T);
    };
    // One iteration = one 64-unit hidden block: flush, (DMA of the next block + MFMAs of row tile 0 + epilogue
    // of the previous tile 1), (fill stores + MFMAs of row tile 1 + epilogue of tile 0), counted wait, barrier.
    // Written once, instantiated for the pilot loop and the sweep loop (two loops, so that the pilot's maxima
    // do not occupy registers during the sweep).
    // The two waves of a SIMD (w and w + 4) flush at different points of a stage (no-fill build): a flush is ~600
    // cycles of VALU, LDS and store issue without a single MFMA, and with both waves flushing right behind the barrier
    // the matrix pipe of the SIMD idles for that long every stage.  Waves 0-3 flush at the top of the stage, waves 4-7
    // between the two row-tile passes (their records: row tiles 0 and 1 of the previous iteration's block).  Same
    // records, same lists (a row's list is written by one wave only, in the same order).
    // (sweep + fill 2.41 -> 2.37 ms.  Only in the no-fill build: with the zero-fill inside the sweep the fill stores have to
    // stay the youngest vector-memory operations of a stage.)
    constexpr bool STAGGER = ABL == 9 || ABL == 10;
    const bool late_wave = STAGGER && wave >= 4;             // wave-uniform
    auto iteration = [&](int it) __attribute__((always_inline)) {
        // records in the slots: row tile 0 of the previous iteration's block, row tile 1 of the one before it
        // (sweep iterations work on consecutive blocks: plain arithmetic, no block_of with its compare and branch; the
        // first two iterations have no such records yet and do not use the numbers)
        const int cur = s_begin + (it - ps);
        if (!late_wave) flush(cur - 1, cur - 2);             // older than the DMA issued next
        stamp(0);
        if (dma_wave) next_block();                          // iteration it+1 -> the buffer read during it-1
        fill_begin_stage();                                  // fill stores: the youngest vector-memory operations of the stage
        stamp(1);
        const char* sbase = xs_smem + (it % kXsStages) * STAGE_BYTES;
        // first pass: epilogue of the previous iteration's row tile 1 (at the first sweep iteration acc[1] is -inf: nothing
        // passes -- a run-time guard here was a scalar compare and a branch in front of every filtered value)
        tile_pass(sbase, 0, true, 1, it, 0, true);
        ++ld;
        // a lane that already holds five records could overflow its nine slots in the second pass: flush now (only
        // row tile 1 records of the previous block are there; rare; the stores are younger than the stage's DMA,
        // which only makes the wait below stricter)
        if (late_wave || __builtin_amdgcn_ballot_w64(waddr > ring_addr + 4u * 256u) != 0ull)
            flush(cur - 1, cur - 1);                         // (waves 0-3 hold no row tile 0 records at this point)
        stamp(2);
        tile_pass(sbase, 1, true, 0, it, 1, false);
        stamp(3);
        while (fill_left > 0) fill_one();                    // (budgets beyond the 16 slots of a stage)
        // retire iteration it+1 (for every wave) before anyone reads it; also frees this iteration's buffer
        wait_all_but(late_wave ? 0 : nfill);                  // (waves 4-7: their flush's stores are younger than the DMA)
        stamp(4);
        __builtin_amdgcn_s_barrier();
        stamp(5);
    };
    if (ps > 0) {
        // pilot iterations: MFMAs of a row tile, then its maxima right away (64 VALU instructions per iteration,
        // not worth hiding; deferring them as the sweep does would keep 16 more registers live)
#pragma unroll 1
        for (int it = 0; it < ps; ++it) {
            next_block();
            fill_begin_stage();
            const char* sbase = xs_smem + (it % kXsStages) * STAGE_BYTES;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                tile_pass(sbase, mt, false, 0, -1, mt, mt == 0);
                if (mt == 0) ++ld;
                if (FILTER) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_lds + (it % 3) * BIAS_BYTES + 16 * lane_half +
                                                                         mt * 128 + 32 * g);
#pragma unroll
                        for (int i = 0; i < 4; ++i) track_value(mt, 4 * g + i, bq);
                    }
                }
            }
            while (fill_left > 0) fill_one();
            wait_all_but(nfill);
            __builtin_amdgcn_s_barrier();
        }
        const float tau_row = select_tau();
        thr_r = tau_row - margin_row;
        thr_s = thr_r * rinv_r;
        if (part == 0 && row_ok && lane_half == 0) a.tau_out[row] = tau_row;
    }
    // nothing of "the previous iteration's row tile 1" exists at the first sweep iteration
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[1][r] = -__builtin_huge_valf();
#pragma unroll 1
    for (int it = ps; it < n_iter; ++it) iteration(it);
    while (FILL && a.dense && fill_next < fill_total) {              // (quota * iterations covers the block; safety net)
        fill_begin_stage();
        while (fill_left > 0) fill_one();
    }
    if (ABL >= 5 && ABL != 9) { asm volatile("" : "+v"(acc[0]), "+v"(acc[1])); }
    if (ABL >= 5 && ABL != 9 && a.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a.stamps[(static_cast<size_t>(blockIdx.x) * kXsWaves + wave) * 8 + i] = tacc[i];
    }
    if (FILTER && n_iter > ps) {
        // what the last iteration left (tile 0 of the last block, tile 1 of the one before), then the last block's tile 1
        flush(block_of(n_iter - 1), block_of(n_iter > 1 ? n_iter - 2 : 0));
#pragma unroll
        for (int q = 0; q < 16; ++q) filter_value(1, q);
    }
    flush(0, block_of(n_iter - 1));
    if (row_ok && lane_half == 0) my_cnt[row] = count;
}

inline bool xstat_supported(int D, int Hs, int hidden_offset) {
    return (D == 512 || D == 256 || D == 128) && Hs > 0 && Hs % kXsHT == 0 && hidden_offset % 4 == 0 &&
           static_cast<int64_t>(hidden_offset) + Hs <= (1 << 30);
}

// hidden-range split for small batches: double while the grid stays within the chip's 256 CUs, the stage
// count divides evenly and every part keeps a list segment of at least 128 entries
inline int xstat_parts(int B, int Hs, int cap) {
    const int wgs = (B + kXsRows - 1) / kXsRows, stages = Hs / kXsHT;
    int parts = 1;
    while (parts < 8 && wgs * parts * 2 <= 256 && stages % (parts * 2) == 0 && cap / (parts * 2) >= 128) parts *= 2;
    return parts;
}

template <int KB, int ABL = 0>
inline int launch_xstat_one(const XsArgs& a, hipStream_t stream) {
    constexpr size_t lds = static_cast<size_t>(kXsStages) * kXsHT * 16 * KB * 2 +
                           3 * kXsHT * 4 + static_cast<size_t>(kXsWaves) * kXsRingSlots * 256;
    auto kern = sweep_xstat_f16_kernel<KB, ABL>;
    QSAE_SET_MAX_LDS_ONCE(kern, lds);     // per instantiation and device
    hipLaunchKernelGGL(kern, dim3((a.B + kXsRows - 1) / kXsRows, a.parts), dim3(64 * kXsWaves), lds, stream, a);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

// ablate: 0 = the complete kernel; 9 = the build without any zero-fill code (the product path when the zeros come from
// the co-resident fill kernel); 1-8 = timing ablations whose results are wrong, instantiated in the debug library only.
inline int launch_xstat(int D, const XsArgs& a, hipStream_t stream, int ablate = 0) {
#ifdef QSAE_DEBUG_BUILD
    if (D == 512 && ablate == 1) return launch_xstat_one<32, 1>(a, stream);
    if (D == 512 && ablate == 2) return launch_xstat_one<32, 2>(a, stream);
    if (D == 512 && ablate == 3) return launch_xstat_one<32, 3>(a, stream);
    if (D == 512 && ablate == 4) return launch_xstat_one<32, 4>(a, stream);
    if (D == 512 && ablate == 5) return launch_xstat_one<32, 5>(a, stream);
    if (D == 512 && ablate == 6) return launch_xstat_one<32, 6>(a, stream);
    if (D == 512 && ablate == 7) return launch_xstat_one<32, 7>(a, stream);
    if (D == 512 && ablate == 8) return launch_xstat_one<32, 8>(a, stream);
    if (D == 512 && ablate == 10) return launch_xstat_one<32, 10>(a, stream);
#endif
    if (D == 512 && ablate == 9) return launch_xstat_one<32, 9>(a, stream);
    switch (D) {
        case 512: return launch_xstat_one<32>(a, stream);
        case 256: return launch_xstat_one<16>(a, stream);
        case 128: return launch_xstat_one<8>(a, stream);
        default: return fail(QSAE_ERR_UNSUPPORTED, "%s: D must be 128, 256 or 512", __func__);
    }
}

}  // namespace qsae

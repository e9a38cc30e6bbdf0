// encode_topk.hip -- encoder + per-row top-k without a [B, H] latent in HBM.
//
// Replaces `latent = encode(x); latent.topk(k)` (reference sae/binary.py:93-94,
// sae/baseline.py:22-35).  Results are identical to qsae_encode_dense + qsae_topk_rows.
//
// Fused form (large batches):
//   1. pilot   : the first P hidden units are contracted densely ([B, P], P = H/16) and a per-row
//                threshold tau = j-th largest pilot value is taken (j = 20).  tau is a valid lower
//                bound of the row's k-th largest latent as long as fewer than j of the row's top-k
//                live in the pilot block (hypergeometric, mean k/16; for exchangeable hidden units
//                the chance of >= 20 is ~1e-8) -- and validity is CHECKED, not assumed (step 3).
//   2. sweep   : every workgroup owns 128 activation rows and sweeps all remaining hidden tiles
//                with the exact-fp32 MFMA kernel (hidden units on accumulator registers, rows on
//                lanes); the epilogue compares each accumulator with its row's tau (one v_cmp per
//                value) and appends the few survivors (~1 %) to that row's candidate list.
//   3. select  : one wave per row picks the exact top-k of the candidates with a 48-step
//                ballot radix select on (value, index) keys.  A row with fewer than k candidates
//                (tau was not a lower bound) or an overflowing list is flagged ...
//   4. fallback: ... and flagged rows (normally none) are recomputed by the unfused kernels.
// Small problems use the chunked two-kernel form directly.
#include <vector>
#include <type_traits>

#include "gemm_mfma_f32.h"
#include "gemm_mfma_f32_dma.h"
#include "sweep_xstat_f16.h"
#include "decode_row.h"

namespace qsae {

int topk_rows_dispatch(float* latent, int64_t ld, int B, int H, int k, int32_t* idx, float* val, int zero_rest,
                       float* tau, uint2* cand, int* cnt, int cap, float* dense, int64_t dense_ld, hipStream_t s,
                       const float* margin = nullptr, int stride = 0);
int scatter_rows(const int32_t* idx, const float* val, int B, int k, int H, float* dense, int64_t ld, hipStream_t s);
int decode_binary_sparse_rows(const int* rows, int nrows, const int32_t* idx, const float* val, int k, int H,
                              const RowDecode& d, hipStream_t s);
int densify_rows(const int32_t* idx, const float* val, int B, int k, int H, float* dense, int64_t ld, hipStream_t s);

constexpr int kChunkRows = 1024;   // chunked form: 1024 x 32768 x 4 B = 128 MiB of latent per chunk
constexpr int kCandCap = 1024;     // candidate slots per row
constexpr int kFusedMinRows = 2048;
constexpr int kFusedMinHidden = 8192;
constexpr int kFillCoWaves = 1024; // fill waves beside the sweep: one per SIMD, so every sweep wave has the same neighbour
constexpr int kFillCoPace = 3;     // s_sleep(1) per store: the fill ends with the sweep (scan in the kernel's comment; 4 until the sweep lost 0.12 ms in round 2)

// Tuning / ablation switches.  The product library (libqsae_hip.so) is built without QSAE_DEBUG_BUILD: every switch is
// a compile-time constant there, no qsae_debug_* symbol exists and no ablation kernel is instantiated.  The debug
// library (libqsae_hip_debug.so, same sources with -DQSAE_DEBUG_BUILD; used by tools/ and by the tests that need to
// force a path on a small shape) makes them process-wide variables behind the qsae_debug_* setters.
#ifdef QSAE_DEBUG_BUILD
#define QSAE_TUNABLE static int
#define QSAE_TUNABLE_PTR static unsigned long long*
#else
#define QSAE_TUNABLE static constexpr int
#define QSAE_TUNABLE_PTR static constexpr unsigned long long*
#endif
QSAE_TUNABLE kPilotRank = 20;        // tau = kPilotRank-th largest pilot value (together with the pilot width)
QSAE_TUNABLE g_pilot_div = 16;       // pilot block = H / g_pilot_div hidden units
QSAE_TUNABLE g_force_path = 0;       // 0 auto, 1 chunked, 2 fused
QSAE_TUNABLE g_sweep_kernel = 0;     // K-interleaved operands: 0 = LDS-DMA sweep kernel, 1 = register-staged one
QSAE_TUNABLE_PTR g_xstat_stamps = nullptr;   // device buffer for the phase stamps (ablation 5)
QSAE_TUNABLE g_fuse_xprep = 0;       // 1: the stationary sweep scales / converts the activations in its prologue (no gain
                                     // measured: the prologue costs what the 0.07 ms preparation launch saves)
QSAE_TUNABLE g_inkernel_pilot = 1;   // the stationary sweep derives tau itself (no pilot GEMM / selection launches)
QSAE_TUNABLE g_inkernel_rank = 0;    // tau = this rank among the row's 32 group maxima; 0 = from k (inkernel_rank)
QSAE_TUNABLE g_pilot_tile = 0;       // fp16 pilot GEMM tile: 0 = 256 x 256 (2 stages), 1 = 256 x 128 (3 stages)
QSAE_TUNABLE g_fill_in_sweep = 1;    // zero-fill of the dense latent inside the activation-stationary sweep
QSAE_TUNABLE g_fill_co = 1;          // zeros from a co-resident fill kernel on a second stream (0: inside the sweep; > 1: that many fill waves)
QSAE_TUNABLE g_xstat_rot = 2;        // DMA rotation multiplier (sweep_xstat_f16.h)
QSAE_TUNABLE_PTR g_ref_stamps = nullptr;     // device buffer [8] for refine phase stamps
QSAE_TUNABLE g_ref_ablate = 0;       // timing experiments on the refine kernel (results wrong when non-zero)
#ifdef QSAE_AB_NO_SLICED
QSAE_TUNABLE g_ref_sliced = 0;
#else
QSAE_TUNABLE g_ref_sliced = 1;       // refinement as select / slice-major chains / rank launches: 1 = where it pays (large batches), 0 = never, 2 = wherever the shape allows
#endif
QSAE_TUNABLE g_xstat_ablate = 0;     // timing experiments only (results are wrong when non-zero)
QSAE_TUNABLE g_x_phase = 3;          // experiment: bit 0 = run x prep + sweep (+ fill), bit 1 = run the refinement
QSAE_TUNABLE g_x_parts = 0;          // experiment: hidden-range parts of the stationary sweep (0 = xstat_parts)
QSAE_TUNABLE g_pref_tile = 2;        // fp16 sweep: 2 = activation-stationary kernel (where supported), 0 = 256 x 256 tile
                                     // (2 stages), 1 = 256 x 128 tile (3 stages)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static bool use_fused(int B, int D, int H, int k) {
    if (g_force_path == 1) return false;
    const bool shape_ok = (H % 4 == 0) && (H / 16 >= 256) && k <= 256 && H <= 65536;
    if (g_force_path == 2) return shape_ok;
    return shape_ok && B >= kFusedMinRows && H >= kFusedMinHidden;
}

static int pilot_width(int H) {
    int p = H / g_pilot_div;
    p = (p + 127) / 128 * 128;
    return p;
}

constexpr int kFusedSplit = 4;     // exact fp32 sweep: workgroups per activation panel (hidden range in quarters, see run_fused)

struct FusedLayout {
    size_t pilot, tau, cnt, cnt_split, cand, flags, fx, flat, fidx, fval, total;
};

static FusedLayout fused_layout(int B, int D, int H, int k) {
    FusedLayout L;
    const int P = pilot_width(H);
    size_t off = 0;
    L.pilot = off; off = align_up(off + static_cast<size_t>(B) * P * 4, 256);
    L.tau = off;   off = align_up(off + static_cast<size_t>(B) * 4, 256);
    L.cnt = off;   off = align_up(off + static_cast<size_t>(B) * 4, 256);
    L.cnt_split = off; off = align_up(off + static_cast<size_t>(B) * 4 * (kFusedSplit - 1), 256);   // list-segment counters of slices 1..
    L.cand = off;  off = align_up(off + static_cast<size_t>(B) * kCandCap * 8, 256);
    L.flags = off; off = align_up(off + (static_cast<size_t>(B) + 4) * 4, 256);     // [0] = count, then row ids
    L.fx = off;    off = align_up(off + static_cast<size_t>(kChunkRows) * D * 4, 256);
    L.flat = off;  off = align_up(off + static_cast<size_t>(kChunkRows) * H * 4, 256);
    L.fidx = off;  off = align_up(off + static_cast<size_t>(kChunkRows) * k * 4, 256);
    L.fval = off;  off = align_up(off + static_cast<size_t>(kChunkRows) * k * 4, 256);
    L.total = off;
    return L;
}

// ---- sweep epilogue: threshold filter ---------------------------------------------------------
// APPROX (fp16 prefilter): the accumulator holds the scaled fp16 contraction; the value compared and
// stored is fma(acc, inv[row], bias[h]) and the row threshold is tau[row] - margin[row].
template <int BM, int BN, int WMW = 2, int WNW = 2, bool APPROX = false>
struct EpiFilter {
    static constexpr int WTM = BM / WMW, WTN = BN / WNW, MT = WTM / 32, NT = WTN / 32;
    static constexpr int kThreads = 64 * WMW * WNW;
    static constexpr int kCheckpoints = 0;
    static constexpr int kLdsFloats = BN;      // per-row candidate counters
    static constexpr int kStoresPerFinish = (BM * BN * 4) / (kThreads * 16);   // zero-fill stores per wave
    struct Args {
        const float* bias;   // [hidden], already offset to the first swept hidden unit (may be null)
        const float* tau;    // [B]
        uint2* cand;         // [B][cap]
        int* cnt;            // [B]  in: candidates already present, out: total
        int cap;
        int hidden_offset;   // index of the first swept hidden unit
        float* dense;        // optional [B][dense_ld]: the tile's block of the dense latent is zero-filled
        int64_t dense_ld;    //   here (the k survivors are scattered in afterwards); nullptr = no dense output
        const float* inv;    // APPROX: [B] 1 / (row scale * weight scale), a power of two
        const float* margin; // APPROX: [B] 2 * eps_b
        // The hidden range may be split over `parts` workgroups per activation panel (SweepMap::msplit, TileCtx::part):
        // slice p appends to segment [p * cap / parts, (p + 1) * cap / parts) of every row's list and counts in
        // cnt (p == 0, which also holds the pilot's seeds) or cnt_parts[(p - 1) * rows + row].  parts <= 1: one segment.
        int parts = 1;
        int* cnt_parts = nullptr;
    };
    float tau[NT];
    float inv[NT];
    bool col_ok[NT];

    __device__ __forceinline__ void begin(const Args& a, const TileCtx& c) {
        int* counters = reinterpret_cast<int*>(c.lds_epi);
        if (c.tid < BN) {
            const int row = c.n0 + c.tid;
            counters[c.tid] = (row < c.N && c.part == 0) ? a.cnt[row] : 0;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = c.n0 + c.wn * WTN + nt * 32 + c.lane_col;
            col_ok[nt] = col < c.N;
            tau[nt] = col_ok[nt] ? a.tau[col] : __builtin_huge_valf();
            inv[nt] = 1.0f;
            if (APPROX && col_ok[nt]) {
                tau[nt] = tau[nt] - a.margin[col];
                inv[nt] = a.inv[col];
            }
        }
        // visibility of the counters: the kernel's first __syncthreads() follows begin()
    }
    __device__ __forceinline__ void init(const Args& a, f32x16 (&acc)[MT][NT], const TileCtx& c) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int h = c.m0 + c.wm * WTM + mt * 32 + mfma_row(r, c.lane_half);
                h = h < c.M ? h : c.M - 1;
                const float b = (!APPROX && a.bias) ? a.bias[h] : 0.0f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r] = b;
            }
    }
    __device__ __forceinline__ void checkpoint(const Args&, f32x16 (&)[MT][NT], const TileCtx&, int) {}
    __device__ __forceinline__ void finish(const Args& a, f32x16 (&acc)[MT][NT], const TileCtx& c) {
        int* counters = reinterpret_cast<int*>(c.lds_epi);
        if (a.dense != nullptr) {
            // The reference returns latent*mask as a dense [B, H] tensor (sae/binary.py:96-99): 99.8 %
            // zeros.  Each tile zero-fills its own BN x BM block with fire-and-forget 16-byte stores that
            // ride under the next tile's MFMAs, instead of a separate 8 GiB memset pass.  Thread t takes
            // the 16-byte chunks t, t + T, ...; consecutive threads -> consecutive chunks of one row.
            constexpr int CPR = BM / 4;                           // chunks per row of the block
            const int h0 = c.m0 + a.hidden_offset;               // first hidden unit of this tile
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < (BN * CPR) / kThreads; ++i) {
                const int chunk = i * kThreads + c.tid;
                const int row = c.n0 + chunk / CPR, cc = 4 * (chunk % CPR);
                if (row < c.N && (c.m0 + cc) < c.M)
                    *reinterpret_cast<f32x4*>(a.dense + static_cast<int64_t>(row) * a.dense_ld + h0 + cc) = z;
            }
        }
        float hb[MT][16];   // APPROX: bias of the hidden unit behind each accumulator register
        if (APPROX) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int h = c.m0 + c.wm * WTM + mt * 32 + mfma_row(r, c.lane_half);
                    h = h < c.M ? h : c.M - 1;
                    hb[mt][r] = a.bias ? a.bias[h] : 0.0f;
                }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int lcol = c.wn * WTN + nt * 32 + c.lane_col;
            const float t = tau[nt];
            const int cap_part = a.parts > 1 ? a.cap / a.parts : a.cap;
            uint2* list = a.cand + static_cast<int64_t>(c.n0 + lcol) * a.cap + (a.parts > 1 ? c.part * cap_part : 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = APPROX ? fmaf(acc[mt][nt][r], inv[nt], hb[mt][r]) : acc[mt][nt][r];
                    if (!(v < t)) {      // v >= tau, or NaN (which ranks above everything)
                        const int h = c.m0 + c.wm * WTM + mt * 32 + mfma_row(r, c.lane_half);
                        if (h < c.M && col_ok[nt]) {
                            const int pos = atomicAdd(&counters[lcol], 1);
                            if (pos < cap_part)
                                list[pos] = make_uint2(__float_as_uint(v), static_cast<uint32_t>(h + a.hidden_offset));
                        }
                    }
                }
        }
    }
    __device__ __forceinline__ void end(const Args& a, const TileCtx& c) {
        __syncthreads();
        const int* counters = reinterpret_cast<const int*>(c.lds_epi);
        if (c.tid < BN) {
            const int row = c.n0 + c.tid;
            int* dst = (a.parts > 1 && c.part > 0) ? a.cnt_parts + static_cast<int64_t>(c.part - 1) * c.N : a.cnt;
            if (row < c.N) dst[row] = counters[c.tid];
        }
    }
};

// ---- select: exact top-k of a row's candidates, one wave per row ---------------------------------
constexpr int kSelWaves = 4;
constexpr int kSelSlots = kCandCap / 64;   // candidates per lane

__global__ void __launch_bounds__(64 * kSelWaves)
select_topk_kernel(const uint2* __restrict__ cand, const int* __restrict__ cnt, int cap, int B, int H, int k,
                   int32_t* __restrict__ idx_out, float* __restrict__ val_out, int* __restrict__ flags, int parts,
                   const int* __restrict__ cnt_parts) {
    __shared__ unsigned long long sel[kSelWaves][256];
    __shared__ unsigned short sel_src[kSelWaves][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * kSelWaves + wave;
    if (b >= B) return;
    // the row's list is `parts` segments of cap / parts entries (one per hidden-range slice of the sweep); seg_end[p] =
    // candidates in segments 0..p
    static_assert(kFusedSplit <= 4, "segment bookkeeping below is written for up to four segments");
    const int cap_part = cap / parts;
    int seg_end[4];
    bool seg_overflow = false;
    int n = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int np = p >= parts ? 0 : (p == 0 ? cnt[b] : cnt_parts[static_cast<int64_t>(p - 1) * B + b]);
        seg_overflow |= np > cap_part;
        n += np;
        seg_end[p] = n;
    }
    // candidate number i (0 <= i < n, segments in order) -> its slot in the row's list
    const int e0 = seg_end[0], e1 = seg_end[1], e2 = seg_end[2];      // (named: no runtime-indexed local array)
    auto slot_of = [&](int i) {
        const int p = (i >= e0 ? 1 : 0) + (i >= e1 ? 1 : 0) + (i >= e2 ? 1 : 0);
        const int first = i >= e2 ? e2 : (i >= e1 ? e1 : (i >= e0 ? e0 : 0));
        return p * cap_part + (i - first);
    };
    if (n < k || n > cap || seg_overflow) {       // tau not a lower bound, or list overflow
        if (lane == 0) {
            const int slot = atomicAdd(&flags[0], 1);
            flags[1 + slot] = b;
        }
        return;
    }
    // 48-bit keys: (monotone value << 16) | (H-1-index); all distinct, larger = better
    unsigned long long key[kSelSlots];
    const uint2* list = cand + static_cast<int64_t>(b) * cap;
#pragma unroll
    for (int s = 0; s < kSelSlots; ++s) {
        const int i = s * 64 + lane;
        if (i < n) {
            const uint2 c = list[slot_of(i)];
            key[s] = (static_cast<unsigned long long>(mono_key(__uint_as_float(c.x))) << 16) |
                     static_cast<unsigned long long>((H - 1) - static_cast<int>(c.y));
        } else {
            key[s] = 0ull;                        // below every real key (mono >= 0x007FFFFF)
        }
    }
    const int nslots = (n + 63) / 64;             // wave-uniform
    // Bits on which all keys agree need no trial: start below the highest differing bit.
    unsigned long long all_or = 0ull, all_and = ~0ull;
    for (int s = 0; s < nslots; ++s) {
        const bool live = (s * 64 + lane) < n;
        all_or |= live ? key[s] : 0ull;
        all_and &= live ? key[s] : ~0ull;
    }
    for (int off = 32; off > 0; off >>= 1) {
        all_or |= __shfl_xor(all_or, off, 64);
        all_and &= __shfl_xor(all_and, off, 64);
    }
    const unsigned long long differ = (all_or ^ all_and) & 0xFFFFFFFFFFFFull;
    const int top_bit = differ ? (63 - __clzll(differ)) : -1;
    unsigned long long T = all_and & ~((top_bit >= 0) ? ((2ull << top_bit) - 1ull) : 0ull);   // common prefix
    int at_or_above = n;                          // keys >= T
    // MSB-first bisection; stops as soon as the set {key >= T} has exactly k members (usually well
    // before the 16 index bits, which only matter when values tie at the boundary)
    for (int bit = top_bit; bit >= 0; --bit) {
        if (at_or_above == k) break;            // exactly k keys at or above T: they are the top-k
        const unsigned long long trial = T | (1ull << bit);
        int c = 0;
        for (int s = 0; s < nslots; ++s) c += __popcll(__ballot(key[s] >= trial));
        if (c >= k) { T = trial; at_or_above = c; }
    }
    // T is the k-th largest key: exactly k keys are >= T.  Compact them, then rank.
    unsigned long long* mine = sel[wave];
    unsigned short* src = sel_src[wave];
    int base = 0;
    for (int s = 0; s < nslots; ++s) {
        const bool keep = key[s] >= T;
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            mine[pos] = key[s];
            src[pos] = static_cast<unsigned short>(s * 64 + lane);
        }
        base += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int j = lane; j < k; j += 64) {
        const unsigned long long kj = mine[j];
        int rank = 0;
        for (int i = 0; i < k; ++i) rank += (mine[i] > kj) ? 1 : 0;
        const uint2 c = list[slot_of(src[j])];      // raw value bits and index of the survivor
        idx_out[static_cast<int64_t>(b) * k + rank] = static_cast<int32_t>(c.y);
        val_out[static_cast<int64_t>(b) * k + rank] = __uint_as_float(c.x);
    }
}

// ---- fallback helpers ---------------------------------------------------------------------------
constexpr int kMaxSpecRows = kChunkRows;   // upper bound of the caller's spec_rows (one fallback chunk)
__global__ void __launch_bounds__(256)
gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ rows, int n, int D, float* __restrict__ dst) {
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= static_cast<long long>(n) * D) return;
    const int r = static_cast<int>(gid / D), c = static_cast<int>(gid % D);
    dst[gid] = src[static_cast<long long>(rows[r]) * D + c];
}

// Variants with the row count read on the device (min(*count, cap) rows): the first chunk of the fallback is
// enqueued before the host knows the count.  Unused rows of dst take activation row r itself (cap <= B): defined,
// ordinary data -- all-zero rows would tie everywhere and send the top-k kernel down its slow tie path.
__global__ void __launch_bounds__(256)
gather_rows_dev_kernel(const float* __restrict__ src, const int* __restrict__ rows, const int* __restrict__ count,
                       int cap, int D, float* __restrict__ dst) {
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= static_cast<long long>(cap) * D) return;
    const int n = *count < cap ? *count : cap;
    const int r = static_cast<int>(gid / D), c = static_cast<int>(gid % D);
    dst[gid] = src[static_cast<long long>(r < n ? rows[r] : r) * D + c];
}

__global__ void __launch_bounds__(256)
scatter_topk_kernel(const int32_t* __restrict__ sidx, const float* __restrict__ sval, const int* __restrict__ rows,
                    int n, int k, int32_t* __restrict__ idx, float* __restrict__ val, float* __restrict__ dense,
                    int64_t dense_ld, int H) {
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= static_cast<long long>(n) * k) return;
    const int r = static_cast<int>(gid / k), j = static_cast<int>(gid % k);
    idx[static_cast<long long>(rows[r]) * k + j] = sidx[gid];
    val[static_cast<long long>(rows[r]) * k + j] = sval[gid];
    // optional: the row's entries of an already zero-filled dense latent
    if (dense && sidx[gid] >= 0 && sidx[gid] < H) dense[static_cast<long long>(rows[r]) * dense_ld + sidx[gid]] = sval[gid];
}

__global__ void __launch_bounds__(256)
scatter_topk_dev_kernel(const int32_t* __restrict__ sidx, const float* __restrict__ sval, const int* __restrict__ rows,
                        const int* __restrict__ count, int cap, int k, int32_t* __restrict__ idx, float* __restrict__ val,
                        float* __restrict__ dense, int64_t dense_ld, int H) {
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int n = *count < cap ? *count : cap;
    if (gid >= static_cast<long long>(n) * k) return;
    const int r = static_cast<int>(gid / k), j = static_cast<int>(gid % k);
    idx[static_cast<long long>(rows[r]) * k + j] = sidx[gid];
    val[static_cast<long long>(rows[r]) * k + j] = sval[gid];
    if (dense && sidx[gid] >= 0 && sidx[gid] < H) dense[static_cast<long long>(rows[r]) * dense_ld + sidx[gid]] = sval[gid];
}

static int dense_latent(const float* x, const float* W, const float* bias, int B, int D, int H, float* out, int64_t ld,
                 qsae_stream_t stream, bool kperm) {
    return kperm ? qsae_encode_dense_kperm(x, W, bias, B, D, H, QSAE_ACT_NONE, out, ld, stream)
                 : qsae_encode_dense(x, W, bias, B, D, H, QSAE_ACT_NONE, out, ld, stream);
}

static int run_chunked(const float* x, const float* W, const float* bias, int B, int D, int H, int k, int32_t* idx,
                       float* val, float* lat, qsae_stream_t stream, bool kperm) {
    for (int b0 = 0; b0 < B; b0 += kChunkRows) {
        const int rows = (B - b0) < kChunkRows ? (B - b0) : kChunkRows;
        int rc = dense_latent(x + static_cast<size_t>(b0) * D, W, bias, rows, D, H, lat, H, stream, kperm);
        if (rc != QSAE_OK) return rc;
        rc = qsae_topk_rows(lat, H, rows, H, k, idx + static_cast<size_t>(b0) * k, val + static_cast<size_t>(b0) * k, 0,
                            stream);
        if (rc != QSAE_OK) return rc;
    }
    return QSAE_OK;
}

// Flagged rows (tau not a valid lower bound, overflowing list, non-finite inputs): normally none.  They are
// recomputed by the unfused exact kernels.  Their number lives in device memory (flags[0], the row ids behind it); the
// host needs it to size those launches.  Three pieces, so that the caller decides where the one 4-byte read-back goes:
//   * flagged_spec  : the exact fallback for the first `spec` flagged rows with the count read ON THE DEVICE -- enqueued
//                     before the host knows the count (unused slots recompute ordinary rows into scratch);
//   * flagged_range : the exact fallback for flagged rows [first, nflag), count known to the host;
//   * the blocking entry points copy the count into the calling thread's pinned word, wait for THAT COPY only (an event
//     right behind it) and call flagged_range; the submit / finish pair hands the word to the caller instead.
struct FlaggedArgs {
    const float* x; const float* W; const float* bias;
    int B, D, H, k;
    int32_t* idx; float* val;
    char* ws; FusedLayout L;
    qsae_stream_t stream; bool kperm;
    float* dense; int64_t dense_ld;      // optional already zero-filled dense latent: the rows' entries are written into it
};

static int flagged_spec(const FlaggedArgs& a, int spec) {
    if (spec <= 0) return QSAE_OK;
    hipStream_t s = as_stream(a.stream);
    int* flags = reinterpret_cast<int*>(a.ws + a.L.flags);
    float* fx = reinterpret_cast<float*>(a.ws + a.L.fx);
    float* flat = reinterpret_cast<float*>(a.ws + a.L.flat);
    int32_t* fidx = reinterpret_cast<int32_t*>(a.ws + a.L.fidx);
    float* fval = reinterpret_cast<float*>(a.ws + a.L.fval);
    const long long tot = static_cast<long long>(spec) * a.D;
    hipLaunchKernelGGL(gather_rows_dev_kernel, dim3(static_cast<unsigned>((tot + 255) / 256)), dim3(256), 0, s, a.x,
                       flags + 1, flags, spec, a.D, fx);
    QSAE_LAUNCH_CHECK();
    int rc = dense_latent(fx, a.W, a.bias, spec, a.D, a.H, flat, a.H, a.stream, a.kperm);
    if (rc != QSAE_OK) return rc;
    rc = qsae_topk_rows(flat, a.H, spec, a.H, a.k, fidx, fval, 0, a.stream);
    if (rc != QSAE_OK) return rc;
    const long long tk = static_cast<long long>(spec) * a.k;
    hipLaunchKernelGGL(scatter_topk_dev_kernel, dim3(static_cast<unsigned>((tk + 255) / 256)), dim3(256), 0, s, fidx,
                       fval, flags + 1, flags, spec, a.k, a.idx, a.val, a.dense, a.dense_ld, a.H);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

static int flagged_range(const FlaggedArgs& a, int first, int nflag) {
    hipStream_t s = as_stream(a.stream);
    int* flags = reinterpret_cast<int*>(a.ws + a.L.flags);
    float* fx = reinterpret_cast<float*>(a.ws + a.L.fx);
    float* flat = reinterpret_cast<float*>(a.ws + a.L.flat);
    int32_t* fidx = reinterpret_cast<int32_t*>(a.ws + a.L.fidx);
    float* fval = reinterpret_cast<float*>(a.ws + a.L.fval);
    for (int f0 = first; f0 < nflag; f0 += kChunkRows) {
        const int n = (nflag - f0) < kChunkRows ? (nflag - f0) : kChunkRows;
        const int* rows = flags + 1 + f0;
        const long long tot = static_cast<long long>(n) * a.D;
        hipLaunchKernelGGL(gather_rows_kernel, dim3(static_cast<unsigned>((tot + 255) / 256)), dim3(256), 0, s, a.x, rows, n,
                           a.D, fx);
        QSAE_LAUNCH_CHECK();
        int rc = dense_latent(fx, a.W, a.bias, n, a.D, a.H, flat, a.H, a.stream, a.kperm);
        if (rc != QSAE_OK) return rc;
        rc = qsae_topk_rows(flat, a.H, n, a.H, a.k, fidx, fval, 0, a.stream);
        if (rc != QSAE_OK) return rc;
        const long long tk = static_cast<long long>(n) * a.k;
        hipLaunchKernelGGL(scatter_topk_kernel, dim3(static_cast<unsigned>((tk + 255) / 256)), dim3(256), 0, s, fidx, fval,
                           rows, n, a.k, a.idx, a.val, a.dense, a.dense_ld, a.H);
        QSAE_LAUNCH_CHECK();
    }
    return QSAE_OK;
}

// Blocking form: count -> this thread's pinned word, wait for that copy, exact fallback.  *nflag_out = the count.
static int flagged_blocking(const FlaggedArgs& a, int spec, int* nflag_out) {
    hipStream_t s = as_stream(a.stream);
    const int* flags = reinterpret_cast<const int*>(a.ws + a.L.flags);
    ThreadDeviceCtx* ctx = nullptr;
    int rc = thread_device_ctx(&ctx);
    if (rc != QSAE_OK) return rc;
    *ctx->pinned = 0;
    QSAE_HIP(hipMemcpyAsync(ctx->pinned, flags, sizeof(int), hipMemcpyDeviceToHost, s));
    QSAE_HIP(hipEventRecord(ctx->ev_copied, s));
    spec = spec < 0 ? 0 : (spec > kMaxSpecRows ? kMaxSpecRows : spec);
    spec = spec < a.B ? spec : a.B;
    rc = flagged_spec(a, spec);          // the device works on these while the host waits for the count
    if (rc != QSAE_OK) return rc;
    QSAE_HIP(hipEventSynchronize(ctx->ev_copied));
    const int nflag = *ctx->pinned;
    if (nflag_out) *nflag_out = nflag;
    if (nflag < 0 || nflag > a.B) return fail(QSAE_ERR_HIP, "%s: corrupt flagged-row count", __func__);
    return flagged_range(a, spec, nflag);
}

static int run_fused(const float* x, const float* W, const float* bias, int B, int D, int H, int k, int32_t* idx,
                     float* val, char* ws, qsae_stream_t stream, bool kperm, float* dense, int64_t dense_ld) {
    hipStream_t s = as_stream(stream);
    const SweepProfile prof = take_sweep_profile();
    const FusedLayout L = fused_layout(B, D, H, k);
    const int P = pilot_width(H);
    float* pilot = reinterpret_cast<float*>(ws + L.pilot);
    float* tau = reinterpret_cast<float*>(ws + L.tau);
    int* cnt = reinterpret_cast<int*>(ws + L.cnt);
    uint2* cand = reinterpret_cast<uint2*>(ws + L.cand);
    int* flags = reinterpret_cast<int*>(ws + L.flags);
    QSAE_HIP(hipMemsetAsync(flags, 0, sizeof(int), s));
    // 1. pilot block and per-row threshold
    int rc = dense_latent(x, W, bias, B, D, P, pilot, P, stream, kperm);
    if (rc != QSAE_OK) return rc;
    const int j = kPilotRank < P ? kPilotRank : P;
    // (the pilot kernel also zero-fills the pilot columns of the dense latent: the sweep only visits h >= P)
    rc = topk_rows_dispatch(pilot, P, B, P, j, nullptr, nullptr, 0, tau, cand, cnt, kCandCap, dense, dense_ld, s);
    if (rc != QSAE_OK) return rc;
    // 2. sweep of the remaining hidden units with the threshold filter (R = W rows, Cm = x rows)
    int split = 1;                                           // hidden-range slices per activation panel (list segments)
    int* cnt_split = reinterpret_cast<int*>(ws + L.cnt_split);
    {
        constexpr int BM = 128, BN = 128, BK = 32;
        using Epi = EpiFilter<BM, BN>;
        typename Epi::Args ea{bias ? bias + P : nullptr, tau, cand, cnt, kCandCap, P, dense, dense_ld, nullptr, nullptr};
        const int Hs = H - P;
        if (prof.begin) QSAE_HIP(hipEventRecord(prof.begin, s));
        if (kperm && g_sweep_kernel == 0 && D % kDmaBK == 0) {
            // One workgroup per (activation panel, quarter of the hidden range) instead of one per panel: the block map
            // deals the four workgroups of a panel to the same XCD next to each other, so an XCD's 32 resident workgroups
            // share 8 panels (2 MiB: they stay in its 4 MiB L2 for the whole sweep) instead of owning 32 (8 MiB: every
            // panel was re-fetched for every hidden tile -- 3.1x the algorithmic traffic, profiles/r01_traffic.json).
            // Each quarter appends to its own segment of the rows' candidate lists.
            using EpiD = EpiFilter<256, 128, 4, 2>;
            const int tiles = (Hs + 255) / 256;
            split = (tiles >= 4 * kFusedSplit && B >= 4096) ? kFusedSplit : 1;
            typename EpiD::Args ed{bias ? bias + P : nullptr, tau, cand, cnt, kCandCap, P, dense, dense_ld, nullptr, nullptr,
                                   split, cnt_split};
            rc = launch_gemm_dma<EpiD, 256, 128>(W + static_cast<size_t>(P) * D, Hs, x, B, D, ed, s,
                                                 split > 1 ? (tiles + split - 1) / split : 0);
        } else if (kperm) {
            using LA = LoaderF32<BM, BK, false, true, true>;
            using LB = LoaderF32<BN, BK, false, true, true>;
            typename LA::Args la{W + static_cast<size_t>(P) * D, D, Hs};
            typename LB::Args lb{x, D, B};
            rc = launch_gemm<LA, LB, Epi, BM, BN, BK>(la, lb, ea, Hs, B, D, /*sweep=*/0, s);
        } else if (D % BK == 0) {
            using LA = LoaderF32<BM, BK, false, true>;
            using LB = LoaderF32<BN, BK, false, true>;
            typename LA::Args la{W + static_cast<size_t>(P) * D, D, Hs};
            typename LB::Args lb{x, D, B};
            rc = launch_gemm<LA, LB, Epi, BM, BN, BK>(la, lb, ea, Hs, B, D, /*sweep=*/0, s);
        } else {
            using LA = LoaderF32<BM, BK, true>;
            using LB = LoaderF32<BN, BK, true>;
            typename LA::Args la{W + static_cast<size_t>(P) * D, D, Hs};
            typename LB::Args lb{x, D, B};
            rc = launch_gemm<LA, LB, Epi, BM, BN, BK>(la, lb, ea, Hs, B, D, /*sweep=*/0, s);
        }
        if (prof.end) QSAE_HIP(hipEventRecord(prof.end, s));
        if (rc != QSAE_OK) return rc;
    }
    // 3. exact selection among the candidates
    hipLaunchKernelGGL(select_topk_kernel, dim3((B + kSelWaves - 1) / kSelWaves), dim3(64 * kSelWaves), 0, s, cand, cnt,
                       kCandCap, B, H, k, idx, val, flags, split, cnt_split);
    QSAE_LAUNCH_CHECK();
    // 4. flagged rows (normally none): one 4-byte read-back, then the unfused kernels on those rows
    const FlaggedArgs fa{x, W, bias, B, D, H, k, idx, val, ws, L, stream, kperm, nullptr, 0};
    rc = flagged_blocking(fa, /*spec=*/0, nullptr);
    if (rc != QSAE_OK) return rc;
    if (dense) return scatter_rows(idx, val, B, k, H, dense, dense_ld, s);
    return QSAE_OK;
}

// =====================================================================================================
// fp16 prefilter: an order-preserving approximation decides WHICH hidden units can be in a row's top-k;
// every returned value and the final selection are exact fp32.
//
//   s^_bh = bias_h + (sum_k fp16(x_bk * sx_b) * fp16(W_hk * sw)) / (sx_b * sw)        (fp16 MFMA, fp32 accumulate)
//   |s^_bh - s_bh| <= eps_b   for the exact fmaf chain s_bh, with
//   eps_b = c1 * ||x_b||_2 * max_h ||W_h||_2 + (D + 8) 2^-24 max|bias| + tiny absolute terms,  c1 =
//       2^-10 (1 + 2^-11)   two fp16 roundings per product (power-of-two scalings are exact)
//     + 4 * D * 2^-24       fp32 accumulation of the exact fp16 x fp16 products, 4x safety on the unit roundoff
//     + D * 2^-24           the exact chain's own distance from the real-number dot product
//   (Cauchy-Schwarz bounds sum_k |x_k||w_k|; the bias term is there because every step of the exact chain
//   rounds at the magnitude of its running sum, which starts at the bias).  If t~ is the k-th largest s^ of a row, every member of the
//   exact top-k satisfies s^ >= t~ - 2 eps_b; those survivors (~90 of 32768) are re-evaluated with the
//   exact chain and ranked exactly.  tests/test_kernels_gpu.py measures max|s^ - s| / eps_b on hardware.
struct PrefLayout {
    size_t xq, inv, margin, cnt_parts, sl_offs, total_extra;
};
static PrefLayout pref_layout(int B, int D, size_t base) {
    PrefLayout P;
    size_t off = base;
    P.xq = off;     off = align_up(off + static_cast<size_t>(B) * D * 2, 256);
    P.inv = off;    off = align_up(off + static_cast<size_t>(B) * 4, 256);
    P.margin = off; off = align_up(off + static_cast<size_t>(B) * 4, 256);
    P.cnt_parts = off; off = align_up(off + static_cast<size_t>(B) * 4 * 7, 256);    // list-segment counters of parts 1..7
    P.sl_offs = off; off = align_up(off + static_cast<size_t>(B) * 65, 256);        // sliced refinement: survivors below slice s, [S + 1][B] bytes (kSlMaxSlices + 1 rows)
    P.total_extra = off;
    return P;
}

// meta (device float[4]): [0] sw (power-of-two weight scale), [1] max_h ||W_h||_2, [2] max|bias|, [3] max|W| while
// packing, afterwards max_h ||W_h - W^_h||_2 (the distance of the fp16 copy, pref_w_err_kernel)
__global__ void __launch_bounds__(256)
pref_w_stats_kernel(const float* __restrict__ W, const float* __restrict__ bias, int H, int D, unsigned* __restrict__ meta) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= H) return;
    float mx = 0.f, ss = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float w = W[static_cast<int64_t>(row) * D + d];
        const float a = fabsf(w);
        mx = (a > mx || a != a) ? a : mx;       // NaN propagates (a != a)
        ss = fmaf(w, w, ss);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(mx, off, 64);
        mx = (o > mx || o != o) ? o : mx;
        ss += __shfl_xor(ss, off, 64);
    }
    if (lane == 0) {
        const float nrm = sqrtf(ss) * 1.000001f;
        // non-negative floats (and NaN, which has the largest bit pattern) order like their bit patterns
        atomicMax(&meta[1], __float_as_uint(nrm));
        atomicMax(&meta[3], __float_as_uint(mx));
        if (bias) atomicMax(&meta[2], __float_as_uint(fabsf(bias[row])));
    }
}

__global__ void __launch_bounds__(256)
pref_w_cast_kernel(const float* __restrict__ W, long long n, float* __restrict__ meta, _Float16* __restrict__ Wq) {
    const float sw = pow2_scale_for(meta[3]);
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid == 0) meta[0] = sw;
    if (gid < n) Wq[gid] = static_cast<_Float16>(W[gid] * sw);       // exact scaling, one RNE rounding
}

// one wave per hidden unit: distance between the row and its fp16 copy as the matrix core reads it
__global__ void __launch_bounds__(256)
pref_w_err_kernel(const float* __restrict__ W, int H, int D, const float* __restrict__ sw_ptr, unsigned* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= H) return;
    const float sw = *sw_ptr;
    if (!(sw > 0.f)) return;                                        // non-finite weights: every row is flagged anyway
    const float back = 1.0f / sw;
    float ff = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float w = W[static_cast<int64_t>(row) * D + d];
        const float e = fp16_input_error(w, w * sw, back);
        ff = fmaf(e, e, ff);
    }
    for (int off = 32; off > 0; off >>= 1) ff += __shfl_xor(ff, off, 64);
    if (lane == 0) atomicMax(out, __float_as_uint(sqrtf(ff) * 1.0001f));
}

// one wave per activation row: fp16 copy scaled by a per-row power of two, 1/(sx*sw), margin = 2*eps_b.
// NV > 0: D = 256 NV, the row stays in registers between the two passes (NV 16-byte loads per lane, read once);
// NV = 0: any D, second pass from L1 / L2.
template <int NV>
__global__ void __launch_bounds__(256)
pref_x_prep_kernel(const float* __restrict__ x, int B, int D, const float* __restrict__ meta,
                   _Float16* __restrict__ xq, float* __restrict__ inv, float* __restrict__ margin, int* __restrict__ zero_word) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;   // the call's flagged-row counter (saves a memset launch)
    if (row >= B) return;
    const float* xr = x + static_cast<int64_t>(row) * D;
    _Float16* qr = xq + static_cast<int64_t>(row) * D;
    float mx = 0.f, ss = 0.f;
    f32x4 keep[NV > 0 ? NV : 1];
    if (NV > 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) keep[j] = *reinterpret_cast<const f32x4*>(xr + 256 * j + 4 * lane);
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = keep[j][e], a = fabsf(v);
                mx = (a > mx || a != a) ? a : mx;
                ss = fmaf(v, v, ss);
            }
    } else {
        for (int d = lane; d < D; d += 64) {
            const float v = xr[d];
            const float a = fabsf(v);
            mx = (a > mx || a != a) ? a : mx;
            ss = fmaf(v, v, ss);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(mx, off, 64);
        mx = (o > mx || o != o) ? o : mx;
        ss += __shfl_xor(ss, off, 64);
    }
    // the fp16 copy and its distance from the row
    const float sx0 = pow2_scale_for(mx), back = sx0 > 0.f ? 1.0f / sx0 : 0.f;
    float ee = 0.f;
    if (NV > 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 q;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = keep[j][e];
                q[e] = static_cast<_Float16>(v * sx0);
                const float er = fp16_input_error(v, v * sx0, back);
                ee = fmaf(er, er, ee);
            }
            *reinterpret_cast<f16x4*>(qr + 256 * j + 4 * lane) = q;
        }
    } else {
        for (int d = lane; d < D; d += 64) {
            const float v = xr[d];
            qr[d] = static_cast<_Float16>(v * sx0);
            const float e = fp16_input_error(v, v * sx0, back);
            ee = fmaf(e, e, ee);
        }
    }
    for (int off = 32; off > 0; off >>= 1) ee += __shfl_xor(ee, off, 64);
    float sx, iv, mg;
    pref_row_params(mx, ss, ee, D, meta[0], meta[1], meta[2], meta[3], sx, iv, mg);
    if (lane == 0) {
        inv[row] = iv;
        margin[row] = mg;
    }
}

static void launch_x_prep(const float* x, int B, int D, const float* meta, _Float16* xq, float* inv, float* margin, hipStream_t s,
                          int* zero_word = nullptr) {
    const dim3 grid((B + 3) / 4), block(256);
    const bool vec = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(xq) % 8 == 0);
    if (vec && D == 512) hipLaunchKernelGGL(pref_x_prep_kernel<2>, grid, block, 0, s, x, B, D, meta, xq, inv, margin, zero_word);
    else if (vec && D == 256) hipLaunchKernelGGL(pref_x_prep_kernel<1>, grid, block, 0, s, x, B, D, meta, xq, inv, margin, zero_word);
    else if (vec && D == 1024) hipLaunchKernelGGL(pref_x_prep_kernel<4>, grid, block, 0, s, x, B, D, meta, xq, inv, margin, zero_word);
    else hipLaunchKernelGGL(pref_x_prep_kernel<0>, grid, block, 0, s, x, B, D, meta, xq, inv, margin, zero_word);
}

// pilot epilogue: approximate dense latents of the first P hidden units, rows = activations (registers),
// columns = hidden units (lanes): out[b][h] = fma(acc, inv[b], bias[h])
template <int BM, int BN, int WMW, int WNW>
struct EpiApproxDense {
    static constexpr int WTM = BM / WMW, WTN = BN / WNW, MT = WTM / 32, NT = WTN / 32;
    static constexpr int kCheckpoints = 0;
    static constexpr int kLdsFloats = 0;
    static constexpr int kStoresPerFinish = 0;
    struct Args {
        const float* inv;
        const float* bias;
        float* out;
        int64_t ld;
    };
    __device__ __forceinline__ void begin(const Args&, const TileCtx&) {}
    __device__ __forceinline__ void end(const Args&, const TileCtx&) {}
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
        float bcol[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = c.n0 + c.wn * WTN + nt * 32 + c.lane_col;
            bcol[nt] = (a.bias && col < c.N) ? a.bias[col] : 0.0f;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c.m0 + c.wm * WTM + mt * 32 + mfma_row(r, c.lane_half);
                if (row >= c.M) continue;
                const float iv = a.inv[row];
                float* orow = a.out + static_cast<int64_t>(row) * a.ld;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = c.n0 + c.wn * WTN + nt * 32 + c.lane_col;
                    if (col < c.N) orow[col] = fmaf(acc[mt][nt][r], iv, bcol[nt]);
                }
            }
    }
};

// ---- refine: approximate k-th -> survivors -> exact fp32 chain -> exact top-k; one wave per row --------
// The kernel is latency- and issue-bound, not bandwidth-bound (s_memtime stamps, tools/prof_refine_phases.py:
// 72 us per row and wave, half of it outside the W gather), so it is written for short code and few round
// trips: row scalars (count, tau, margin) and the activation row come through scalar loads (the row is the
// wave-uniform operand of every FMA: v_fmac with an SGPR source, no LDS copy); the candidate list is loaded
// with all slots in flight; the approximate k-th largest is a 32-bit bisection on the monotone value keys
// (only its VALUE is needed, ties are irrelevant); kRefSets W blocks stay in flight during the chains (the
// LDS hand-offs inside a wave need no fence -- one wave's LDS operations execute in order -- and a fence
// would drain the load queue).
constexpr int kRefWaves = 4;
constexpr int kRefMaxD = 2048;
constexpr int kRefSets = 3;        // W blocks in flight per wave
constexpr int kRefMaxSurv = 256;   // survivors per row (more -> flagged, exact fallback)
constexpr int kSelInFlight = 6;    // candidate-list slots per lane loaded together (refine_select_row)
constexpr int kRefTileStride = 36; // floats per transposed-tile row (32 + 4 pad: conflict-free b128 access)
// dynamic LDS per wave: exact keys [512] u64 | transposed W tile [64][36] | hidden index / value [512]
__host__ __device__ static inline size_t ref_lds_per_wave(int) {
    return static_cast<size_t>(kRefMaxSurv) * 8 + 64 * kRefTileStride * 4 + kRefMaxSurv * 4;
}

// One pass of the exact chains: survivors j0 .. j0 + 63 on lanes 0..63 and, in the two-chain form (kDual), survivors
// j0 + 64 .. j0 + 64 + nx - 1 (nx <= 8) as a second chain of lanes 0 .. nx - 1.
// A chain is sequential in k, so one lane owns one survivor; but 64 lanes walking 64 different W rows 16 bytes at a time touch
// 64 cache lines per load.  Instead the wave fetches [64 survivors x 32 k] blocks line-wise (8 lanes per 128-byte row segment),
// transposes them through LDS, and every lane then reads its own row's 32 values from there: each W line is fetched once.
// Two-chain form: a ninth line-load per block fetches the eight extra rows into tile rows 64..71 (kept in the exact-key array
// behind entry kRefDualKeys, which no survivor of this pass writes before the chains are done); the scalar activation loads, the
// LDS hand-offs and the gather round trips of the block are shared by both chains.  k = 64 leaves ~69 survivors per row: without
// this the five beyond the 64th cost a second pass as long as the first.
constexpr int kRefDualExtra = 8;
constexpr int kRefDualKeys = 72;     // first exact-key slot the extra tile rows may overlay (this pass writes keys 0..71 only)
static_assert((kRefMaxSurv - kRefDualKeys) * 8 >= kRefDualExtra * kRefTileStride * 4, "extra tile rows must fit behind the keys");
static_assert((kRefDualKeys * 8) % 16 == 0, "extra tile rows are read with b128");

template <bool kCounted, int kAbl, bool kDual>
__device__ __forceinline__ void refine_chain_pass(int j0, int m, int nx, int lane, int* hidx, float* wt, float* wt_x,
                                                  unsigned long long* ekey, const float* __restrict__ W,
                                                  const float* __restrict__ bias,
                                                  const __attribute__((address_space(4))) f32x4* xrow, int D, int ablate,
                                                  float tau_b, float margin_b) {
    constexpr int NL = kDual ? 9 : 8;                 // line-loads per block and lane
    constexpr int kSets = kDual ? 2 : kRefSets;       // W blocks in flight (two-chain form: two sets of nine, the registers of three of eight)
    auto lds_handoff = [&]() { asm volatile("" ::: "memory"); };
    const int nblk = D / 32;
    const int j = j0 + lane;
    const int h = (j < m) ? hidx[j] : hidx[j0];
    const int j2 = j0 + 64 + lane;
    const int h2 = (kDual && lane < nx) ? hidx[j2] : h;
    float acc = bias ? bias[h] : 0.0f;
    float acc2 = (kDual && bias) ? bias[h2] : 0.0f;
    // rows of the line-loads this lane takes part in: rows 8i + lane/8 of the group, as byte offsets into W
    // (32 bits in the counted form -- the launcher checks 4 H D < 2^32 --, which is also 8 registers less)
    typename std::conditional<kCounted, uint32_t, int64_t>::type voff[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        int jj = j0 + 8 * i + (lane >> 3);
        jj = (i < 8 ? jj < m : (lane >> 3) < nx) ? jj : j0;
        // (timing experiments: 1 = eight fixed rows, L1 hits; 7 = every XCD gathers from 1024 rows of its own, L2 hits)
        const int row = ablate == 1 ? (lane >> 3) : ablate == 7 ? ((hidx[jj] & 1023) | ((blockIdx.x & 7) << 10)) : hidx[jj];
        if (kCounted) voff[i] = static_cast<uint32_t>(row) * static_cast<uint32_t>(D * 4) + 16u * (lane & 7);
        else voff[i] = static_cast<int64_t>(row) * (D * 4) + 16 * (lane & 7);
    }
    const char* wbase = reinterpret_cast<const char*>(W);
    auto visible_load = [&](f32x4 (&sv)[NL], int blk) {     // loads the compiler sees (and waits for by its own count)
#pragma unroll
        for (int i = 0; i < NL; ++i) sv[i] = *reinterpret_cast<const f32x4*>(wbase + voff[i] + 128 * blk);
    };
    f32x4 st[kSets][NL];
    auto consume = [&](const f32x4 (&sv)[NL], int t) {
        f32x4 xv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (kAbl == 3 || kAbl == 6) xv[q] = f32x4{tau_b, margin_b, tau_b, margin_b};
            else xv[q] = xrow[8 * t + q];
        }
        f32x4 w[8];
        if (kAbl == 4 || kAbl == 6) {
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = sv[q];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<f32x4*>(wt + (8 * i + (lane >> 3)) * kRefTileStride + 4 * (lane & 7)) = sv[i];
            if (kDual) *reinterpret_cast<f32x4*>(wt_x + (lane >> 3) * kRefTileStride + 4 * (lane & 7)) = sv[NL - 1];
            lds_handoff();
            const float* mine = wt + lane * kRefTileStride;
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = *reinterpret_cast<const f32x4*>(mine + 4 * q);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc = fmaf(xv[q][0], w[q][0], acc);
            acc = fmaf(xv[q][1], w[q][1], acc);
            acc = fmaf(xv[q][2], w[q][2], acc);
            acc = fmaf(xv[q][3], w[q][3], acc);
        }
        if (kDual) {
            if (!(kAbl == 4 || kAbl == 6)) {
                const float* mine2 = wt_x + (lane & 7) * kRefTileStride;     // lanes >= nx: a valid row, result unused
#pragma unroll
                for (int q = 0; q < 8; ++q) w[q] = *reinterpret_cast<const f32x4*>(mine2 + 4 * q);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc2 = fmaf(xv[q][0], w[q][0], acc2);
                acc2 = fmaf(xv[q][1], w[q][1], acc2);
                acc2 = fmaf(xv[q][2], w[q][2], acc2);
                acc2 = fmaf(xv[q][3], w[q][3], acc2);
            }
        }
        lds_handoff();
    };
    int t = 0;
    if (kCounted) {
        // Counted form (nblk >= 2 kSets).  The compiler's own wait counting gives up on this loop: with the refills
        // inside it, it puts vmcnt(0) in front of the first block of every round, so each round waits for the set
        // issued LAST at full latency -- about one set in flight per wave instead of kSets.  Here the loads of
        // the prologue and of the main loop are inline asm (invisible to that bookkeeping; base in SGPRs, 32-bit
        // lane offsets) and so are the waits: loads retire in issue order, set q is always followed by exactly
        // kSets - 1 younger sets, so vmcnt(NL (kSets - 1)) in front of a block means "this set has landed".
        // The wait statement names the set's registers as read-write operands: every use of the data depends on
        // it.  The refills are unconditional (main loop: rounds whose refills all exist), so no value defined by an
        // asm load meets another definition at a join (a copy there would read the register before the data lands).
        auto issue = [&](f32x4 (&sv)[NL], int blk) {
            const char* sb = wbase + 128 * blk;             // wave-uniform
#pragma unroll
            for (int i = 0; i < NL; ++i)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(sv[i]) : "v"(voff[i]), "s"(sb));
        };
        auto landed = [&](f32x4 (&sv)[NL]) {
            static_assert(kRefSets == 3, "wait counts below");
            if (kDual)              // two sets of nine
                asm volatile("s_waitcnt vmcnt(9)" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]),
                             "+v"(sv[5]), "+v"(sv[6]), "+v"(sv[7]), "+v"(sv[NL - 1]));
            else                    // three sets of eight
                asm volatile("s_waitcnt vmcnt(16)" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]),
                             "+v"(sv[5]), "+v"(sv[6]), "+v"(sv[7]));
        };
        // the loads the compiler does know about (the bias) have to be retired in front of the asm loads: its wait
        // for them would otherwise sit at the first use inside the loop, as vmcnt(0), in every round
        asm volatile("" : "+v"(acc), "+v"(acc2));
#pragma unroll
        for (int q = 0; q < kSets; ++q) issue(st[q], q);
        for (; t + 2 * kSets <= nblk; t += kSets) {
#pragma unroll
            for (int q = 0; q < kSets; ++q) {
                if (kAbl != 5) landed(st[q]);
                consume(st[q], t + q);
                if (kAbl != 5) issue(st[q], t + q + kSets);
            }
        }
        // everything issued so far has to land before the last rounds (their refills are ordinary loads again)
#pragma unroll
        for (int q = 0; q < kSets; ++q) {
            if (kDual)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(st[q][0]), "+v"(st[q][1]), "+v"(st[q][2]), "+v"(st[q][3]),
                             "+v"(st[q][4]), "+v"(st[q][5]), "+v"(st[q][6]), "+v"(st[q][7]), "+v"(st[q][NL - 1]));
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(st[q][0]), "+v"(st[q][1]), "+v"(st[q][2]), "+v"(st[q][3]),
                             "+v"(st[q][4]), "+v"(st[q][5]), "+v"(st[q][6]), "+v"(st[q][7]));
        }
    } else {
#pragma unroll
        for (int q = 0; q < kSets; ++q)
            if (q < nblk) visible_load(st[q], q);
    }
    for (; t < nblk; t += kSets) {
#pragma unroll
        for (int q = 0; q < kSets; ++q) {
            if (t + q < nblk) {
                consume(st[q], t + q);
                if (t + q + kSets < nblk) visible_load(st[q], t + q + kSets);
            }
        }
    }
    lds_handoff();
    if (j < m) {
        ekey[j] = full_key(acc, static_cast<uint32_t>(h));
        // keep the exact bits next to the key (NaN payloads / -0 are not recoverable from the key)
        reinterpret_cast<float*>(hidx)[j] = acc;     // hidx[j] is consumed; reuse the slot for the value
    }
    if (kDual && lane < nx) {
        ekey[j2] = full_key(acc2, static_cast<uint32_t>(h2));
        reinterpret_cast<float*>(hidx)[j2] = acc2;
    }
    lds_handoff();
}

// Front half of the refinement of one row (one wave): the row's candidate list -> LDS, the approximate k-th largest value, the
// cut, the survivors' hidden indices into hidx[0 .. m) (list order).  Returns m, or -1 if the row was handed to the exact kernels
// (flag_row called).  `wt` is the wave's W-tile space (>= 2 kCandCap words), used for the staged list.
// kIdxInLds = false (single-part lists only): the hidden indices are not staged; the ~70 survivors fetch theirs from the list
// again (L2 hits) and the wave needs 5 KiB of LDS instead of 7 -- the select launch is a latency chain, waves per CU are its rate.
template <bool kIdxInLds = true, class FlagFn, class StampFn>
__device__ __forceinline__ int refine_select_row(const uint2* __restrict__ cand, const int* __restrict__ cnt, int cap,
                                                 const float* __restrict__ tau, const float* __restrict__ margin, int B, int H, int k,
                                                 int parts, const int* __restrict__ cnt_parts, int b, int lane, float* wt, int* hidx,
                                                 FlagFn flag_row, StampFn stamp, float& tau_b_out, float& margin_b_out) {
    auto lds_handoff = [&]() { asm volatile("" ::: "memory"); };       // in-order LDS queue: compiler barrier only
    // row scalars through the constant address space (written by earlier launches only): s_load, no VGPRs
    typedef const __attribute__((address_space(4))) int* cint_t;
    typedef const __attribute__((address_space(4))) float* cflt_t;
    // the row's list is `parts` segments of cap/parts entries (one per hidden-range part of the sweep)
    const int cap_part = cap / parts;
    int n = 0;
    bool seg_overflow = false;
    for (int p = 0; p < parts; ++p) {
        const int np = p == 0 ? ((cint_t)cnt)[b] : ((cint_t)cnt_parts)[static_cast<size_t>(p - 1) * B + b];
        seg_overflow |= np > cap_part;
        n += np;
    }
    const float tau_b = ((cflt_t)tau)[b];
    const float margin_b = ((cflt_t)margin)[b];
    if (n < k || seg_overflow) { flag_row(); return -1; }
    // ---- candidate list -> LDS (the W tile's space: value keys [1024] | hidden indices [1024]) ----------
    // Keys live in LDS, not in 16 register slots per lane: short loops instead of 4000 lines of unrolled
    // select code, and the registers go to the W staging sets.
    const uint2* list = cand + static_cast<int64_t>(b) * cap;
    const int nslots = (n + 63) / 64;                                  // wave-uniform
    uint32_t* lkey = reinterpret_cast<uint32_t*>(wt);                  // 0 = no candidate (mono keys are >= 0x007FFFFF)
    uint16_t* lidx = reinterpret_cast<uint16_t*>(lkey + kCandCap);   // hidden indices fit 16 bits (H <= 65536, use_fused); a larger
                                                                     // one has flagged the row (any_nan) before it is read back
    static_assert(2 * kCandCap * 4 <= 64 * kRefTileStride * 4, "candidate keys must fit the W tile");
    bool any_nan = false;
    uint32_t all_or = 0u, all_and = 0xFFFFFFFFu;
    int filled = 0;                                                    // entries staged so far (wave-uniform)
    for (int p = 0; p < parts; ++p) {
        const int np = p == 0 ? ((cint_t)cnt)[b] : ((cint_t)cnt_parts)[static_cast<size_t>(p - 1) * B + b];
        const uint2* seg = list + p * cap_part;
        for (int i0 = 0; i0 < np; i0 += 64 * kSelInFlight) {         // kSelInFlight list slots per lane in flight: one round trip
            uint2 c[kSelInFlight];                                     // for the usual ~300 entries, not five
#pragma unroll
            for (int u = 0; u < kSelInFlight; ++u) {
                const int i = i0 + 64 * u + lane;
                c[u] = i < np ? seg[i] : uint2{0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < kSelInFlight; ++u) {
                const int i = i0 + 64 * u + lane;
                if (i < np) {
                    const float v = __uint_as_float(c[u].x);
                    const uint32_t kk = mono_key(v);
                    any_nan |= (v != v) || c[u].y >= static_cast<uint32_t>(H);   // (a hidden index outside the dictionary: never gather with it)
                    all_or |= kk;
                    all_and &= kk;
                    lkey[filled + i] = kk;
                    if (kIdxInLds) lidx[filled + i] = static_cast<uint16_t>(c[u].y);
                }
            }
        }
        filled += np;
    }
    if (n + lane < nslots * 64) lkey[n + lane] = 0u;                   // padding of the last slot
    if (__any(any_nan)) { flag_row(); return -1; }                        // NaN latents: let the exact path rank them
    lds_handoff();
    stamp(0);
    // ---- approximate k-th largest VALUE: MSB-first bisection below the highest differing bit -------------
    for (int off = 32; off > 0; off >>= 1) {
        all_or |= __shfl_xor(all_or, off, 64);
        all_and &= __shfl_xor(all_and, off, 64);
    }
    const uint32_t diff = all_or ^ all_and;
    uint32_t T = all_and & ~(diff ? (0xFFFFFFFFu >> __builtin_clz(diff)) : 0u);   // common prefix
    int at_or_above = n;
    // Only the cut's VALUE matters and only to a fraction of the margin: key bits that move it by less than margin / 8 are
    // left at 0 (T stays a key with at least k candidates at or above it, so the cut only moves DOWN, by < margin / 8: a few
    // more survivors at worst, never a missing one).  With positive keys a step of 2^b in the key is 2^b ulps of at most the
    // largest candidate: b <= exponent(margin) - 3 - (exponent(largest) - 23).  Typically 10 of ~23 bisection rounds go.
    int lowbit = 0;
#ifndef QSAE_AB_FULL_BISECT
    if (all_and & 0x80000000u) {
        const int e_top = static_cast<int>((all_or >> 23) & 0xFFu), e_m = static_cast<int>((__float_as_uint(margin_b) >> 23) & 0xFFu);
        lowbit = e_m - e_top + 20;
        lowbit = lowbit < 0 ? 0 : lowbit > 22 ? 22 : lowbit;
    }
#endif
    for (int bit = diff ? 31 - __builtin_clz(diff) : -1; bit >= lowbit; --bit) {
        if (at_or_above == k) break;
        const uint32_t trial = T | (1u << bit);
        int c = 0;
        for (int s = 0; s < nslots; ++s) c += __popcll(__ballot(lkey[s * 64 + lane] >= trial));
        if (c >= k) { T = trial; at_or_above = c; }
    }
    stamp(1);
    // t~ = smallest approximate key inside the approximate top-k (the k-th largest when at_or_above == k,
    // otherwise the tie key T itself); keys are monotone in the value, so min over keys = min over values
    uint32_t tkey = 0xFFFFFFFFu;
    for (int s = 0; s < nslots; ++s) {
        const uint32_t kk = lkey[s * 64 + lane];
        if (kk >= T && kk < tkey) tkey = kk;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(tkey, off, 64);
        tkey = o < tkey ? o : tkey;
    }
    // key -> value (inverse of mono_key on non-NaN keys)
    const float tk = __uint_as_float((tkey & 0x80000000u) ? (tkey & 0x7FFFFFFFu) : ~tkey);
    // the list holds everything >= tau - margin; t~ must not lie below tau or survivors could be missing
    if (!(tk >= tau_b)) { flag_row(); return -1; }
    const uint32_t cutkey = mono_key(tk - margin_b);                   // keep <=> !(value < cut) <=> key >= cutkey
    // ---- survivors -> LDS ----------------------------------------------------------------------------
    int m = 0;
    for (int s = 0; s < nslots; ++s) {
        const int i = s * 64 + lane;
        const uint32_t kk = lkey[i];
        const bool keep = kk != 0u && kk >= cutkey;
        const unsigned long long msk = __ballot(keep);
        if (keep) {
            const int pos = m + __popcll(msk & ((1ull << lane) - 1ull));
            if (pos < kRefMaxSurv) hidx[pos] = kIdxInLds ? static_cast<int>(lidx[i]) : static_cast<int>(list[i].y);
        }
        m += __popcll(msk);
    }
    if (m > kRefMaxSurv) { flag_row(); return -1; }
    tau_b_out = tau_b;
    margin_b_out = margin_b;
    return m;
}

// Back half: exact keys ekey[0 .. m) and exact values (as floats in hidx[0 .. m)) -> exact rank, the k winners to idx / val / the
// dense latent, and the row's reconstruction when a decoder is attached.  `wt` (the W tile's space) and `ekey` are reused.
// kDecode (what the launch's decoder can be, so that the rank launch carries one decoder's registers, not all of them):
// 0 any (dispatch at run time), 1 packed 4-bit fields, 2 packed 8-bit fields, 3 none
template <int kDecode = 0, class StampFn>
__device__ __forceinline__ void refine_rank_decode(unsigned long long* ekey, int* hidx, float* wt, int m, int k, int b, int lane,
                                                   int32_t* __restrict__ idx_out, float* __restrict__ val_out,
                                                   float* __restrict__ dense, int64_t dense_ld, int* __restrict__ flags,
                                                   const RowDecode& dec, StampFn stamp) {
    auto lds_handoff = [&]() { asm volatile("" ::: "memory"); };
    // ---- exact rank among the survivors ----------------------------------------------------------------
    // (with a decoder attached the winners are also kept in LDS, in the W tile's space, which is free by now)
    int* w_idx = reinterpret_cast<int*>(wt);
    float* w_val = reinterpret_cast<float*>(wt) + kRefMaxSurv;
    static_assert(2 * kRefMaxSurv * 4 <= 64 * kRefTileStride * 4, "winner arrays must fit the W tile");
    // A hidden unit listed twice would give two survivors one key and one rank: a winner slot would stay unwritten and
    // the decode below would gather with whatever it holds.  The sweep never lists a unit twice; a row whose list says
    // otherwise is handed to the exact kernels like any other row the lists cannot serve -- after the loop: what it has
    // written by then are exact values of true members of the top-k (a duplicate displaces one, it adds none), which the
    // exact kernels write again.
    bool twice = false;
    auto emit = [&](int j, unsigned long long mine, int rank, int same) {
        twice |= same != 1;
        if (rank < k) {
            const int32_t hi = static_cast<int32_t>(key_index(mine));
            const float vv = reinterpret_cast<const float*>(hidx)[j];
            idx_out[static_cast<int64_t>(b) * k + rank] = hi;
            val_out[static_cast<int64_t>(b) * k + rank] = vv;
            if (dense) dense[static_cast<int64_t>(b) * dense_ld + hi] = vv;      // latent * mask; zeros are already there
            if (dec.active()) {
                w_idx[rank] = hi;
                w_val[rank] = vv;
            }
        }
    };
    if (m <= 128) {
        // the usual case (k = 64: ~69 survivors): both of a lane's keys are ranked by ONE walk over the keys (one broadcast
        // LDS read per key serves both)
        const int j1 = 64 + lane;
        const unsigned long long mine0 = lane < m ? ekey[lane] : 0ull, mine1 = j1 < m ? ekey[j1] : 0ull;
        int rank0 = 0, same0 = 0, rank1 = 0, same1 = 0;
        // First on the value halves of the keys alone (32-bit compares): exact fp32 latents of one row are almost never equal, and
        // if no lane sees its value twice the ranks are final and no unit can be listed twice.  Otherwise: the 64-bit walk.
        const uint32_t* khi = reinterpret_cast<const uint32_t*>(ekey) + 1;          // high words, stride 2
        const uint32_t v0 = static_cast<uint32_t>(mine0 >> 32), v1 = static_cast<uint32_t>(mine1 >> 32);
        // (no equality counts in these walks: with rank = number of larger values, any tie lowers the sum of the ranks below
        // m (m - 1) / 2 -- a group of g equal values gets one rank instead of g consecutive ones -- so one wave reduction
        // afterwards tells whether the 64-bit walk is needed)
        if (m <= 64 + 8) {
            // one walk for the first 64 keys; a short tail (k = 64: ~5 keys beyond the 64th) is ranked by the whole wave, one
            // ballot per tail key and slot, instead of a second compare / add pair in every round of the walk
            for (int i = 0; i < m; ++i) rank0 += (khi[2 * i] > v0) ? 1 : 0;
            for (int e = 64; e < m; ++e) {
                const uint32_t ve = khi[2 * e];                          // broadcast read
                const int r = __popcll(__ballot(lane < m && v0 > ve)) + __popcll(__ballot(j1 < m && v1 > ve));
                if (j1 == e) rank1 = r;
            }
        } else {
            for (int i = 0; i < m; ++i) {
                const uint32_t other = khi[2 * i];
                rank0 += (other > v0) ? 1 : 0;
                rank1 += (other > v1) ? 1 : 0;
            }
        }
        int rsum = (lane < m ? rank0 : 0) + (j1 < m ? rank1 : 0);
        for (int off = 32; off > 0; off >>= 1) rsum += __shfl_xor(rsum, off, 64);
        same0 = same1 = 1;
        const bool tied = rsum != m * (m - 1) / 2;
        if (__any(tied)) {
            rank0 = same0 = rank1 = same1 = 0;
            for (int i = 0; i < m; ++i) {
                const unsigned long long other = ekey[i];
                rank0 += (other > mine0) ? 1 : 0;
                same0 += (other == mine0) ? 1 : 0;
                rank1 += (other > mine1) ? 1 : 0;
                same1 += (other == mine1) ? 1 : 0;
            }
        }
        if (lane < m) emit(lane, mine0, rank0, same0);
        if (j1 < m) emit(j1, mine1, rank1, same1);
    } else {
        for (int j = lane; j < m; j += 64) {
            const unsigned long long mine = ekey[j];
            int rank = 0, same = 0;
            for (int i = 0; i < m; ++i) {
                const unsigned long long other = ekey[i];
                rank += (other > mine) ? 1 : 0;
                same += (other == mine) ? 1 : 0;
            }
            emit(j, mine, rank, same);
        }
    }
    if (__any(twice)) {
        if (lane == 0) {
            const int slot = atomicAdd(&flags[0], 1);
            flags[1 + slot] = b;
        }
        return;
    }
    stamp(5);
    // ---- sparse decode of this row (BinarySAE): winners into ascending index order, then the fmaf chain over the
    // k dictionary rows.  Same code as the stand-alone decode kernel; here its gathers and integer converts run in
    // the issue slots the other waves' chain gathers leave idle.
    if (kDecode != 3 && dec.active()) {
        lds_handoff();
        int* s_idx = reinterpret_cast<int*>(ekey);                       // the exact keys are no longer needed
        float* s_val = reinterpret_cast<float*>(ekey) + kRefMaxSurv;
        int mine_i[(kRefMaxSurv + 63) / 64], pos[(kRefMaxSurv + 63) / 64];
        float mine_v[(kRefMaxSurv + 63) / 64];
#pragma unroll
        for (int t = 0; t < (kRefMaxSurv + 63) / 64; ++t) {
            const int j = 64 * t + lane;
            pos[t] = -1;
            mine_i[t] = 0x7FFFFFFF;                                      // (no entry: above every hidden index)
            mine_v[t] = 0.0f;
            if (64 * t < k && j < k) {
                mine_i[t] = w_idx[j];
                mine_v[t] = w_val[j];
            }
        }
#pragma unroll
        for (int t = 0; t < (kRefMaxSurv + 63) / 64; ++t) {
            const int j = 64 * t + lane;
            const int tail = k - 64 * t;                                 // entries of this slot (wave-uniform)
            if (tail <= 0) continue;
            if (tail <= 8 && t > 0) {
                // a short last slot (k = 65: one entry): the wave counts for each of its entries together -- one ballot per slot
                // of held indices -- instead of walking all k entries with one lane active
                for (int e = 0; e < tail; ++e) {
                    const int he = w_idx[64 * t + e];                    // broadcast read
                    int c = 0;
#pragma unroll
                    for (int tt = 0; tt < (kRefMaxSurv + 63) / 64; ++tt)
                        if (64 * tt < k) c += __popcll(__ballot(mine_i[tt] < he));
                    if (lane == e) pos[t] = c;
                }
            } else if (j < k) {
                int p = 0;
                for (int i = 0; i < k; ++i) p += (w_idx[i] < mine_i[t]) ? 1 : 0;   // hidden indices are distinct
                pos[t] = p;
            }
        }
        lds_handoff();
#pragma unroll
        for (int t = 0; t < (kRefMaxSurv + 63) / 64; ++t)
            if (pos[t] >= 0) {
                s_idx[pos[t]] = mine_i[t];
                s_val[pos[t]] = mine_v[t];
            }
        lds_handoff();
#ifdef QSAE_AB_NARROW_DECODE
        decode_row_sorted_any<4>(s_idx, s_val, k, dec, b, lane);
#else
        if (kDecode == 1) decode_row_sorted_wide4(s_idx, s_val, k, dec, b, lane);
        else if (kDecode == 2) decode_row_sorted_wide<8>(s_idx, s_val, k, dec, b, lane);
        else decode_row_sorted_any_wide<4>(s_idx, s_val, k, dec, b, lane);
#endif
        stamp(6);
    }
}

// kAbl (debug library only, results wrong): 3 = no scalar loads of the activation row, 4 = no LDS transpose, 5 = no
// gathers in the main loop, 6 = 3 + 4
template <bool kCounted, int kAbl = 0>
__global__ void __launch_bounds__(64 * kRefWaves, 3)            // three workgroups per CU: <= 168 registers
refine_topk_kernel(const uint2* __restrict__ cand, const int* __restrict__ cnt, int cap, const float* __restrict__ tau,
                   const float* __restrict__ margin, const float* __restrict__ x, const float* __restrict__ W,
                   const float* __restrict__ bias, int B, int D, int H, int k, int32_t* __restrict__ idx_out,
                   float* __restrict__ val_out, int* __restrict__ flags, int ablate, unsigned long long* __restrict__ stamps,
                   float* __restrict__ dense, int64_t dense_ld, int parts, const int* __restrict__ cnt_parts, RowDecode dec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ref_smem[];
    // debug: per-phase cycle totals over all waves (stamps == nullptr in normal operation)
    // (one workgroup in 64 stamps: with every wave's atomics on the same eight words the stamped launch takes three times as long)
    if (stamps && (blockIdx.x & 63) != 0) stamps = nullptr;
    unsigned long long tprev = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    auto stamp = [&](int which) {
        if (stamps) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if ((threadIdx.x & 63) == 0) atomicAdd(&stamps[which], t - tprev);
            tprev = t;
        }
    };
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x * kRefWaves + wave;                       // wave-uniform
    if (b >= B) return;
    unsigned char* mybase = ref_smem + static_cast<size_t>(wave) * ref_lds_per_wave(D);
    unsigned long long* ekey = reinterpret_cast<unsigned long long*>(mybase);
    float* wt = reinterpret_cast<float*>(mybase + kRefMaxSurv * 8);
    int* hidx = reinterpret_cast<int*>(wt + 64 * kRefTileStride);
    auto flag_row = [&]() {
        if (lane == 0) {
            const int slot = atomicAdd(&flags[0], 1);
            flags[1 + slot] = b;
        }
    };
    auto lds_handoff = [&]() { asm volatile("" ::: "memory"); };       // in-order LDS queue: compiler barrier only
    float tau_b, margin_b;
    const int m = refine_select_row(cand, cnt, cap, tau, margin, B, H, k, parts, cnt_parts, b, lane, wt, hidx, flag_row, stamp, tau_b,
                                    margin_b);
    if (m < 0) return;
    lds_handoff();
    stamp(2);
    stamp(3);
    // ---- exact fp32 chain per survivor (ascending k, seeded with the bias: the oracle's arithmetic) ---
    // A chain is sequential in k, so one lane owns one survivor; but 64 lanes walking 64 different W rows
    // 16 bytes at a time touch 64 cache lines per load.  Instead the wave fetches [64 survivors x 32 k]
    // blocks line-wise (8 lanes per 128-byte row segment), transposes them through LDS, and every lane
    // then reads its own row's 32 values from there: each W line is fetched once.
    typedef const __attribute__((address_space(4))) f32x4* cvec_t;
    cvec_t xrow = (cvec_t)(x + static_cast<int64_t>(b) * D);          // wave-uniform: scalar loads
    float* wt_x = reinterpret_cast<float*>(ekey + kRefDualKeys);      // tile rows 64..71 of a two-chain pass (see there)
    for (int j0 = 0; j0 < (ablate == 2 ? 0 : m);) {
        // a first pass with a short tail behind it (k = 64: ~69 survivors) carries up to eight of the tail's chains as SECOND
        // chains of lanes 0..7 instead of leaving them a pass of their own
#ifndef QSAE_AB_NO_DUAL
        const int nx = (j0 == 0 && m > 64 && D / 32 >= 2 * 2) ? (m - 64 < kRefDualExtra ? m - 64 : kRefDualExtra) : 0;
#else
        const int nx = 0;
#endif
        if (nx > 0)
            refine_chain_pass<kCounted, kAbl, true>(j0, m, nx, lane, hidx, wt, wt_x, ekey, W, bias, xrow, D, ablate, tau_b, margin_b);
        else
            refine_chain_pass<kCounted, kAbl, false>(j0, m, 0, lane, hidx, wt, wt_x, ekey, W, bias, xrow, D, ablate, tau_b, margin_b);
        j0 += 64 + nx;
    }
    lds_handoff();
    stamp(4);
    if (ablate == 2) return;           // (timing experiment without the chains: the keys below were never written -- no outputs)
    refine_rank_decode(ekey, hidx, wt, m, k, b, lane, idx_out, val_out, dense, dense_ld, flags, dec, stamp);
}

// ---- the refinement regrouped by hidden slice: select -> slice-major chains -> rank / decode ------------------------
// The one-launch refinement above is bound by the rate at which the fabric delivers W rows: 69 survivors x 2 KiB per
// activation row, 64 MiB of W against 4 MiB of L2 per XCD, 25 % hits.  Here the exact chains run
// SLICE-MAJOR instead: the hidden units are cut into S slices of <= 4 MiB of W, XCD x owns slices x, x + 8, ..., and
// works through all rows' survivors of one slice before it touches the next, so a slice is fetched from the fabric once per
// XCD and every later gather of it is an L2 hit.  Three launches:
//   1. refine_select_kernel (one wave per row): list -> approximate k-th -> survivors (refine_select_row), sorted by hidden
//      index into the row's own candidate segment (the list is in LDS by then), plus offs[s][b] = number of the row's
//      survivors below slice s (one byte each, slice-major so that the chain kernel reads them coalesced).
//   2. refine_slice_chain_kernel: wave task = (slice, 128 rows).  The rows' entries of that slice are expanded into a queue
//      of (row, entry) pairs and taken 64 at a time, one chain per lane.  W rows AND activation rows are fetched line-wise
//      (8 lanes per 128-byte segment) and transposed through LDS; a lane reads its W row and its activation row (shared
//      with the neighbouring lanes of the same row) from there.  Same fmaf chain, ascending k, seeded with the bias: the
//      values are bit-identical to the row-major kernel's.  They go behind the sorted list in the row's segment.
//   3. refine_rank_kernel (one wave per row): exact keys from (value, index), rank, outputs, row decode (refine_rank_decode).
// The activation rows are re-read once per slice (S x 128 MiB, mostly L2 / memory-side-cache hits) in exchange for ~7 GB of
// W misses.  Prototype (tools/experiments/r03_slice_chain.hip, chains only, 69 survivors per row): S = 8 | 16 | 32:
// 0.84 | 0.76 | 0.82 ms, bound by the LDS traffic of the two transpositions (26 KiB per 64 pairs x 32 k).
constexpr int kSlList = 256;           // ints per row for the sorted survivor list; the exact values follow as kSlList floats
constexpr int kSlMaxSlices = 64;
constexpr int kSlRowsPerWave = 128;
constexpr int kSlXRows = 24;           // distinct activation rows per batch of 64 pairs (more: the batch is cut short)
constexpr int kSlQueue = 512;          // (row, entry) pairs per expansion round (16 bits each)
constexpr int kSlicedMinK = 48;       // below this the one-launch form is 2 % faster (k = 16, 32: one row-major pass per row); above, the sliced one (k = 65: 6 %, k = 128: 12 %)
constexpr int kSlicedMinRows = 8192;   // below this a slice's share of the rows does not fill the chip (tools/experiments/r03_sliced_batch_sizes.py)
static_assert(kSlList * 8 <= kCandCap * 8, "sorted list + values must fit the row's candidate segment");
constexpr int kSlSelectLds = kCandCap * 4 + kCandCap * 2 + kRefMaxSurv * 4;                          // per wave: keys | u16 indices | survivors
constexpr int kSlSelectLdsNoIdx = kCandCap * 4 + kRefMaxSurv * 4;                                  // per wave: keys | survivors
constexpr int kSlChainLds = 64 * 32 * 4 + kSlXRows * kRefTileStride * 4 + kSlQueue * 2 + 32 * 4;      // W tile | x tile | queue | x row ids
constexpr int kSlRankLds = kRefMaxSurv * 8 + 2 * kRefMaxSurv * 4 + kRefMaxSurv * 4;

template <bool kIdxInLds>     // false: single-part lists (large batches), 5 KiB of LDS per wave -> 32 waves per CU
__global__ void __launch_bounds__(64 * kRefWaves)
refine_select_kernel(uint2* __restrict__ cand, const int* __restrict__ cnt, int cap, const float* __restrict__ tau,
                     const float* __restrict__ margin, int B, int H, int k, int parts, const int* __restrict__ cnt_parts,
                     int* __restrict__ flags, int S, int per_shift, uint8_t* __restrict__ offs /* [S + 1][B] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sel_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x * kRefWaves + wave;
    if (b >= B) return;
    constexpr int kLds = kIdxInLds ? kSlSelectLds : kSlSelectLdsNoIdx;
    float* wt = reinterpret_cast<float*>(sel_smem + static_cast<size_t>(wave) * kLds);
    int* hidx = reinterpret_cast<int*>(sel_smem + static_cast<size_t>(wave) * kLds + (kIdxInLds ? kCandCap * 6 : kCandCap * 4));
    auto flag_row = [&]() {
        if (lane == 0) {
            const int slot = atomicAdd(&flags[0], 1);
            flags[1 + slot] = b;
        }
    };
    auto no_stamp = [](int) {};
    auto no_survivors = [&]() {                                        // a flagged row has nothing for the next two launches
        for (int s = lane; s <= S; s += 64) offs[static_cast<size_t>(s) * B + b] = 0;
    };
    float tau_b, margin_b;
    int m = refine_select_row<kIdxInLds>(cand, cnt, cap, tau, margin, B, H, k, parts, cnt_parts, b, lane, wt, hidx, flag_row, no_stamp,
                                         tau_b, margin_b);
    if (m > 255) { flag_row(); m = -1; }                               // (offsets are bytes)
    if (m < 0) { no_survivors(); return; }
    asm volatile("" ::: "memory");
    // ---- the list by slice.  The sweep appends a row's candidates stage by stage (64 hidden units each, ascending), so the
    // survivors normally arrive with their slices already in ascending runs and are stored as they are; if not (lists from
    // another producer, parts out of order), they are sorted by hidden index first.
    const int nsl = (m + 63) / 64;
    int mine[4];
    bool unordered = false;
    int last_slice = 0;                                                // slice of the entry in front of this slot (wave-uniform)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = 64 * t + lane;
        mine[t] = 0x7FFFFFFF;
        if (t < nsl) {
            if (j < m) mine[t] = hidx[j];
            const int sl = mine[t] >> per_shift;                       // (unused lanes: far beyond the last slice)
            int before = __shfl_up(sl, 1, 64);
            before = lane == 0 ? last_slice : before;
            unordered |= j < m && sl < before;
            last_slice = __builtin_amdgcn_readlane(sl, 63);
        }
    }
    int* list = reinterpret_cast<int*>(cand + static_cast<int64_t>(b) * cap);      // every entry of the segment has been read by now
    if (!__any(unordered)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < nsl && 64 * t + lane < m) list[64 * t + lane] = mine[t];
    } else {
        int pos[4];
        bool twice = false;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            pos[t] = -1;
            if (t < nsl && 64 * t + lane < m) {
                int below = 0, same = 0;
                for (int i = 0; i < m; ++i) {
                    const int o = hidx[i];
                    below += (o < mine[t]) ? 1 : 0;
                    same += (o == mine[t]) ? 1 : 0;
                }
                pos[t] = below;
                twice |= same != 1;
            }
        }
        if (__any(twice)) { flag_row(); no_survivors(); return; }     // a unit listed twice: positions collide
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (pos[t] >= 0) list[pos[t]] = mine[t];
    }
    const int per = 1 << per_shift;
    int myoff = 0;
    for (int s = 1; s < S; ++s) {
        const int lim = s * per;
        int c = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < nsl) c += __popcll(__ballot(mine[t] < lim));        // (unused slots hold INT_MAX)
        if (lane == s) myoff = c;
    }
    if (lane < S) offs[static_cast<size_t>(lane) * B + b] = static_cast<uint8_t>(myoff);
    if (lane == 0) offs[static_cast<size_t>(S) * B + b] = static_cast<uint8_t>(m);
}

// one batch of <= 64 (row, entry) pairs: lane l runs the chain of pair l; NXL = line-loads per block for the activation rows
// (8 rows each).  Registers and LDS are sized for THREE workgroups per CU (<= 168 VGPRs, 12.5 KiB per wave): the launch is
// bound by how many gathers the CU keeps in flight, not by any one pipe.  W tile [64][32] floats without padding, 16-byte
// chunk c of row r at chunk position c ^ (r & 7): the line-wise stores (8 lanes = one row) and the row-wise reads (8 lanes = 8
// consecutive rows, one chunk index) both touch every bank once.  Two sets of 8 + NXL loads in flight, counted waits.
template <int NXL>
__device__ __forceinline__ void slice_chain_batch(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                  uint2* __restrict__ cand, int cap, int D, int row0, uint32_t my, bool valid, int lane,
                                                  float* wt, float* xt, const int* xr, int R, int rx, int h) {
    const int b = row0 + static_cast<int>(my >> 8), ent = static_cast<int>(my & 255u);
    int* list = reinterpret_cast<int*>(cand + static_cast<int64_t>(b) * cap);
    float acc = bias ? bias[h] : 0.0f;                  // (in flight beside the first two sets below)
    uint32_t woff[8], xoff[NXL];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        woff[i] = static_cast<uint32_t>(__shfl(h, 8 * i + (lane >> 3), 64)) * static_cast<uint32_t>(D * 4) + 16u * (lane & 7);
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
        int tr = 8 * i + (lane >> 3);
        tr = tr < R ? tr : R - 1;
        xoff[i] = static_cast<uint32_t>(xr[tr]) * static_cast<uint32_t>(D * 4) + 16u * (lane & 7);
    }
    const char* wb = reinterpret_cast<const char*>(W);
    const char* xb = reinterpret_cast<const char*>(x);
    const int nblk = D / 32;
    constexpr int NL = 8 + NXL;                        // line-loads per block and lane
    constexpr int kSets = 2;                           // blocks in flight
    constexpr int kChunks = 2;                         // 16-byte chunks of the two tile rows read per step of a block
    f32x4 st[kSets][NL];
    // this lane's slots in the W tile: where its line-load chunks go, and where its own row's chunks are
    float* wput = wt + (lane >> 3) * 32 + 4 * ((lane & 7) ^ ((lane >> 3) & 7));          // + 8 i rows (256 floats) per load
    const float* wrow = wt + lane * 32;
    const int wkey = lane & 7;
    float* xput = xt + (lane >> 3) * kRefTileStride + 4 * (lane & 7);
    const float* xrow_t = xt + rx * kRefTileStride;
    auto visible_load = [&](f32x4 (&sv)[NL], int blk) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sv[i] = *reinterpret_cast<const f32x4*>(wb + woff[i] + 128 * blk);
#pragma unroll
        for (int i = 0; i < NXL; ++i) sv[8 + i] = *reinterpret_cast<const f32x4*>(xb + xoff[i] + 128 * blk);
    };
    auto consume = [&](const f32x4 (&sv)[NL]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(wput + 256 * i) = sv[i];
#pragma unroll
        for (int i = 0; i < NXL; ++i) *reinterpret_cast<f32x4*>(xput + 8 * kRefTileStride * i) = sv[8 + i];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int part = 0; part < 8 / kChunks; ++part) {  // (kChunks chunks at a time: 8 kChunks instead of 64 staging registers)
            f32x4 w[kChunks], xv[kChunks];
#pragma unroll
            for (int q = 0; q < kChunks; ++q) {
                w[q] = *reinterpret_cast<const f32x4*>(wrow + 4 * ((kChunks * part + q) ^ wkey));
                xv[q] = *reinterpret_cast<const f32x4*>(xrow_t + 4 * (kChunks * part + q));
            }
#pragma unroll
            for (int q = 0; q < kChunks; ++q) {
                acc = fmaf(xv[q][0], w[q][0], acc);
                acc = fmaf(xv[q][1], w[q][1], acc);
                acc = fmaf(xv[q][2], w[q][2], acc);
                acc = fmaf(xv[q][3], w[q][3], acc);
            }
            asm volatile("" : "+v"(acc) :: "memory");       // (the next step's reads stay behind this step's chain)
        }
    };
    // counted waits as in refine_chain_pass: asm loads (SGPR base, 32-bit lane offsets), loads retire in issue order, a set is
    // always followed by one younger set, so vmcnt(NL) means "this set has landed"
    auto issue = [&](f32x4 (&sv)[NL], int blk) {
        const char* sw = wb + 128 * blk;                // wave-uniform
        const char* sx = xb + 128 * blk;
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(sv[i]) : "v"(woff[i]), "s"(sw));
#pragma unroll
        for (int i = 0; i < NXL; ++i) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(sv[8 + i]) : "v"(xoff[i]), "s"(sx));
    };
#define QSAE_SL_REGS8 "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]), "+v"(sv[5]), "+v"(sv[6]), "+v"(sv[7])
    auto landed = [&](f32x4 (&sv)[NL]) {
        if (NXL == 1) asm volatile("s_waitcnt vmcnt(9)" : QSAE_SL_REGS8, "+v"(sv[8]));
        else if (NXL == 2) asm volatile("s_waitcnt vmcnt(10)" : QSAE_SL_REGS8, "+v"(sv[8]), "+v"(sv[NL - 1]));
        else asm volatile("s_waitcnt vmcnt(11)" : QSAE_SL_REGS8, "+v"(sv[8]), "+v"(sv[9]), "+v"(sv[NL - 1]));
    };
    auto all_landed = [&](f32x4 (&sv)[NL]) {
        if (NXL == 1) asm volatile("s_waitcnt vmcnt(0)" : QSAE_SL_REGS8, "+v"(sv[8]));
        else if (NXL == 2) asm volatile("s_waitcnt vmcnt(0)" : QSAE_SL_REGS8, "+v"(sv[8]), "+v"(sv[NL - 1]));
        else asm volatile("s_waitcnt vmcnt(0)" : QSAE_SL_REGS8, "+v"(sv[8]), "+v"(sv[9]), "+v"(sv[NL - 1]));
    };
#undef QSAE_SL_REGS8
    static_assert(NXL == 1 || NXL == 2 || NXL == 3, "wait counts above");
    int t = 0;
    if (nblk >= kSets) {
#pragma unroll
        for (int q = 0; q < kSets; ++q) issue(st[q], q);
        // the loads the compiler knows about (the bias, the caller's prefetch for the next batch) retire here, once, as vmcnt(0)
        // together with the two sets just issued -- not at the first use of acc inside the loop, in every round
        asm volatile("" : "+v"(acc));
        for (; t + 2 * kSets <= nblk; t += kSets) {
#pragma unroll
            for (int q = 0; q < kSets; ++q) {
                landed(st[q]);
                consume(st[q]);
                issue(st[q], t + q + kSets);
            }
        }
#pragma unroll
        for (int q = 0; q < kSets; ++q) all_landed(st[q]);
    } else {
#pragma unroll
        for (int q = 0; q < kSets; ++q)
            if (q < nblk) visible_load(st[q], q);
    }
    for (; t < nblk; t += kSets) {
#pragma unroll
        for (int q = 0; q < kSets; ++q) {
            if (t + q < nblk) {
                consume(st[q]);
                if (t + q + kSets < nblk) visible_load(st[q], t + q + kSets);
            }
        }
    }
    if (valid) reinterpret_cast<float*>(list)[kSlList + ent] = acc;
}
static_assert(kSlXRows == 24, "slice_chain_batch<3> fills exactly 24 tile rows");

__global__ void __launch_bounds__(64 * kRefWaves, 3)
refine_slice_chain_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                          uint2* __restrict__ cand, int cap, const uint8_t* __restrict__ offs, int B, int D, int S) {
    extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* base = chain_smem + static_cast<size_t>(wave) * kSlChainLds;
    float* wt = reinterpret_cast<float*>(base);
    float* xt = wt + 64 * 32;
    uint16_t* queue = reinterpret_cast<uint16_t*>(xt + kSlXRows * kRefTileStride);
    int* xr = reinterpret_cast<int*>(queue + kSlQueue);
    // workgroup g runs on XCD g mod 8 (round-robin dispatch); XCD x owns slices x, x + 8, ... and takes them one after the other
    const int g = blockIdx.x, xcd = g & 7, q = g >> 3;
    const int wgs_per_slice = (B + kSlRowsPerWave * kRefWaves - 1) / (kSlRowsPerWave * kRefWaves);
    const int slice = xcd + 8 * (q / wgs_per_slice);
    if (slice >= S) return;
    const int row0 = ((q % wgs_per_slice) * kRefWaves + wave) * kSlRowsPerWave;
    if (row0 >= B) return;
    int at[2], left[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int r = row0 + 64 * half + lane;
        at[half] = 0;
        left[half] = 0;
        if (r < B) {
            at[half] = offs[static_cast<size_t>(slice) * B + r];
            left[half] = static_cast<int>(offs[static_cast<size_t>(slice + 1) * B + r]) - at[half];
        }
    }
    while (__any(left[0] > 0 || left[1] > 0)) {
        // ---- expand the rows' entries of this slice into the queue (as many rounds as it takes) ----
        int total = 0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int c = left[half];
            int incl = c;
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            const int start = total + incl - c;
            int wrote = 0;
            for (int i = 0; i < c; ++i)
                if (start + i < kSlQueue) {
                    queue[start + i] = static_cast<uint16_t>(((64 * half + lane) << 8) | (at[half] + i));   // row 7 bits | entry 8 bits
                    ++wrote;
                }
            at[half] += wrote;
            left[half] -= wrote;
            total += __shfl(incl, 63, 64);
        }
        total = total < kSlQueue ? total : kSlQueue;
        asm volatile("" ::: "memory");
        // ---- batches of up to 64 pairs, cut short where the activation tile would overflow ----
        // the hidden index and the bias of a pair are two dependent loads in front of its chain: the indices of the NEXT batch are
        // loaded while this batch's chains run (the queue says which pairs come next), the bias beside the chain's first gathers
        auto pair_h = [&](uint32_t q) {
            const int* l = reinterpret_cast<const int*>(cand + static_cast<int64_t>(row0 + static_cast<int>(q >> 8)) * cap);
            return l[q & 255u];
        };
        int p0 = 0;
        int h_next = pair_h(queue[lane < total ? lane : 0]);
        while (p0 < total) {
            const int p = p0 + lane;
            const bool in = p < total;
            const uint32_t my = queue[in ? p : p0];
            const int h_now = h_next;
            const int rl = static_cast<int>(my >> 8);
            const int prev = __shfl_up(rl, 1, 64);
            const bool head = in && (lane == 0 || prev != rl);
            const unsigned long long hb = __ballot(head);
            int rx = __popcll(hb & ((2ull << lane) - 1ull)) - 1;      // tile row of this lane's activation row
            const unsigned long long over = __ballot(in && rx >= kSlXRows);
            const int take = over ? __builtin_ctzll(over) : (total - p0 < 64 ? total - p0 : 64);
            const bool valid = lane < take;
            const int R = __popcll(hb & (take >= 64 ? ~0ull : ((1ull << take) - 1ull)));
            if (head && valid) xr[rx] = row0 + rl;
            rx = valid ? rx : 0;
            asm volatile("" ::: "memory");
            {
                const int pn = p0 + take + lane;                        // the next batch starts at p0 + take
                const uint32_t qn = queue[pn < total ? pn : (p0 + take < total ? p0 + take : 0)];
                h_next = pair_h(qn);
            }
            if (R <= 8) slice_chain_batch<1>(x, W, bias, cand, cap, D, row0, my, valid, lane, wt, xt, xr, R, rx, h_now);
            else if (R <= 16) slice_chain_batch<2>(x, W, bias, cand, cap, D, row0, my, valid, lane, wt, xt, xr, R, rx, h_now);
            else slice_chain_batch<3>(x, W, bias, cand, cap, D, row0, my, valid, lane, wt, xt, xr, R, rx, h_now);
            asm volatile("" ::: "memory");
            p0 += take;
        }
    }
}

template <int kDecode>
__global__ void __launch_bounds__(64 * kRefWaves, kDecode == 0 ? 3 : 4)
refine_rank_kernel(const uint2* __restrict__ cand, int cap, const uint8_t* __restrict__ offs, int S, int B, int k,
                   int32_t* __restrict__ idx_out, float* __restrict__ val_out, int* __restrict__ flags, float* __restrict__ dense,
                   int64_t dense_ld, RowDecode dec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rank_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x * kRefWaves + wave;
    if (b >= B) return;
    const int m = offs[static_cast<size_t>(S) * B + b];
    if (m == 0) return;                                                // flagged by the select launch
    unsigned char* mybase = rank_smem + static_cast<size_t>(wave) * kSlRankLds;
    unsigned long long* ekey = reinterpret_cast<unsigned long long*>(mybase);
    float* wt = reinterpret_cast<float*>(mybase + kRefMaxSurv * 8);     // winners (2 kRefMaxSurv words)
    int* hval = reinterpret_cast<int*>(wt + 2 * kRefMaxSurv);
    const int* list = reinterpret_cast<const int*>(cand + static_cast<int64_t>(b) * cap);
    for (int j = lane; j < m; j += 64) {
        const int h = list[j];
        const float v = reinterpret_cast<const float*>(list)[kSlList + j];
        ekey[j] = full_key(v, static_cast<uint32_t>(h));
        reinterpret_cast<float*>(hval)[j] = v;
    }
    asm volatile("" ::: "memory");
    auto no_stamp = [](int) {};
    refine_rank_decode<kDecode>(ekey, hval, wt, m, k, b, lane, idx_out, val_out, dense, dense_ld, flags, dec, no_stamp);
}

// slices of 2^shift hidden units, at most 4 MiB of W each; their number a multiple of 8 (one per XCD and round; slices past H
// are empty).  false: more than kSlMaxSlices would be needed.
static bool sliced_plan(int H, int D, int* S, int* shift) {
    int sh = 0;
    while ((static_cast<size_t>(2) << sh) * D * 4 <= (4u << 20)) ++sh;              // largest 2^sh with 2^sh D 4 <= 4 MiB
    int n = (H + (1 << sh) - 1) >> sh;
    n = (n + 7) / 8 * 8;
    *S = n;
    *shift = sh;
    return n <= kSlMaxSlices;
}

static bool sliced_fits(int H, int D) {
    int S, sh;
    return sliced_plan(H, D, &S, &sh);
}

// Zero-fill of the dense latent by a kernel that runs BESIDE the sweep.  Carried by the sweep's own waves the 8.4 M
// 1-KiB stores cost it 0.5 ms: a wave that waits for a slot in the write queue cannot issue its next MFMA either.
// The sweep's no-fill build takes 248 VGPRs per wave, two waves per SIMD, which leaves 16 registers per SIMD -- room
// for one wave of this kernel (10 VGPRs, no LDS), whose stalls hold up nobody.  Single-wave workgroups, grid-stride
// over 1-KiB pieces (all waves together write one contiguous run per step), nontemporal stores, paced with s_sleep so
// that the fill ends when the sweep does (unpaced it finishes early and costs the sweep more while it runs).
// Same-process scans, ms per step (in-sweep fill: 4.75-4.92):  1024 waves x pace 3 | 4 | 5 | 6: 4.53 | 4.41 | 4.53 | 4.75;
// 768 x 2: 4.43; 640 x 1: 4.44; unpaced 384-448: 4.51; a first version with a 64-bit division per store (which paced
// it by accident), 640 waves: 4.41-4.55.  All land on 2.53-2.57 ms for the sweep / fill pair against 2.23 ms for the
// sweep alone: what is left is the memory system, not issue slots.  Round 2, sweep at 2.31 ms: pair time at 1024 waves x pace
// 2 | 3 | 4: 2.34 | 2.30 | 2.45 ms; 896 | 768 waves x pace 3: 2.43 | 2.60 ms (the pace has to follow the sweep).
__global__ void __launch_bounds__(64)
fill_zero_co_kernel(float* __restrict__ dense, long long ld, int rows, int ppr /* 1-KiB pieces per row */, int pace) {
    // piece p = (row r, 1-KiB column block c), p = blockIdx.x, += gridDim.x.  Everything but the lane offset is
    // wave-uniform and advanced incrementally (a 64-bit division per store would cost this kernel forty instructions
    // per store -- issue slots it takes from the sweep it runs beside).
    const int G = static_cast<int>(gridDim.x);
    const int dr = G / ppr, dc = G % ppr;
    int r = static_cast<int>(blockIdx.x) / ppr, c = static_cast<int>(blockIdx.x) % ppr;
    const long long row_bytes = ld * 4;
    const long long step_bytes = dr * row_bytes + static_cast<long long>(dc) * 1024;
    const long long wrap_bytes = row_bytes - static_cast<long long>(ppr) * 1024;
    long long off = r * row_bytes + static_cast<long long>(c) * 1024;
    char* base = reinterpret_cast<char*>(dense) + threadIdx.x * 16;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    while (r < rows) {
        __builtin_nontemporal_store(z, reinterpret_cast<f32x4*>(base + off));
        for (int i = 0; i < pace; ++i) __builtin_amdgcn_s_sleep(1);      // 64 cycles each: spreads the stores over the sweep's duration
        off += step_bytes;
        r += dr;
        c += dc;
        if (c >= ppr) {
            c -= ppr;
            r += 1;
            off += wrap_bytes;
        }
    }
}

// ~20 us of one sleeping wave in front of the fill kernel on the side stream: the sweep (same dependency, other
// stream) is resident on every CU by then.  Fill waves that arrived first could sit two to a SIMD and keep a sweep
// workgroup (496 of a SIMD's 512 registers) off that CU for the whole fill.
__global__ void __launch_bounds__(64) co_delay_kernel(int ticks /* of the 100 MHz real-time counter */) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < static_cast<unsigned long long>(ticks)) __builtin_amdgcn_s_sleep(32);
}

// The co-resident fill needs the register budgets above; if a rebuild changes them, fall back to the in-sweep fill.
static bool co_fill_fits() {
    static int fits = -1;
    if (fits < 0) {
        hipFuncAttributes fa_sweep{}, fa_fill{};
        const bool ok = hipFuncGetAttributes(&fa_sweep, reinterpret_cast<const void*>(sweep_xstat_f16_kernel<32, 9>)) == hipSuccess &&
                        hipFuncGetAttributes(&fa_fill, reinterpret_cast<const void*>(fill_zero_co_kernel)) == hipSuccess;
        fits = (ok && fa_sweep.numRegs <= 248 && fa_fill.numRegs <= 16) ? 1 : 0;
    }
    return fits == 1;
}

static bool prefilter_shape_ok(int B, int D, int H, int k) {
    return use_fused(B, D, H, k) && D % 64 == 0 && D <= kRefMaxD && (H - pilot_width(H)) > 0;
}

// Rank among the 32 group maxima (each over H/512 pilot units) that puts about max(5 k, 200) values of a row
// above tau: P(group max >= tau) = r/32 = 1 - F^(H/512)  =>  expected count H (1 - F) ~ -512 ln(1 - r/32).
static int inkernel_rank(int k) {
    const double target = 5.0 * k > 200.0 ? 5.0 * k : 200.0;    // k = 65: rank 15 (13-16 time alike; 15 flags the fewest rows)
    int r = static_cast<int>(32.0 * (1.0 - exp(-target / 512.0)) + 0.5);
    return r < 6 ? 6 : (r > 24 ? 24 : r);
}

// One prefilter call: the arguments of the entry point plus what follows from them and from the build's switches.
struct PrefCall {
    const float* x; const float* W; const float* bias; const _Float16* Wq; const float* meta;
    int B, D, H, k;
    int32_t* idx; float* val;
    char* ws; qsae_stream_t stream;
    float* dense; int64_t dense_ld;
    const RowDecode* dec;                 // BinarySAE: rows are decoded as they are ranked; nullptr = no reconstruction
};
struct PrefPlan {
    FusedLayout L; PrefLayout PL;
    int P;                                // pilot width
    bool xstat, inkernel, fill_co, fill_in_sweep;
    int Hs, hoff, parts, cap_part;
    float* filled;                        // the dense latent if its zeros are written during the sweep launch, else nullptr
};
static PrefPlan pref_plan(const PrefCall& c) {
    PrefPlan p;
    p.L = fused_layout(c.B, c.D, c.H, c.k);
    p.PL = pref_layout(c.B, c.D, p.L.total);
    p.P = pilot_width(c.H);
    // With the activation-stationary sweep nothing upstream touches the dense latent: its zeros are written during the
    // sweep launch (co-resident fill kernel, or the sweep's own waves) and the survivors by the refinement; without
    // either fill it is written once at the end (zeros + the k survivors of every row in one pass, densify_rows).  The
    // LDS-tiled sweep kernels zero-fill their own blocks in the epilogue and the survivors are scattered in afterwards.
    p.xstat = g_pref_tile == 2 && xstat_supported(c.D, c.H - p.P, p.P) && c.H % 4 == 0;
    // In-kernel pilot: the stationary sweep derives tau itself from a stratified H/16 sample of the hidden units (group
    // maxima, see sweep_xstat_f16.h) and then sweeps ALL hidden units; no pilot GEMM, no pilot buffer, no seeds.
    p.inkernel = p.xstat && g_inkernel_pilot && p.P % kXsHT == 0 && c.H % kXsHT == 0 && xstat_supported(c.D, c.H, 0);
    p.Hs = p.inkernel ? c.H : c.H - p.P;                     // hidden units the sweep launch covers
    p.hoff = p.inkernel ? 0 : p.P;
    // small batches: the hidden range of the sweep is split over `parts` workgroup columns, each with its own
    // segment of every row's candidate list
    p.parts = p.xstat ? (g_x_parts > 0 ? g_x_parts : xstat_parts(c.B, p.Hs, kCandCap)) : 1;
    p.cap_part = kCandCap / p.parts;
    p.fill_co = p.xstat && c.dense && g_fill_co && c.D == 512 && c.H % 256 == 0 && c.dense_ld % 4 == 0 && g_xstat_ablate == 0 &&
                co_fill_fits();
    p.fill_in_sweep = !p.fill_co && p.xstat && c.dense && g_fill_in_sweep && c.H % 256 == 0 && c.dense_ld % 4 == 0;
    p.filled = (p.fill_in_sweep || p.fill_co) ? c.dense : nullptr;
    return p;
}

// Steps 1-5: everything up to and including the refinement.  Afterwards flags[0] (device) holds the number of rows that
// need the exact fallback and flags[1..] their ids; every other row's outputs are final.
static int prefilter_submit(const PrefCall& c) {
    hipStream_t s = as_stream(c.stream);
    const SweepProfile prof = take_sweep_profile();
    const PrefPlan pl = pref_plan(c);
    const FusedLayout& L = pl.L;
    const PrefLayout& PL = pl.PL;
    const int B = c.B, D = c.D, H = c.H, k = c.k, P = pl.P;
    char* ws = c.ws;
    float* pilot = reinterpret_cast<float*>(ws + L.pilot);
    float* tau = reinterpret_cast<float*>(ws + L.tau);
    int* cnt = reinterpret_cast<int*>(ws + L.cnt);
    uint2* cand = reinterpret_cast<uint2*>(ws + L.cand);
    int* flags = reinterpret_cast<int*>(ws + L.flags);
    _Float16* xq = reinterpret_cast<_Float16*>(ws + PL.xq);
    float* inv = reinterpret_cast<float*>(ws + PL.inv);
    float* margin = reinterpret_cast<float*>(ws + PL.margin);
    int* cnt_parts = reinterpret_cast<int*>(ws + PL.cnt_parts);
    const bool xstat = pl.xstat, inkernel = pl.inkernel, fill_co = pl.fill_co, fill_in_sweep = pl.fill_in_sweep;
    float* fused_fill = xstat ? nullptr : c.dense;
    const int Hs = pl.Hs, hoff = pl.hoff, parts = pl.parts, cap_part = pl.cap_part;
    // activation-stationary sweep with its own fill: all H columns, spread over the iterations of every part (its
    // share of the sweep stages plus the pilot iterations)
    const int xs_iters = xstat ? (Hs / kXsHT) / parts + (inkernel ? P / kXsHT : 0) : 0;
    const int fill_cw = xs_iters > 0 ? (32 * (H / 256) / parts + xs_iters - 1) / xs_iters : 0;   // 1-KiB pieces per wave and iteration
    ThreadDeviceCtx* ctx = nullptr;
    if (fill_co) {
        const int rc0 = thread_device_ctx(&ctx);
        if (rc0 != QSAE_OK) return rc0;
    }
    // 1. fp16 copy of the batch + per-row scale and error margin (the stationary sweep with the in-kernel pilot can do
    //    this in its own prologue, straight into registers)
    const bool fuse_prep = inkernel && g_fuse_xprep;
    if (!fuse_prep && (g_x_phase & 1)) {
        launch_x_prep(c.x, B, D, c.meta, xq, inv, margin, s, flags);    // (also zeroes the flagged-row counter)
        QSAE_LAUNCH_CHECK();
    } else {
        QSAE_HIP(hipMemsetAsync(flags, 0, sizeof(int), s));
    }
    const int Kw = D / 2;                                    // 4-byte words per fp16 row
    const float* xq_w = reinterpret_cast<const float*>(xq);
    const float* wq_w = reinterpret_cast<const float*>(c.Wq);
    // 2. approximate pilot block [B][P] (activation rows on registers, hidden units on lanes)
    int rc = QSAE_OK;
    if (!inkernel) {
        if (g_pilot_tile == 0 && P % 256 == 0) {
            using EpiP = EpiApproxDense<256, 256, 4, 2>;
            typename EpiP::Args ep{inv, c.bias, pilot, P};
            rc = launch_gemm_dma<EpiP, 256, 256, true, 2>(xq_w, B, wq_w, P, Kw, ep, s, /*sweep=*/8);
        } else {
            using EpiP = EpiApproxDense<256, 128, 4, 2>;
            typename EpiP::Args ep{inv, c.bias, pilot, P};
            rc = launch_gemm_dma<EpiP, 256, 128, true>(xq_w, B, wq_w, P, Kw, ep, s, /*sweep=*/8);
        }
        if (rc != QSAE_OK) return rc;
    }
    // 3. tau~ = j-th largest approximate pilot value; seeds = pilot elements >= tau~ - 2 eps
    const int j = kPilotRank < P ? kPilotRank : P;
    if (!inkernel) rc = topk_rows_dispatch(pilot, P, B, P, j, nullptr, nullptr, 0, tau, cand, cnt, cap_part, fused_fill, c.dense_ld, s,
                                margin, kCandCap);
    if (rc != QSAE_OK) return rc;
    // 4. fp16 sweep of the remaining hidden units with the threshold filter (tau~ - 2 eps)
    if (g_x_phase & 1) {
        using EpiS = EpiFilter<256, 128, 4, 2, true>;
        typename EpiS::Args es{c.bias ? c.bias + P : nullptr, tau, cand, cnt, kCandCap, P, fused_fill, c.dense_ld, inv, margin};
        if (prof.begin) QSAE_HIP(hipEventRecord(prof.begin, s));
        if (xstat) {
            XsArgs xa{xq + 0, c.Wq + static_cast<size_t>(hoff) * D, c.bias ? c.bias + hoff : nullptr, tau, margin, inv, cand, cnt,
                      B, Hs, kCandCap, hoff, g_xstat_rot, g_xstat_stamps, fill_in_sweep ? c.dense : nullptr, c.dense_ld, H,
                      fill_cw, inkernel ? P / kXsHT : 0, g_inkernel_rank > 0 ? g_inkernel_rank : inkernel_rank(k), tau,
                      fuse_prep ? c.x : nullptr, c.meta, inv, margin, parts, cnt_parts};
            if (fill_co) QSAE_HIP(hipEventRecord(ctx->ev_fork, s));     // everything before the sweep (x prep, earlier users of `dense`)
            // (the build without fill code whenever this launch has no zeros to write itself)
            rc = launch_xstat(D, xa, s, (g_xstat_ablate == 0 && D == 512 && !fill_in_sweep) ? 9 : g_xstat_ablate);
            if (fill_co && rc == QSAE_OK) {
                // The zeros are written by a second kernel beside the sweep, on this thread's side stream for this
                // device: forked from `s` at ev_fork, joined back at ev_join (both events belong to this thread, and a
                // thread's calls are issued one after the other, so a later record cannot overtake an earlier wait).
                QSAE_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
                hipLaunchKernelGGL(co_delay_kernel, dim3(1), dim3(64), 0, ctx->side, 2000);     // 20 us
                QSAE_LAUNCH_CHECK();
                hipLaunchKernelGGL(fill_zero_co_kernel, dim3(g_fill_co > 1 ? g_fill_co % 10000 : kFillCoWaves), dim3(64), 0, ctx->side,
                                   c.dense, static_cast<long long>(c.dense_ld), B, H / 256, g_fill_co > 1 ? g_fill_co / 10000 : kFillCoPace);
                QSAE_LAUNCH_CHECK();
                QSAE_HIP(hipEventRecord(ctx->ev_join, ctx->side));
                QSAE_HIP(hipStreamWaitEvent(s, ctx->ev_join, 0));        // refine writes the survivors into the zeros
            }
        } else if (g_pref_tile != 1) {
            // 256 hidden x 256 activation rows per workgroup: 128 FLOP per staged byte (256 x 128: 85)
            using EpiW = EpiFilter<256, 256, 4, 2, true>;
            typename EpiW::Args ew{es.bias, es.tau, es.cand, es.cnt, es.cap, es.hidden_offset, es.dense, es.dense_ld, es.inv,
                                   es.margin};
            rc = launch_gemm_dma<EpiW, 256, 256, true, 2>(wq_w + static_cast<size_t>(P) * Kw, H - P, xq_w, B, Kw, ew, s);
        } else {
            rc = launch_gemm_dma<EpiS, 256, 128, true>(wq_w + static_cast<size_t>(P) * Kw, H - P, xq_w, B, Kw, es, s);
        }
        if (prof.end) QSAE_HIP(hipEventRecord(prof.end, s));
        if (rc != QSAE_OK) return rc;
        if (xstat && g_xstat_ablate != 0) return QSAE_OK;    // timing experiment: the lists are not trustworthy
    }
    // 5. survivors -> exact chain -> exact top-k (-> the row's reconstruction)
    const bool sliced = g_ref_sliced != 0 && g_ref_ablate == 0 && g_ref_stamps == nullptr && D % 32 == 0 && D / 32 >= 2 &&
                        static_cast<uint64_t>(H) * D * 4 < (1ull << 32) && static_cast<uint64_t>(B) * D * 4 < (1ull << 32) &&
                        (g_ref_sliced == 2 || (B >= kSlicedMinRows && k >= kSlicedMinK)) && sliced_fits(H, D);
    if ((g_x_phase & 2) && sliced) {
        // 5'. the same refinement as three launches, chains slice-major (see refine_select_kernel)
        int S = 0, per_shift = 0;
        sliced_plan(H, D, &S, &per_shift);
        uint8_t* offs = reinterpret_cast<uint8_t*>(ws + PL.sl_offs);
        QSAE_SET_MAX_LDS_ONCE(refine_select_kernel<true>, 160 * 1024);
        QSAE_SET_MAX_LDS_ONCE(refine_slice_chain_kernel, 160 * 1024);
        const dim3 rows_grid((B + kRefWaves - 1) / kRefWaves), block(64 * kRefWaves);
        if (!(g_x_phase & 8)) {                             // (debug library, timing experiments: 8 = the rank launch only, 4 = all but it)
            if (parts == 1)
                hipLaunchKernelGGL(refine_select_kernel<false>, rows_grid, block, kSlSelectLdsNoIdx * kRefWaves, s, cand, cnt, kCandCap,
                                   tau, margin, B, H, k, parts, cnt_parts, flags, S, per_shift, offs);
            else
                hipLaunchKernelGGL(refine_select_kernel<true>, rows_grid, block, kSlSelectLds * kRefWaves, s, cand, cnt, kCandCap, tau,
                                   margin, B, H, k, parts, cnt_parts, flags, S, per_shift, offs);
            QSAE_LAUNCH_CHECK();
            const int wgs_per_slice = (B + kSlRowsPerWave * kRefWaves - 1) / (kSlRowsPerWave * kRefWaves);
            hipLaunchKernelGGL(refine_slice_chain_kernel, dim3(8 * (S / 8) * wgs_per_slice), block, kSlChainLds * kRefWaves, s, c.x,
                               c.W, c.bias, cand, kCandCap, offs, B, D, S);
            QSAE_LAUNCH_CHECK();
        }
        if (g_x_phase & 4) return QSAE_OK;
        const RowDecode rd = c.dec ? *c.dec : RowDecode{nullptr, 0, 0, 0, 0, 0.f, nullptr, nullptr, nullptr};
        auto rank = refine_rank_kernel<0>;
        if (!rd.active()) rank = refine_rank_kernel<3>;
        else if (rd.packed && !rd.table && rd.fw == 4) rank = refine_rank_kernel<1>;
        else if (rd.packed && !rd.table && rd.fw == 8) rank = refine_rank_kernel<2>;
        hipLaunchKernelGGL(rank, rows_grid, block, kSlRankLds * kRefWaves, s, cand, kCandCap, offs, S, B, k, c.idx, c.val, flags,
                           pl.filled, c.dense_ld, rd);
    } else if (g_x_phase & 2) {
        const size_t lds = ref_lds_per_wave(D) * kRefWaves;
        // counted-wait form of the chains: at least kRefSets blocks of 32 per row, W addressable with 32-bit offsets
        const bool counted = D / 32 >= kRefSets && static_cast<uint64_t>(H) * D * 4 < (1ull << 32);
        auto kern = counted ? refine_topk_kernel<true> : refine_topk_kernel<false>;
#ifdef QSAE_DEBUG_BUILD
        if (counted && g_ref_ablate == 3) kern = refine_topk_kernel<true, 3>;
        if (counted && g_ref_ablate == 4) kern = refine_topk_kernel<true, 4>;
        if (counted && g_ref_ablate == 5) kern = refine_topk_kernel<true, 5>;
        if (counted && g_ref_ablate == 6) kern = refine_topk_kernel<true, 6>;
        if (g_ref_ablate >= 3) QSAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
        QSAE_SET_MAX_LDS_ONCE(refine_topk_kernel<true>, 160 * 1024);
        QSAE_SET_MAX_LDS_ONCE(refine_topk_kernel<false>, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3((B + kRefWaves - 1) / kRefWaves), dim3(64 * kRefWaves), lds, s, cand,
                           cnt, kCandCap, tau, margin, c.x, c.W, c.bias, B, D, H, k, c.idx, c.val, flags, g_ref_ablate, g_ref_stamps,
                           pl.filled, c.dense_ld, parts, cnt_parts,
                           c.dec ? *c.dec : RowDecode{nullptr, 0, 0, 0, 0, 0.f, nullptr, nullptr, nullptr});
    }
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

static FlaggedArgs pref_flagged_args(const PrefCall& c, const PrefPlan& pl) {
    // (with the zeros written during the sweep launch, refine and the fallback write the survivors straight into the latent)
    return FlaggedArgs{c.x, c.W, c.bias, c.B, c.D, c.H, c.k, c.idx, c.val, c.ws, pl.L, c.stream, /*kperm=*/false,
                       pl.filled, c.dense_ld};
}

// Step 6, with the host knowing the flagged-row count: exact fallback for flagged rows [first, nflag) (rows below
// `first` were handled by flagged_spec), their reconstruction, and the dense latent where nothing has written it yet.
static int prefilter_finish(const PrefCall& c, int first, int nflag) {
    hipStream_t s = as_stream(c.stream);
    const PrefPlan pl = pref_plan(c);
    if (pl.xstat && g_xstat_ablate != 0) return QSAE_OK;
    if (nflag < 0 || nflag > c.B) return fail(QSAE_ERR_INVALID_ARG, "%s: flagged-row count out of range", __func__);
    int rc = flagged_range(pref_flagged_args(c, pl), first, nflag);
    if (rc != QSAE_OK) return rc;
    // rows the exact kernels ranked: their reconstruction through the stand-alone decode kernel, by row list
    if (c.dec && nflag > 0) {
        const int* flags = reinterpret_cast<const int*>(c.ws + pl.L.flags);
        rc = decode_binary_sparse_rows(flags + 1, nflag, c.idx, c.val, c.k, c.H, *c.dec, s);
        if (rc != QSAE_OK) return rc;
    }
    if (c.dense && !pl.filled)
        return pl.xstat ? densify_rows(c.idx, c.val, c.B, c.k, c.H, c.dense, c.dense_ld, s)
                        : scatter_rows(c.idx, c.val, c.B, c.k, c.H, c.dense, c.dense_ld, s);
    return QSAE_OK;
}

// Blocking form: submit, one 4-byte read-back (with the first `spec` flagged rows recomputed meanwhile), finish.
static int run_prefilter(const PrefCall& c, int spec, int* flagged_rows) {
    int rc = prefilter_submit(c);
    if (rc != QSAE_OK) return rc;
    const PrefPlan pl = pref_plan(c);
    if (pl.xstat && g_xstat_ablate != 0) return QSAE_OK;
    const FlaggedArgs fa = pref_flagged_args(c, pl);
    int nflag = 0;
    hipStream_t s = as_stream(c.stream);
    {
        const int* flags = reinterpret_cast<const int*>(c.ws + pl.L.flags);
        ThreadDeviceCtx* ctx = nullptr;
        rc = thread_device_ctx(&ctx);
        if (rc != QSAE_OK) return rc;
        *ctx->pinned = 0;
        QSAE_HIP(hipMemcpyAsync(ctx->pinned, flags, sizeof(int), hipMemcpyDeviceToHost, s));
        QSAE_HIP(hipEventRecord(ctx->ev_copied, s));
        spec = spec < 0 ? 0 : (spec > kMaxSpecRows ? kMaxSpecRows : spec);
        spec = spec < c.B ? spec : c.B;
        rc = flagged_spec(fa, spec);
        if (rc != QSAE_OK) return rc;
        QSAE_HIP(hipEventSynchronize(ctx->ev_copied));
        nflag = *ctx->pinned;
    }
    if (flagged_rows) *flagged_rows = nflag;
    return prefilter_finish(c, spec, nflag);
}

// ---- threshold bits from the candidate sweep: z = (sigmoid(x W^T + b) > 0.5), exact ---------------------------
// The matryoshka forward (reference sae/quantized_matryoshka.py:97-99,217-220) needs only the BIT
// z = sigmoid(latent) > 0.5  <=>  latent >= c (QSAE_SIG_GT_BITS) of every latent, never its value.  The fp16
// sweep with tau = c lists every hidden unit whose approximate latent s^ is >= c - 2 eps_b.  With |s^ - s| <= eps_b:
//   s^ - c >  eps_b   =>  s > c          bit 1, no further work
//   s^ - c < -eps_b   =>  s < c          bit 0 (listed only because the sweep's cut is 2 eps_b wide)
//   otherwise                            exact fp32 chain (the refine gather), bit = chain >= c
// One wave per activation row builds the row's bit vector in LDS and writes it out once.  Rows whose list
// overflowed (dense activations, NaN inputs) are flagged and recomputed by the exact dense kernel.
constexpr int kBitsWaves = 4;
constexpr int kBitsMaxUnc = 256;      // latents per row inside the uncertainty band (more -> flagged)
constexpr int kBitsChunk = 8192;      // flagged rows per exact fallback launch
constexpr int kBitsCap = 4096;        // list entries per row: ~10 % of 32768 units active plus the uncertainty band (denser
                                      // rows also overflow the sweep's 6 records per lane and 32 latents, and are flagged)
constexpr int kBitsSets = 3;          // W blocks in flight per wave (of D/32): 8 x 16-byte loads per lane each
__host__ __device__ static inline size_t bits_lds_per_wave(int H) {
    return static_cast<size_t>((H + 31) / 32) * 4 + 64 * kRefTileStride * 4 + kBitsMaxUnc * 4;
}

__global__ void __launch_bounds__(64 * kBitsWaves)
resolve_bits_kernel(const uint2* __restrict__ cand, const int* __restrict__ cnt, int cap, int parts,
                    const int* __restrict__ cnt_parts, const float* __restrict__ margin, const float* __restrict__ x,
                    const float* __restrict__ W, const float* __restrict__ bias, int B, int D, int H,
                    uint32_t* __restrict__ zbits, int64_t words_ld, int* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bits_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x * kBitsWaves + wave;
    if (b >= B) return;
    const int words = (H + 31) / 32;
    unsigned char* mybase = bits_smem + static_cast<size_t>(wave) * bits_lds_per_wave(H);
    uint32_t* zrow = reinterpret_cast<uint32_t*>(mybase);
    float* wt = reinterpret_cast<float*>(mybase + static_cast<size_t>(words) * 4);
    int* hidx = reinterpret_cast<int*>(wt + 64 * kRefTileStride);
    auto flag_row = [&]() {
        if (lane == 0) {
            const int slot = atomicAdd(&flags[0], 1);
            flags[1 + slot] = b;
        }
    };
    auto lds_handoff = [&]() { asm volatile("" ::: "memory"); };       // one wave's LDS operations execute in order
    typedef const __attribute__((address_space(4))) int* cint_t;
    typedef const __attribute__((address_space(4))) float* cflt_t;
    const int cap_part = cap / parts;
    bool seg_overflow = false;
    for (int p = 0; p < parts; ++p) {
        const int np = p == 0 ? ((cint_t)cnt)[b] : ((cint_t)cnt_parts)[static_cast<size_t>(p - 1) * B + b];
        seg_overflow |= np > cap_part;
    }
    if (seg_overflow) { flag_row(); return; }
    for (int w = lane; w < words; w += 64) zrow[w] = 0u;
    lds_handoff();
    const float c = __uint_as_float(QSAE_SIG_GT_BITS);
    const float half = 0.5f * ((cflt_t)margin)[b] * 1.00001f;          // eps_b with slack for the roundings below
    const uint2* list = cand + static_cast<int64_t>(b) * cap;
    int m = 0;                                                         // uncertain latents so far (wave-uniform)
    bool bad = false;
    for (int p = 0; p < parts; ++p) {
        const int np = p == 0 ? ((cint_t)cnt)[b] : ((cint_t)cnt_parts)[static_cast<size_t>(p - 1) * B + b];
        const uint2* seg = list + p * cap_part;
        for (int i0 = 0; i0 < np; i0 += 256) {
            uint2 r[4];                                                // four list slots per lane in flight
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 64 * u + lane;
                r[u] = i < np ? seg[i] : uint2{0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + 64 * u >= np) break;                          // wave-uniform
                const int i = i0 + 64 * u + lane;
                bool unc = false;
                const int h = static_cast<int>(r[u].y);
                if (i < np) {
                    const float v = __uint_as_float(r[u].x);
                    const float d = v - c;
                    bad |= (v != v) || h < 0 || h >= H;
                    if (d > half) atomicOr(&zrow[h >> 5], 1u << (h & 31));
                    else unc = d >= -half;
                }
                const unsigned long long msk = __ballot(unc);
                if (unc) {
                    const int pos = m + __popcll(msk & ((1ull << lane) - 1ull));
                    if (pos < kBitsMaxUnc) hidx[pos] = h;
                }
                m += __popcll(msk);
            }
        }
    }
    if (__any(bad) || m > kBitsMaxUnc) { flag_row(); return; }         // NaN latents / too many: the exact kernel decides
    lds_handoff();
    // exact fp32 chain of the uncertain latents (ascending k, seeded with the bias): the transposed block gather
    // of refine_topk_kernel.  ~30 latents x 2 KiB of W per row: this gather (3.9 GB per 65536 rows at the headline
    // shape) is what the kernel's time is.
    typedef const __attribute__((address_space(4))) f32x4* cvec_t;
    cvec_t xrow = (cvec_t)(x + static_cast<int64_t>(b) * D);
    const int nblk = D / 32;
    for (int j0 = 0; j0 < m; j0 += 64) {
        const int j = j0 + lane;
        const int h = (j < m) ? hidx[j] : hidx[j0];
        float acc = bias ? bias[h] : 0.0f;
        const float* rp[8];
        bool live[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int jj = j0 + 8 * i + (lane >> 3);
            live[i] = jj < m;                                          // rows past the list are not fetched at all
            jj = live[i] ? jj : j0;
            rp[i] = W + static_cast<int64_t>(hidx[jj]) * D + 4 * (lane & 7);
        }
        f32x4 st[kBitsSets][8];
#pragma unroll
        for (int q = 0; q < kBitsSets; ++q)
            if (q < nblk) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (live[i]) st[q][i] = *reinterpret_cast<const f32x4*>(rp[i] + 32 * q);
            }
        auto consume = [&](const f32x4 (&sv)[8], int t) {
            f32x4 xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xv[q] = xrow[8 * t + q];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<f32x4*>(wt + (8 * i + (lane >> 3)) * kRefTileStride + 4 * (lane & 7)) = sv[i];
            lds_handoff();
            const float* mine = wt + lane * kRefTileStride;
            f32x4 w[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = *reinterpret_cast<const f32x4*>(mine + 4 * q);
            lds_handoff();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc = fmaf(xv[q][0], w[q][0], acc);
                acc = fmaf(xv[q][1], w[q][1], acc);
                acc = fmaf(xv[q][2], w[q][2], acc);
                acc = fmaf(xv[q][3], w[q][3], acc);
            }
        };
        for (int t = 0; t < nblk; t += kBitsSets) {
#pragma unroll
            for (int q = 0; q < kBitsSets; ++q) {
                if (t + q < nblk) {
                    consume(st[q], t + q);
                    if (t + q + kBitsSets < nblk) {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (live[i]) st[q][i] = *reinterpret_cast<const f32x4*>(rp[i] + 32 * (t + q + kBitsSets));
                    }
                }
            }
        }
        if (j < m && sig_gt_half(acc)) atomicOr(&zrow[h >> 5], 1u << (h & 31));
    }
    lds_handoff();
    uint32_t* out = zbits + static_cast<int64_t>(b) * words_ld;
    for (int w = lane; w < words; w += 64) out[w] = zrow[w];
}

__global__ void __launch_bounds__(256)
scatter_bit_rows_kernel(const uint32_t* __restrict__ src, const int* __restrict__ rows, int n, int words,
                        uint32_t* __restrict__ dst, int64_t words_ld) {
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= static_cast<long long>(n) * words) return;
    const int r = static_cast<int>(gid / words), w = static_cast<int>(gid % words);
    dst[static_cast<long long>(rows[r]) * words_ld + w] = src[gid];
}

struct BitsLayout {
    size_t tau, cnt, cnt_parts, cand, flags, xq, inv, margin, fx, fbits, total;
};
static BitsLayout bits_layout(int B, int D, int H, int cap = kBitsCap) {
    BitsLayout L;
    size_t off = 0;
    L.tau = off;       off = align_up(off + static_cast<size_t>(B) * 4, 256);
    L.cnt = off;       off = align_up(off + static_cast<size_t>(B) * 4, 256);
    L.cnt_parts = off; off = align_up(off + static_cast<size_t>(B) * 4 * 7, 256);
    L.cand = off;      off = align_up(off + static_cast<size_t>(B) * cap * 8, 256);
    L.flags = off;     off = align_up(off + (static_cast<size_t>(B) + 4) * 4, 256);
    L.xq = off;        off = align_up(off + static_cast<size_t>(B) * D * 2, 256);
    L.inv = off;       off = align_up(off + static_cast<size_t>(B) * 4, 256);
    L.margin = off;    off = align_up(off + static_cast<size_t>(B) * 4, 256);
    L.fx = off;        off = align_up(off + static_cast<size_t>(kBitsChunk) * D * 4, 256);
    L.fbits = off;     off = align_up(off + static_cast<size_t>(kBitsChunk) * ((H + 31) / 32) * 4, 256);
    L.total = off;
    return L;
}

static bool bits_prefilter_shape_ok(int B, int D, int H) {
    return B > 0 && xstat_supported(D, H, 0) && D % 64 == 0 && D <= kRefMaxD && H <= (1 << 20) &&
           bits_lds_per_wave(H) * kBitsWaves <= 160 * 1024;
}

// Everything up to and including the bit resolution; afterwards flags[0] (device) = rows that need the exact dense
// kernel, flags[1..] their ids, every other row's bits are final.
static int bits_submit(const float* x, const float* W, const float* bias, const _Float16* Wq, const float* meta,
                       int B, int D, int H, uint32_t* zbits, int64_t words_ld, char* ws, qsae_stream_t stream) {
    hipStream_t s = as_stream(stream);
    const SweepProfile prof = take_sweep_profile();
    const BitsLayout L = bits_layout(B, D, H);
    float* tau = reinterpret_cast<float*>(ws + L.tau);
    int* cnt = reinterpret_cast<int*>(ws + L.cnt);
    int* cnt_parts = reinterpret_cast<int*>(ws + L.cnt_parts);
    uint2* cand = reinterpret_cast<uint2*>(ws + L.cand);
    int* flags = reinterpret_cast<int*>(ws + L.flags);
    _Float16* xq = reinterpret_cast<_Float16*>(ws + L.xq);
    float* inv = reinterpret_cast<float*>(ws + L.inv);
    float* margin = reinterpret_cast<float*>(ws + L.margin);
    const int words = (H + 31) / 32;
    QSAE_HIP(hipMemsetAsync(flags, 0, sizeof(int), s));
    QSAE_HIP(hipMemsetAsync(cnt, 0, static_cast<size_t>(B) * 4, s));
    QSAE_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(tau), static_cast<int>(QSAE_SIG_GT_BITS), B, s));
    if (words_ld > words)
        QSAE_HIP(hipMemset2DAsync(zbits + words, words_ld * 4, 0, (words_ld - words) * 4, B, s));
    launch_x_prep(x, B, D, meta, xq, inv, margin, s);
    QSAE_LAUNCH_CHECK();
    const int parts = xstat_parts(B, H, kBitsCap);
    if (prof.begin) QSAE_HIP(hipEventRecord(prof.begin, s));
    XsArgs xa{xq, Wq, bias, tau, margin, inv, cand, cnt, B, H, kBitsCap, 0, g_xstat_rot, nullptr, nullptr, 0, H, 0, 0, 0,
              nullptr, nullptr, meta, nullptr, nullptr, parts, cnt_parts};
    int rc = launch_xstat(D, xa, s, D == 512 ? 9 : 0);      // (nothing to zero-fill here: the build without fill code)
    if (prof.end) QSAE_HIP(hipEventRecord(prof.end, s));
    if (rc != QSAE_OK) return rc;
    const size_t lds = bits_lds_per_wave(H) * kBitsWaves;
    QSAE_SET_MAX_LDS_ONCE(resolve_bits_kernel, 160 * 1024);
    hipLaunchKernelGGL(resolve_bits_kernel, dim3((B + kBitsWaves - 1) / kBitsWaves), dim3(64 * kBitsWaves), lds, s,
                       cand, cnt, kBitsCap, parts, cnt_parts, margin, x, W, bias, B, D, H, zbits, words_ld, flags);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

// The exact dense kernel on the nflag flagged rows (count known to the host).
static int bits_finish(const float* x, const float* W, const float* bias, int B, int D, int H, uint32_t* zbits,
                       int64_t words_ld, char* ws, qsae_stream_t stream, int nflag, int cap = kBitsCap) {
    if (nflag < 0 || nflag > B) return fail(QSAE_ERR_INVALID_ARG, "%s: flagged-row count out of range", __func__);
    hipStream_t s = as_stream(stream);
    const BitsLayout L = bits_layout(B, D, H, cap);
    const int* flags = reinterpret_cast<const int*>(ws + L.flags);
    const int words = (H + 31) / 32;
    float* fx = reinterpret_cast<float*>(ws + L.fx);
    uint32_t* fbits = reinterpret_cast<uint32_t*>(ws + L.fbits);
    for (int f0 = 0; f0 < nflag; f0 += kBitsChunk) {
        const int n = (nflag - f0) < kBitsChunk ? (nflag - f0) : kBitsChunk;
        const int* rows = flags + 1 + f0;
        const long long tot = static_cast<long long>(n) * D;
        hipLaunchKernelGGL(gather_rows_kernel, dim3(static_cast<unsigned>((tot + 255) / 256)), dim3(256), 0, s, x, rows, n,
                           D, fx);
        QSAE_LAUNCH_CHECK();
        const int rc = qsae_encode_bits(fx, W, bias, n, D, H, fbits, words, stream);
        if (rc != QSAE_OK) return rc;
        const long long tw = static_cast<long long>(n) * words;
        hipLaunchKernelGGL(scatter_bit_rows_kernel, dim3(static_cast<unsigned>((tw + 255) / 256)), dim3(256), 0, s, fbits,
                           rows, n, words, zbits, words_ld);
        QSAE_LAUNCH_CHECK();
    }
    return QSAE_OK;
}

// ---- fp32-accurate dense encoder on the fp16 matrix pipe (opt-in: HipEncoder.precision = "emulated") -----------------------
// out = act(bias + x W^T) with BOTH operands split into two fp16 terms under power-of-two scales (x: per row, W: global):
//   x s_x = x1 + x2 (+ <= 2^-22 |x s_x|),   W s_w = w1 + w2 (+ <= 2^-22 |W s_w|)
//   x . w  ~  (x1.w1 + x1.w2 + x2.w1) / (s_x s_w)          dropped: x2.w2 and the two remainders, <= 3 2^-22 sum |x_k w_k|
// Every fp16 x fp16 product is exact in fp32 and the three partial contractions run as ONE fp16 GEMM over a concatenated K:
// [x1 | x1 | x2] . [w1 | w2 | w1]^T (K' = 3 D), fp32 accumulation.  The result differs from the exact fmaf chain of
// qsae_encode_dense by fp32 accumulation-order noise (~1e-6 of a latent's standard deviation: the size of the reference's own
// sgemm-vs-chain difference), so it is NOT bit-identical to the oracle: nothing that ranks or thresholds latents uses it, only
// the dense ReLU latent of TernarySparseAutoencoder on request.  3 x the fp16 MFMA work of one pass against 16 x the rate.
__global__ void __launch_bounds__(256)
emu_w_max_kernel(const float* __restrict__ W, long long n, unsigned* __restrict__ mx_bits) {
    float mx = 0.f;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * blockDim.x) {
        const float a = fabsf(W[i]);
        mx = (a > mx || a != a) ? a : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(mx, off, 64);
        mx = (o > mx || o != o) ? o : mx;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(mx_bits, __float_as_uint(mx));       // non-negative floats (and NaN) order like their bits
}

// Wc[h] = [w1 | w2 | w1] (3 D halves); meta2[0] = s_w (0 when the weights are not finite: every output becomes NaN)
__global__ void __launch_bounds__(256)
emu_pack_w_kernel(const float* __restrict__ W, int H, int D, float* __restrict__ meta2, _Float16* __restrict__ Wc) {
    const float sw = pow2_scale_for(meta2[1]);
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid == 0) meta2[0] = sw;
    if (gid >= static_cast<long long>(H) * D) return;
    const int h = static_cast<int>(gid / D), d = static_cast<int>(gid % D);
    const float v = W[gid] * sw;                                        // exact scaling
    const _Float16 w1 = static_cast<_Float16>(v);
    const _Float16 w2 = static_cast<_Float16>(v - static_cast<float>(w1));       // exact subtraction, one rounding
    _Float16* row = Wc + static_cast<long long>(h) * 3 * D;
    row[d] = w1;
    row[D + d] = w2;
    row[2 * D + d] = w1;
}

// one wave per activation row: Xc[b] = [x1 | x1 | x2], inv[b] = 1 / (s_x s_w) (NaN for a row or weights that are not finite)
__global__ void __launch_bounds__(256)
emu_x_prep_kernel(const float* __restrict__ x, int B, int D, const float* __restrict__ meta2, _Float16* __restrict__ Xc,
                  float* __restrict__ inv) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* xr = x + static_cast<long long>(row) * D;
    float mx = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float a = fabsf(xr[d]);
        mx = (a > mx || a != a) ? a : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(mx, off, 64);
        mx = (o > mx || o != o) ? o : mx;
    }
    const float sx = pow2_scale_for(mx), sw = meta2[0];
    _Float16* out = Xc + static_cast<long long>(row) * 3 * D;
    for (int d = lane; d < D; d += 64) {
        const float v = xr[d] * sx;
        const _Float16 x1 = static_cast<_Float16>(v);
        const _Float16 x2 = static_cast<_Float16>(v - static_cast<float>(x1));
        out[d] = x1;
        out[D + d] = x1;
        out[2 * D + d] = x2;
    }
    if (lane == 0) inv[row] = (sx > 0.f && sw > 0.f) ? (1.0f / sx) * (1.0f / sw) : __builtin_nanf("");
}

template <int ACT, int BM, int BN, int WMW, int WNW>
struct EpiEmuDense {
    static constexpr int WTM = BM / WMW, WTN = BN / WNW, MT = WTM / 32, NT = WTN / 32;
    static constexpr int kCheckpoints = 0;
    static constexpr int kLdsFloats = 0;
    static constexpr int kStoresPerFinish = 0;
    struct Args {
        const float* inv;      // [B]
        const float* bias;     // [H] or nullptr
        float* out;            // [B][ld]
        int64_t ld;
    };
    __device__ __forceinline__ void begin(const Args&, const TileCtx&) {}
    __device__ __forceinline__ void end(const Args&, const TileCtx&) {}
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
        float bcol[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = c.n0 + c.wn * WTN + nt * 32 + c.lane_col;
            bcol[nt] = (a.bias && col < c.N) ? a.bias[col] : 0.0f;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c.m0 + c.wm * WTM + mt * 32 + mfma_row(r, c.lane_half);
                if (row >= c.M) continue;
                const float iv = a.inv[row];
                float* orow = a.out + static_cast<int64_t>(row) * a.ld;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = c.n0 + c.wn * WTN + nt * 32 + c.lane_col;
                    float v = fmaf(acc[mt][nt][r], iv, bcol[nt]);
                    if (ACT == QSAE_ACT_RELU) v = v > 0.0f ? v : (v != v ? v : 0.0f);      // (NaN stays NaN, like torch.relu)
                    if (ACT == QSAE_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                    if (col < c.N) orow[col] = v;
                }
            }
    }
};

// ---- dense activations: classify EVERY latent with the fp16 pass, list only the uncertainty band -------------------------------
// The candidate lists above hold every unit whose approximate latent reaches the cutoff -- all active units.  With dense
// activations (an untrained encoder: half of the units fire) they overflow and every row falls back to the exact fp32
// contraction (17.9 ms per 65536 x 32768 at 78 % of the fp32 matrix peak).  But |s^ - s| <= eps_b decides most bits by itself:
//   s^ - c >  eps_b  =>  bit 1        s^ - c < -eps_b  =>  bit 0        otherwise (the band, ~0.7 % of the latents)  =>  exact chain
// The fp16 LDS-DMA GEMM (activation rows on accumulator registers, hidden units on lanes) writes the certain bits with one
// ballot per accumulator register -- 32 hidden units of one row = one word -- and appends the band to the row's list; the
// resolve kernel then patches the listed bits from the exact chain.  Same bound, same exactness argument as above.
constexpr int kBandCap = 1024;        // band entries per row (mean ~240 at the cutoff of a zero-mean latent; more -> flagged)

template <int BM, int BN, int WMW, int WNW>
struct EpiBitsBand {
    static constexpr int WTM = BM / WMW, WTN = BN / WNW, MT = WTM / 32, NT = WTN / 32;
    static constexpr int kThreads = 64 * WMW * WNW;
    static constexpr int kCheckpoints = 0;
    // Band entries of one tile are collected in LDS and appended to the rows' lists with ONE global atomic per row and
    // tile (a global atomic with return per entry -- 22 M of them at 65536 x 32768, 1 % in the band -- doubled the
    // kernel's time: 3.9 -> 7.5 ms).  Scratch: hits per tile row [BM] | list base per tile row [BM] | total | entries.
    static constexpr int kTileCap = 2048;                     // entries per tile held in LDS (mean ~650 at 1 %); more go direct
    static constexpr int kLdsFloats = 2 * BM + 4 + 3 * kTileCap;
    static constexpr int kStoresPerFinish = 0;
    struct Args {
        const float* inv;      // [B] 1 / (row scale * weight scale)
        const float* margin;   // [B] 2 eps_b
        const float* bias;     // [H] or nullptr
        uint32_t* zbits;       // [B][words_ld]
        int64_t words_ld;
        uint2* cand;           // [B][cap] band entries {approximate latent, hidden index}
        int* cnt;              // [B], zeroed by the caller
        int cap;
    };
    __device__ __forceinline__ void begin(const Args&, const TileCtx&) {}
    __device__ __forceinline__ void end(const Args&, const TileCtx&) {}
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
        const float cut = __uint_as_float(QSAE_SIG_GT_BITS);
        int* lcount = reinterpret_cast<int*>(c.lds_epi);
        int* lbase = lcount + BM;
        int* ltotal = lbase + BM;
        uint32_t* ent = reinterpret_cast<uint32_t*>(ltotal + 4);
        if (c.tid < BM) lcount[c.tid] = 0;
        if (c.tid == 0) *ltotal = 0;
        __syncthreads();
        float bcol[NT];
        bool col_ok[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = c.n0 + c.wn * WTN + nt * 32 + c.lane_col;
            col_ok[nt] = col < c.N;
            bcol[nt] = (a.bias && col_ok[nt]) ? a.bias[col] : 0.0f;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // lanes 0-31 carry activation row mfma_row(r, 0), lanes 32-63 row mfma_row(r, 1)
                const int lrow = c.wm * WTM + mt * 32 + mfma_row(r, c.lane_half);
                const int row = c.m0 + lrow;
                const bool row_ok = row < c.M;
                const int rr = row_ok ? row : c.M - 1;
                const float iv = a.inv[rr];
                const float half = 0.5f * a.margin[rr] * 1.00001f;       // eps_b with slack for the roundings below
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col0 = c.n0 + c.wn * WTN + nt * 32;
                    const float v = fmaf(acc[mt][nt][r], iv, bcol[nt]);
                    const float d = v - cut;
                    const bool live = row_ok && col_ok[nt];
                    const bool one = live && d > half;
                    const bool band = live && !(d > half) && !(d < -half);       // (NaN lands here: the resolve step flags the row)
                    const unsigned long long m1 = __ballot(one);
                    if (c.lane_col == 0 && row_ok && col0 < c.N)
                        a.zbits[static_cast<int64_t>(row) * a.words_ld + (col0 >> 5)] =
                            c.lane_half ? static_cast<uint32_t>(m1 >> 32) : static_cast<uint32_t>(m1);
                    if (band) {
                        const uint32_t col = static_cast<uint32_t>(col0 + c.lane_col);
                        const int slot = atomicAdd(ltotal, 1);
                        if (slot < kTileCap) {
                            const int p = atomicAdd(&lcount[lrow], 1);
                            ent[3 * slot] = __float_as_uint(v);
                            ent[3 * slot + 1] = col;
                            ent[3 * slot + 2] = (static_cast<uint32_t>(lrow) << 16) | static_cast<uint32_t>(p & 0xFFFF);
                        } else {                                                  // tile buffer full (rows without a finite margin)
                            const int pos = atomicAdd(&a.cnt[row], 1);
                            if (pos < a.cap) a.cand[static_cast<int64_t>(row) * a.cap + pos] = make_uint2(__float_as_uint(v), col);
                        }
                    }
                }
            }
        __syncthreads();
        if (c.tid < BM) {
            const int n = lcount[c.tid];
            lbase[c.tid] = (n > 0 && c.m0 + c.tid < c.M) ? atomicAdd(&a.cnt[c.m0 + c.tid], n) : 0;
        }
        __syncthreads();
        const int total = *ltotal < kTileCap ? *ltotal : kTileCap;
        for (int e = c.tid; e < total; e += kThreads) {
            const uint32_t rp = ent[3 * e + 2];
            const int lrow = static_cast<int>(rp >> 16);
            const int pos = lbase[lrow] + static_cast<int>(rp & 0xFFFFu);
            if (pos < a.cap)
                a.cand[static_cast<int64_t>(c.m0 + lrow) * a.cap + pos] = make_uint2(ent[3 * e], ent[3 * e + 1]);
        }
        __syncthreads();                                               // the scratch is reused by the next tile
    }
};

// One wave per activation row: every listed latent lies inside the uncertainty band; its bit is decided by the exact fp32
// chain (the transposed block gather of refine_topk_kernel) and, where it comes out 1, set in the row's word with a
// no-return atomic -- the certain bits are already there.  LDS per wave: the transposed W tile + the index list (13 KiB:
// three workgroups per CU; the row's bit vector is not staged).
constexpr size_t kBandLdsPerWave = 64 * kRefTileStride * 4 + static_cast<size_t>(kBandCap) * 4;

__global__ void __launch_bounds__(64 * kBitsWaves, 3)
resolve_band_kernel(const uint2* __restrict__ cand, const int* __restrict__ cnt, const float* __restrict__ x,
                    const float* __restrict__ W, const float* __restrict__ bias, int B, int D, int H,
                    uint32_t* __restrict__ zbits, int64_t words_ld, int* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char band_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x * kBitsWaves + wave;
    if (b >= B) return;
    float* wt = reinterpret_cast<float*>(band_smem + static_cast<size_t>(wave) * kBandLdsPerWave);
    int* hidx = reinterpret_cast<int*>(wt + 64 * kRefTileStride);
    auto lds_handoff = [&]() { asm volatile("" ::: "memory"); };       // one wave's LDS operations execute in order
    typedef const __attribute__((address_space(4))) int* cint_t;
    const int m = ((cint_t)cnt)[b];
    bool bad = m > kBandCap;
    const uint2* list = cand + static_cast<int64_t>(b) * kBandCap;
    if (!bad) {
        for (int i = lane; i < m; i += 64) {
            const uint2 r = list[i];
            const float v = __uint_as_float(r.x);
            bad |= (v != v) || r.y >= static_cast<uint32_t>(H);
            hidx[i] = static_cast<int>(r.y);
        }
    }
    if (__any(bad)) {                                                  // overflowing band / NaN latents: the exact kernel decides
        if (lane == 0) {
            const int slot = atomicAdd(&flags[0], 1);
            flags[1 + slot] = b;
        }
        return;
    }
    lds_handoff();
    typedef const __attribute__((address_space(4))) f32x4* cvec_t;
    cvec_t xrow = (cvec_t)(x + static_cast<int64_t>(b) * D);
    uint32_t* zrow = zbits + static_cast<int64_t>(b) * words_ld;
    const int nblk = D / 32;
    for (int j0 = 0; j0 < m; j0 += 64) {
        const int j = j0 + lane;
        const int h = (j < m) ? hidx[j] : hidx[j0];
        float acc = bias ? bias[h] : 0.0f;
        const float* rp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int jj = j0 + 8 * i + (lane >> 3);
            jj = jj < m ? jj : j0;                                     // (past the list: the first row again, an L1 hit)
            rp[i] = W + static_cast<int64_t>(hidx[jj]) * D + 4 * (lane & 7);
        }
        f32x4 st[kBitsSets][8];
#pragma unroll
        for (int q = 0; q < kBitsSets; ++q)
            if (q < nblk) {
#pragma unroll
                for (int i = 0; i < 8; ++i) st[q][i] = *reinterpret_cast<const f32x4*>(rp[i] + 32 * q);
            }
        auto consume = [&](const f32x4 (&sv)[8], int t) {
            f32x4 xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xv[q] = xrow[8 * t + q];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<f32x4*>(wt + (8 * i + (lane >> 3)) * kRefTileStride + 4 * (lane & 7)) = sv[i];
            lds_handoff();
            const float* mine = wt + lane * kRefTileStride;
            f32x4 w[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = *reinterpret_cast<const f32x4*>(mine + 4 * q);
            lds_handoff();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc = fmaf(xv[q][0], w[q][0], acc);
                acc = fmaf(xv[q][1], w[q][1], acc);
                acc = fmaf(xv[q][2], w[q][2], acc);
                acc = fmaf(xv[q][3], w[q][3], acc);
            }
        };
        for (int t = 0; t < nblk; t += kBitsSets) {
#pragma unroll
            for (int q = 0; q < kBitsSets; ++q) {
                if (t + q < nblk) {
                    consume(st[q], t + q);
                    if (t + q + kBitsSets < nblk) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) st[q][i] = *reinterpret_cast<const f32x4*>(rp[i] + 32 * (t + q + kBitsSets));
                    }
                }
            }
        }
        if (j < m && sig_gt_half(acc)) atomicOr(&zrow[h >> 5], 1u << (h & 31));
    }
}

static bool bits_band_shape_ok(int B, int D, int H) {
    return B > 0 && D % 64 == 0 && D <= kRefMaxD && H % 32 == 0 && H <= (1 << 20);
}

// Everything up to and including the band resolution (flags[0] = rows for the exact dense kernel afterwards).
static int bits_band_submit(const float* x, const float* W, const float* bias, const _Float16* Wq, const float* meta,
                            int B, int D, int H, uint32_t* zbits, int64_t words_ld, char* ws, qsae_stream_t stream) {
    hipStream_t s = as_stream(stream);
    const SweepProfile prof = take_sweep_profile();
    const BitsLayout L = bits_layout(B, D, H, kBandCap);
    int* cnt = reinterpret_cast<int*>(ws + L.cnt);
    uint2* cand = reinterpret_cast<uint2*>(ws + L.cand);
    int* flags = reinterpret_cast<int*>(ws + L.flags);
    _Float16* xq = reinterpret_cast<_Float16*>(ws + L.xq);
    float* inv = reinterpret_cast<float*>(ws + L.inv);
    float* margin = reinterpret_cast<float*>(ws + L.margin);
    const int words = (H + 31) / 32;
    QSAE_HIP(hipMemsetAsync(flags, 0, sizeof(int), s));
    QSAE_HIP(hipMemsetAsync(cnt, 0, static_cast<size_t>(B) * 4, s));
    if (words_ld > words)
        QSAE_HIP(hipMemset2DAsync(zbits + words, words_ld * 4, 0, (words_ld - words) * 4, B, s));
    launch_x_prep(x, B, D, meta, xq, inv, margin, s);
    QSAE_LAUNCH_CHECK();
    using Epi = EpiBitsBand<256, 256, 4, 2>;
    typename Epi::Args ea{inv, margin, bias, zbits, words_ld, cand, cnt, kBandCap};
    if (prof.begin) QSAE_HIP(hipEventRecord(prof.begin, s));
    int rc = launch_gemm_dma<Epi, 256, 256, true, 2>(reinterpret_cast<const float*>(xq), B, reinterpret_cast<const float*>(Wq), H,
                                                     D / 2, ea, s, /*sweep=*/8);
    if (prof.end) QSAE_HIP(hipEventRecord(prof.end, s));
    if (rc != QSAE_OK) return rc;
    const size_t lds = kBandLdsPerWave * kBitsWaves;
    QSAE_SET_MAX_LDS_ONCE(resolve_band_kernel, 160 * 1024);
    hipLaunchKernelGGL(resolve_band_kernel, dim3((B + kBitsWaves - 1) / kBitsWaves), dim3(64 * kBitsWaves), lds, s,
                       cand, cnt, x, W, bias, B, D, H, zbits, words_ld, flags);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

// Blocking form: submit, the count through this thread's pinned word (one host round trip), finish.
static int run_bits_prefilter(const float* x, const float* W, const float* bias, const _Float16* Wq, const float* meta,
                              int B, int D, int H, uint32_t* zbits, int64_t words_ld, char* ws, qsae_stream_t stream,
                              int* flagged_rows) {
    hipStream_t s = as_stream(stream);
    int rc = bits_submit(x, W, bias, Wq, meta, B, D, H, zbits, words_ld, ws, stream);
    if (rc != QSAE_OK) return rc;
    const BitsLayout L = bits_layout(B, D, H);
    ThreadDeviceCtx* ctx = nullptr;
    rc = thread_device_ctx(&ctx);
    if (rc != QSAE_OK) return rc;
    *ctx->pinned = 0;
    QSAE_HIP(hipMemcpyAsync(ctx->pinned, ws + L.flags, sizeof(int), hipMemcpyDeviceToHost, s));
    QSAE_HIP(hipEventRecord(ctx->ev_copied, s));
    QSAE_HIP(hipEventSynchronize(ctx->ev_copied));
    const int nflag = *ctx->pinned;
    if (flagged_rows) *flagged_rows = nflag;
    if (nflag < 0 || nflag > B) return fail(QSAE_ERR_HIP, "%s: corrupt flagged-row count", __func__);
    return bits_finish(x, W, bias, B, D, H, zbits, words_ld, ws, stream, nflag);
}

}  // namespace qsae

using namespace qsae;

#ifdef QSAE_DEBUG_BUILD
// ---- debug library only (libqsae_hip_debug.so): process-wide tuning / ablation switches ------------------------
// fraction of the encoder FLOPs the sweep launch covers (the pilot block takes the rest)
extern "C" int qsae_debug_set_xstat_stamps(void* buf) {
    g_xstat_stamps = static_cast<unsigned long long*>(buf);
    return QSAE_OK;
}

extern "C" int qsae_debug_set_refine_stamps(void* buf) {
    g_ref_stamps = static_cast<unsigned long long*>(buf);
    return QSAE_OK;
}

extern "C" int qsae_debug_set_refine_sliced(int v) {
    g_ref_sliced = v;
    return QSAE_OK;
}

extern "C" int qsae_debug_set_refine_ablate(int v) {
    g_ref_ablate = v;
    return QSAE_OK;
}

extern "C" int qsae_debug_set_pilot(int div, int rank) {
    g_pilot_div = div;
    kPilotRank = rank;
    return QSAE_OK;
}

// in-kernel pilot of the stationary sweep: enable (0 = separate pilot GEMM + selection), rank among 32 group maxima
extern "C" int qsae_debug_set_inkernel_pilot(int enable, int rank) {
    g_fuse_xprep = enable >= 2 ? 1 : 0;                      // 2 = in-kernel pilot + activation preparation fused into the sweep prologue
    enable = enable ? 1 : 0;
    g_inkernel_pilot = enable;
    g_inkernel_rank = rank;                                  // 0 = derive from k
    return QSAE_OK;
}

extern "C" int qsae_debug_set_fill_co(int v) {
    g_fill_co = v;
    return QSAE_OK;
}

extern "C" int qsae_debug_set_xstat_rot(int rot) {
    g_pilot_tile = rot >= 1000 ? 1 : 0;                      // rot >= 1000: 256 x 128 pilot tile (timing comparison)
    rot %= 1000;
    g_fill_in_sweep = rot >= 100 ? 0 : 1;                    // rot >= 100: separate fill pass (timing comparison)
    rot %= 100;
    g_xstat_rot = rot;
    return QSAE_OK;
}

extern "C" int qsae_debug_set_prefilter_tile(int which) {
    g_xstat_ablate = which >= 10 ? which - 10 : 0;
    g_pref_tile = which >= 10 ? 2 : which;
    return QSAE_OK;
}

extern "C" int qsae_debug_set_phases(int phase_mask, int parts) {
    g_x_phase = phase_mask;
    g_x_parts = parts;
    return QSAE_OK;
}

extern "C" int qsae_debug_set_sweep_kernel(int which) {
    g_sweep_kernel = which;
    return QSAE_OK;
}

extern "C" int qsae_debug_set_topk_path(int path) {
    g_force_path = path;
    return QSAE_OK;
}

// test hook: byte offsets of the approximate pilot block [B][P] fp32 and of margin[B] (= 2 eps_b) in the workspace
extern "C" int qsae_debug_prefilter_offsets(int B, int D, int H, int k, size_t* pilot_off, size_t* margin_off,
                                            int* pilot_cols) {
    const FusedLayout L = fused_layout(B, D, H, k);
    const PrefLayout PL = pref_layout(B, D, L.total);
    if (pilot_off) *pilot_off = L.pilot;
    if (margin_off) *margin_off = PL.margin;
    if (pilot_cols) *pilot_cols = pilot_width(H);
    return QSAE_OK;
}

// test hook: where a prefilter call leaves the candidate lists in its workspace -- list entries [B][cap] {value bits, hidden
// index}, segment lengths cnt[B] (part 0) and cnt_parts[(p - 1) B + b] (parts 1..), thresholds tau[B], margins [B]
extern "C" int qsae_debug_prefilter_list_offsets(int B, int D, int H, int k, size_t* cand_off, size_t* cnt_off,
                                                 size_t* cnt_parts_off, size_t* tau_off, size_t* margin_off, int* cap,
                                                 int* parts) {
    const FusedLayout L = fused_layout(B, D, H, k);
    const PrefLayout PL = pref_layout(B, D, L.total);
    if (cand_off) *cand_off = L.cand;
    if (cnt_off) *cnt_off = L.cnt;
    if (cnt_parts_off) *cnt_parts_off = PL.cnt_parts;
    if (tau_off) *tau_off = L.tau;
    if (margin_off) *margin_off = PL.margin;
    if (cap) *cap = kCandCap;
    if (parts) *parts = xstat_parts(B, H, kCandCap);
    return QSAE_OK;
}
#endif  // QSAE_DEBUG_BUILD

// Fraction of the encoder's 2 B D H FLOPs that the profiled sweep launch (qsae_profile_sweep_events) covers: with the
// in-kernel pilot the launch computes every hidden unit (the pilot sample twice; only the algorithmic work is
// counted), otherwise the pilot block is a separate launch.
extern "C" double qsae_profile_sweep_flop_fraction(int H) {
    if (H <= 0) return 0.0;
    if (g_inkernel_pilot && g_pref_tile == 2 && H % kXsHT == 0 && pilot_width(H) % kXsHT == 0) return 1.0;
    return static_cast<double>(H - pilot_width(H)) / static_cast<double>(H);
}

extern "C" size_t qsae_encode_topk_workspace_bytes(int B, int D, int H, int k) {
    if (B <= 0 || H <= 0 || D <= 0 || k <= 0) return 0;
    if (use_fused(B, D, H, k)) return fused_layout(B, D, H, k).total;
    const size_t rows = static_cast<size_t>(B < kChunkRows ? B : kChunkRows);
    return rows * static_cast<size_t>(H) * sizeof(float);
}

static int encode_topk_impl(const float* x, const float* W, const float* bias, int B, int D, int H, int k,
                            int32_t* idx, float* val, void* workspace, size_t workspace_bytes,
                            qsae_stream_t stream, bool kperm, float* dense = nullptr, int64_t dense_ld = 0) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(x && W && idx && val && workspace, "null pointer");
    QSAE_CHECK_ARG(k >= 1 && k <= H, "1 <= k <= H required");
    if (workspace_bytes < qsae_encode_topk_workspace_bytes(B, D, H, k))
        return fail(QSAE_ERR_WORKSPACE, "%s: workspace too small", __func__);
    QSAE_CHECK_ARG(aligned16(workspace), "workspace must be 16-byte aligned");
    if (use_fused(B, D, H, k)) {
        QSAE_CHECK_SUPPORTED(D % 4 == 0, "D must be a multiple of 4");
        QSAE_CHECK_ARG(aligned16(x) && aligned16(W), "x and W must be 16-byte aligned");
        if (dense) {
            QSAE_CHECK_ARG(dense_ld >= H && dense_ld % 4 == 0 && aligned16(dense), "dense latent must be 16-byte aligned with ld >= H, ld % 4 == 0");
        }
        return run_fused(x, W, bias, B, D, H, k, idx, val, static_cast<char*>(workspace), stream, kperm, dense, dense_ld);
    }
    const int rc = run_chunked(x, W, bias, B, D, H, k, idx, val, static_cast<float*>(workspace), stream, kperm);
    if (rc != QSAE_OK || !dense) return rc;
    return qsae_densify(idx, val, B, k, H, dense, dense_ld, stream);
}

extern "C" int qsae_encode_topk_latent(const float* x, const float* W, const float* bias, int B, int D, int H, int k,
                                       int32_t* idx, float* val, float* dense, int64_t dense_ld, int kperm,
                                       void* workspace, size_t workspace_bytes, qsae_stream_t stream) {
    QSAE_CHECK_ARG(dense != nullptr && dense_ld >= H, "dense latent pointer / leading dimension");
    if (kperm && D % 32 != 0) return fail(QSAE_ERR_UNSUPPORTED, "%s: K-interleaved operands need D %% 32 == 0", __func__);
    return encode_topk_impl(x, W, bias, B, D, H, k, idx, val, workspace, workspace_bytes, stream, kperm != 0, dense,
                            dense_ld);
}

extern "C" int qsae_encode_topk(const float* x, const float* W, const float* bias, int B, int D, int H, int k,
                                int32_t* idx, float* val, void* workspace, size_t workspace_bytes,
                                qsae_stream_t stream) {
    return encode_topk_impl(x, W, bias, B, D, H, k, idx, val, workspace, workspace_bytes, stream, false);
}

extern "C" int qsae_encode_topk_kperm(const float* xp, const float* Wp, const float* bias, int B, int D, int H, int k,
                                      int32_t* idx, float* val, void* workspace, size_t workspace_bytes,
                                      qsae_stream_t stream) {
    if (D % 32 != 0) return fail(QSAE_ERR_UNSUPPORTED, "%s: K-interleaved operands need D %% 32 == 0", __func__);
    return encode_topk_impl(xp, Wp, bias, B, D, H, k, idx, val, workspace, workspace_bytes, stream, true);
}

// ---- fp16 prefilter entry points ---------------------------------------------------------------------
extern "C" size_t qsae_prefilter_w_bytes(int H, int D) {
    return (H > 0 && D > 0) ? static_cast<size_t>(H) * D * 2 : 0;
}

extern "C" int qsae_prefilter_pack_w(const float* W, const float* bias, int H, int D, void* Wq, float* meta,
                                     qsae_stream_t stream) {
    QSAE_CHECK_ARG(H > 0 && D > 0 && W && Wq && meta, "H > 0, D > 0, non-null pointers");
    hipStream_t s = as_stream(stream);
    QSAE_HIP(hipMemsetAsync(meta, 0, 4 * sizeof(float), s));
    hipLaunchKernelGGL(pref_w_stats_kernel, dim3((H + 3) / 4), dim3(256), 0, s, W, bias, H, D,
                       reinterpret_cast<unsigned*>(meta));
    QSAE_LAUNCH_CHECK();
    const long long n = static_cast<long long>(H) * D;
    hipLaunchKernelGGL(pref_w_cast_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, W, n, meta,
                       static_cast<_Float16*>(Wq));
    QSAE_LAUNCH_CHECK();
    // meta[3] has served (max |W| -> sw); from here on it holds the largest distance between a row and its fp16 copy
    QSAE_HIP(hipMemsetAsync(meta + 3, 0, sizeof(float), s));
    hipLaunchKernelGGL(pref_w_err_kernel, dim3((H + 3) / 4), dim3(256), 0, s, W, H, D, meta, reinterpret_cast<unsigned*>(meta + 3));
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

extern "C" size_t qsae_encode_topk_prefilter_workspace_bytes(int B, int D, int H, int k) {
    if (B <= 0 || H <= 0 || D <= 0 || k <= 0 || !prefilter_shape_ok(B, D, H, k)) return 0;
    return pref_layout(B, D, fused_layout(B, D, H, k).total).total_extra;
}

// Argument checks shared by the prefilter entry points; on success `call` (and `dec` when a dictionary is given) are filled.
static int prefilter_call(const char* who, const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                          int B, int D, int H, int k, const uint8_t* packed, int n_bits, float step, const float* dec_bias,
                          int32_t* idx, float* val, float* dense, int64_t dense_ld, float* recon, void* workspace,
                          size_t workspace_bytes, qsae_stream_t stream, PrefCall& call, RowDecode& dec,
                          const float* table = nullptr) {
    if (!(x && W && Wq && meta && idx && val && workspace)) return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: null pointer", who);
    if (!(k >= 1 && k <= H)) return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: 1 <= k <= H required", who);
    if (!prefilter_shape_ok(B, D, H, k))
        return fail(QSAE_ERR_UNSUPPORTED, "%s: unsupported: shape outside the prefilter's range (use qsae_encode_topk_latent)", who);
    if (workspace_bytes < qsae_encode_topk_prefilter_workspace_bytes(B, D, H, k))
        return fail(QSAE_ERR_WORKSPACE, "%s: workspace too small", who);
    if (!(aligned16(workspace) && aligned16(x) && aligned16(W) && aligned16(Wq)))
        return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: 16-byte alignment", who);
    if (dense && !(dense_ld >= H && dense_ld % 4 == 0 && aligned16(dense)))
        return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: dense latent alignment / ld", who);
    call = PrefCall{x, W, bias, static_cast<const _Float16*>(Wq), meta, B, D, H, k, idx, val, static_cast<char*>(workspace),
                    stream, dense, dense_ld, nullptr};
    if (packed) {
        if (!recon) return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: packed given without recon", who);
        if (!(n_bits >= 1 && n_bits <= 8)) return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: 1 <= n_bits <= 8 required", who);
        if ((reinterpret_cast<uintptr_t>(packed) & 3u) != 0)
            return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: packed must be 4-byte aligned", who);
        dec = RowDecode{reinterpret_cast<const uint32_t*>(packed), qsae_binary_row_bytes(D, n_bits) / 4, n_bits,
                        field_width(n_bits), D, step, dec_bias, recon, nullptr};
        call.dec = &dec;
    } else if (table) {     // fp32 dictionary rows; `step` is the scale
        if (!recon) return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: table given without recon", who);
        if (!(aligned16(table) && aligned16(recon)))
            return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: table and recon must be 16-byte aligned", who);
        dec = RowDecode{nullptr, 0, 0, 0, D, step, dec_bias, recon, table};
        call.dec = &dec;
    }
    return QSAE_OK;
}

extern "C" int qsae_encode_topk_prefilter(const float* x, const float* W, const float* bias, const void* Wq,
                                          const float* meta, int B, int D, int H, int k, int32_t* idx, float* val,
                                          float* dense, int64_t dense_ld, void* workspace, size_t workspace_bytes,
                                          int spec_rows, int* flagged_rows, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    if (flagged_rows) *flagged_rows = 0;
    if (B == 0) return QSAE_OK;
    PrefCall call;
    RowDecode dec;
    const int rc = prefilter_call(__func__, x, W, bias, Wq, meta, B, D, H, k, nullptr, 0, 0.f, nullptr, idx, val, dense,
                                  dense_ld, nullptr, workspace, workspace_bytes, stream, call, dec);
    if (rc != QSAE_OK) return rc;
    return run_prefilter(call, spec_rows, flagged_rows);
}

extern "C" int qsae_binary_forward_prefilter(const float* x, const float* W, const float* bias, const void* Wq,
                                             const float* meta, int B, int D, int H, int k, const uint8_t* packed,
                                             int n_bits, float step, const float* dec_bias, int32_t* idx, float* val,
                                             float* dense, int64_t dense_ld, float* recon, void* workspace,
                                             size_t workspace_bytes, int spec_rows, int* flagged_rows,
                                             qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    if (flagged_rows) *flagged_rows = 0;
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(packed && recon, "null pointer");
    PrefCall call;
    RowDecode dec;
    const int rc = prefilter_call(__func__, x, W, bias, Wq, meta, B, D, H, k, packed, n_bits, step, dec_bias, idx, val, dense,
                                  dense_ld, recon, workspace, workspace_bytes, stream, call, dec);
    if (rc != QSAE_OK) return rc;
    return run_prefilter(call, spec_rows, flagged_rows);
}

extern "C" int qsae_prefilter_submit(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                                     int B, int D, int H, int k, const uint8_t* packed, int n_bits, float step,
                                     const float* dec_bias, int32_t* idx, float* val, float* dense, int64_t dense_ld,
                                     float* recon, void* workspace, size_t workspace_bytes, int* flagged_host,
                                     qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    QSAE_CHECK_ARG(flagged_host != nullptr, "flagged_host must point to a host int");
    if (B == 0) { *flagged_host = 0; return QSAE_OK; }
    PrefCall call;
    RowDecode dec;
    int rc = prefilter_call(__func__, x, W, bias, Wq, meta, B, D, H, k, packed, n_bits, step, dec_bias, idx, val, dense,
                            dense_ld, recon, workspace, workspace_bytes, stream, call, dec);
    if (rc != QSAE_OK) return rc;
    rc = prefilter_submit(call);
    if (rc != QSAE_OK) return rc;
    const PrefPlan pl = pref_plan(call);
    QSAE_HIP(hipMemcpyAsync(flagged_host, call.ws + pl.L.flags, sizeof(int), hipMemcpyDeviceToHost, as_stream(stream)));
    return QSAE_OK;
}

extern "C" int qsae_prefilter_finish(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                                     int B, int D, int H, int k, const uint8_t* packed, int n_bits, float step,
                                     const float* dec_bias, int32_t* idx, float* val, float* dense, int64_t dense_ld,
                                     float* recon, void* workspace, size_t workspace_bytes, int flagged,
                                     qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    if (B == 0) return QSAE_OK;
    PrefCall call;
    RowDecode dec;
    const int rc = prefilter_call(__func__, x, W, bias, Wq, meta, B, D, H, k, packed, n_bits, step, dec_bias, idx, val, dense,
                                  dense_ld, recon, workspace, workspace_bytes, stream, call, dec);
    if (rc != QSAE_OK) return rc;
    return prefilter_finish(call, /*first=*/0, flagged);
}

extern "C" int qsae_table_forward_prefilter(const float* x, const float* W, const float* bias, const void* Wq,
                                           const float* meta, int B, int D, int H, int k, const float* table, float scale,
                                           const float* dec_bias, int32_t* idx, float* val, float* dense, int64_t dense_ld,
                                           float* recon, void* workspace, size_t workspace_bytes, int spec_rows,
                                           int* flagged_rows, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    if (flagged_rows) *flagged_rows = 0;
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(table && recon, "null pointer");
    PrefCall call;
    RowDecode dec;
    const int rc = prefilter_call(__func__, x, W, bias, Wq, meta, B, D, H, k, nullptr, 0, scale, dec_bias, idx, val, dense,
                                  dense_ld, recon, workspace, workspace_bytes, stream, call, dec, table);
    if (rc != QSAE_OK) return rc;
    return run_prefilter(call, spec_rows, flagged_rows);
}

extern "C" int qsae_prefilter_submit_table(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                                           int B, int D, int H, int k, const float* table, float scale, const float* dec_bias,
                                           int32_t* idx, float* val, float* dense, int64_t dense_ld, float* recon,
                                           void* workspace, size_t workspace_bytes, int* flagged_host, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    QSAE_CHECK_ARG(flagged_host != nullptr, "flagged_host must point to a host int");
    if (B == 0) { *flagged_host = 0; return QSAE_OK; }
    QSAE_CHECK_ARG(table && recon, "null pointer");
    PrefCall call;
    RowDecode dec;
    int rc = prefilter_call(__func__, x, W, bias, Wq, meta, B, D, H, k, nullptr, 0, scale, dec_bias, idx, val, dense,
                            dense_ld, recon, workspace, workspace_bytes, stream, call, dec, table);
    if (rc != QSAE_OK) return rc;
    rc = prefilter_submit(call);
    if (rc != QSAE_OK) return rc;
    const PrefPlan pl = pref_plan(call);
    QSAE_HIP(hipMemcpyAsync(flagged_host, call.ws + pl.L.flags, sizeof(int), hipMemcpyDeviceToHost, as_stream(stream)));
    return QSAE_OK;
}

extern "C" int qsae_prefilter_finish_table(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                                           int B, int D, int H, int k, const float* table, float scale, const float* dec_bias,
                                           int32_t* idx, float* val, float* dense, int64_t dense_ld, float* recon,
                                           void* workspace, size_t workspace_bytes, int flagged, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(table && recon, "null pointer");
    PrefCall call;
    RowDecode dec;
    const int rc = prefilter_call(__func__, x, W, bias, Wq, meta, B, D, H, k, nullptr, 0, scale, dec_bias, idx, val, dense,
                                  dense_ld, recon, workspace, workspace_bytes, stream, call, dec, table);
    if (rc != QSAE_OK) return rc;
    return prefilter_finish(call, /*first=*/0, flagged);
}

extern "C" size_t qsae_encode_bits_prefilter_workspace_bytes(int B, int D, int H) {
    if (!bits_prefilter_shape_ok(B, D, H)) return 0;
    return bits_layout(B, D, H).total;
}

extern "C" int qsae_encode_bits_prefilter(const float* x, const float* W, const float* bias, const void* Wq,
                                          const float* meta, int B, int D, int H, uint32_t* zbits, int64_t words_ld,
                                          void* workspace, size_t workspace_bytes, int* flagged_rows,
                                          qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    if (flagged_rows) *flagged_rows = 0;
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(x && W && Wq && meta && zbits, "null pointer");
    QSAE_CHECK_ARG(words_ld >= (H + 31) / 32, "words_ld < ceil(H/32)");
    QSAE_CHECK_SUPPORTED(bits_prefilter_shape_ok(B, D, H), "shape not covered by the fp16 candidate sweep (D in {128,256,512}, H % 64 == 0)");
    QSAE_CHECK_ARG(aligned16(x) && aligned16(W) && aligned16(Wq), "x, W and Wq must be 16-byte aligned");
    QSAE_CHECK_ARG(workspace && workspace_bytes >= bits_layout(B, D, H).total, "workspace too small");
    QSAE_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "workspace must be 256-byte aligned");
    return run_bits_prefilter(x, W, bias, static_cast<const _Float16*>(Wq), meta, B, D, H, zbits, words_ld,
                              static_cast<char*>(workspace), stream, flagged_rows);
}

/* the two-call form (see qsae_prefilter_submit / _finish) */
static int bits_args_ok(const char* who, const float* x, const float* W, const void* Wq, const float* meta, int B, int D, int H,
                        const uint32_t* zbits, int64_t words_ld, const void* workspace, size_t workspace_bytes) {
    if (!(x && W && Wq && meta && zbits)) return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: null pointer", who);
    if (words_ld < (H + 31) / 32) return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: words_ld < ceil(H/32)", who);
    if (!bits_prefilter_shape_ok(B, D, H))
        return fail(QSAE_ERR_UNSUPPORTED, "%s: unsupported: shape not covered by the fp16 candidate sweep (D in {128,256,512}, H %% 64 == 0)", who);
    if (!(aligned16(x) && aligned16(W) && aligned16(Wq)))
        return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: x, W and Wq must be 16-byte aligned", who);
    if (!(workspace && workspace_bytes >= bits_layout(B, D, H).total))
        return fail(QSAE_ERR_WORKSPACE, "%s: workspace too small", who);
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0)
        return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: workspace must be 256-byte aligned", who);
    return QSAE_OK;
}

extern "C" int qsae_encode_bits_prefilter_submit(const float* x, const float* W, const float* bias, const void* Wq,
                                                 const float* meta, int B, int D, int H, uint32_t* zbits, int64_t words_ld,
                                                 void* workspace, size_t workspace_bytes, int* flagged_host,
                                                 qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    QSAE_CHECK_ARG(flagged_host != nullptr, "flagged_host must point to a host int");
    if (B == 0) { *flagged_host = 0; return QSAE_OK; }
    int rc = bits_args_ok(__func__, x, W, Wq, meta, B, D, H, zbits, words_ld, workspace, workspace_bytes);
    if (rc != QSAE_OK) return rc;
    char* ws = static_cast<char*>(workspace);
    rc = bits_submit(x, W, bias, static_cast<const _Float16*>(Wq), meta, B, D, H, zbits, words_ld, ws, stream);
    if (rc != QSAE_OK) return rc;
    QSAE_HIP(hipMemcpyAsync(flagged_host, ws + bits_layout(B, D, H).flags, sizeof(int), hipMemcpyDeviceToHost, as_stream(stream)));
    return QSAE_OK;
}

extern "C" int qsae_encode_bits_prefilter_finish(const float* x, const float* W, const float* bias, const void* Wq,
                                                 const float* meta, int B, int D, int H, uint32_t* zbits, int64_t words_ld,
                                                 void* workspace, size_t workspace_bytes, int flagged, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    if (B == 0) return QSAE_OK;
    const int rc = bits_args_ok(__func__, x, W, Wq, meta, B, D, H, zbits, words_ld, workspace, workspace_bytes);
    if (rc != QSAE_OK) return rc;
    return bits_finish(x, W, bias, B, D, H, zbits, words_ld, static_cast<char*>(workspace), stream, flagged);
}

/* dense activations: every latent classified by the fp16 pass, the uncertainty band resolved exactly */
extern "C" size_t qsae_encode_bits_band_workspace_bytes(int B, int D, int H) {
    if (!bits_band_shape_ok(B, D, H)) return 0;
    return bits_layout(B, D, H, kBandCap).total;
}

extern "C" int qsae_encode_bits_band(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                                     int B, int D, int H, uint32_t* zbits, int64_t words_ld, void* workspace,
                                     size_t workspace_bytes, int* flagged_rows, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    if (flagged_rows) *flagged_rows = 0;
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(x && W && Wq && meta && zbits, "null pointer");
    QSAE_CHECK_ARG(words_ld >= (H + 31) / 32, "words_ld < ceil(H/32)");
    QSAE_CHECK_SUPPORTED(bits_band_shape_ok(B, D, H), "shape not covered (D % 64 == 0, H % 32 == 0; use qsae_encode_bits)");
    QSAE_CHECK_ARG(aligned16(x) && aligned16(W) && aligned16(Wq), "x, W and Wq must be 16-byte aligned");
    QSAE_CHECK_ARG(workspace && workspace_bytes >= bits_layout(B, D, H, kBandCap).total, "workspace too small");
    QSAE_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "workspace must be 256-byte aligned");
    char* ws = static_cast<char*>(workspace);
    hipStream_t s = as_stream(stream);
    int rc = bits_band_submit(x, W, bias, static_cast<const _Float16*>(Wq), meta, B, D, H, zbits, words_ld, ws, stream);
    if (rc != QSAE_OK) return rc;
    const BitsLayout L = bits_layout(B, D, H, kBandCap);
    ThreadDeviceCtx* ctx = nullptr;
    rc = thread_device_ctx(&ctx);
    if (rc != QSAE_OK) return rc;
    *ctx->pinned = 0;
    QSAE_HIP(hipMemcpyAsync(ctx->pinned, ws + L.flags, sizeof(int), hipMemcpyDeviceToHost, s));
    QSAE_HIP(hipEventRecord(ctx->ev_copied, s));
    QSAE_HIP(hipEventSynchronize(ctx->ev_copied));
    const int nflag = *ctx->pinned;
    if (flagged_rows) *flagged_rows = nflag;
    if (nflag < 0 || nflag > B) return fail(QSAE_ERR_HIP, "%s: corrupt flagged-row count", __func__);
    return bits_finish(x, W, bias, B, D, H, zbits, words_ld, ws, stream, nflag, kBandCap);
}

/* fp32-accurate dense encoder on the fp16 matrix pipe (two-term fp16 split of both operands, three partial contractions) */
extern "C" size_t qsae_emu_w_bytes(int H, int D) {
    return (H > 0 && D > 0 && D % 64 == 0) ? static_cast<size_t>(H) * 3 * D * 2 : 0;
}

extern "C" int qsae_emu_pack_w(const float* W, int H, int D, void* Wc, float* meta2, qsae_stream_t stream) {
    QSAE_CHECK_ARG(H > 0 && D > 0 && W && Wc && meta2, "H > 0, D > 0, non-null pointers");
    QSAE_CHECK_SUPPORTED(D % 64 == 0, "D must be a multiple of 64");
    QSAE_CHECK_ARG(aligned16(Wc), "Wc must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    QSAE_HIP(hipMemsetAsync(meta2, 0, 2 * sizeof(float), s));
    const long long n = static_cast<long long>(H) * D;
    hipLaunchKernelGGL(emu_w_max_kernel, dim3(1024), dim3(256), 0, s, W, n, reinterpret_cast<unsigned*>(meta2 + 1));
    QSAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(emu_pack_w_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, W, H, D, meta2,
                       static_cast<_Float16*>(Wc));
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

extern "C" size_t qsae_encode_dense_emu_workspace_bytes(int B, int D) {
    if (B <= 0 || D <= 0 || D % 64 != 0) return 0;
    return align_up(static_cast<size_t>(B) * 3 * D * 2, 256) + align_up(static_cast<size_t>(B) * 4, 256);
}

extern "C" int qsae_encode_dense_emu(const float* x, const void* Wc, const float* meta2, const float* bias, int B, int D, int H,
                                     int act, float* out, int64_t out_ld, void* workspace, size_t workspace_bytes,
                                     qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(x && Wc && meta2 && out && workspace, "null pointer");
    QSAE_CHECK_SUPPORTED(D % 64 == 0, "D must be a multiple of 64 (use qsae_encode_dense)");
    QSAE_CHECK_ARG(act == QSAE_ACT_NONE || act == QSAE_ACT_RELU || act == QSAE_ACT_SIGMOID, "unknown activation");
    QSAE_CHECK_ARG(out_ld >= H, "out_ld < H");
    QSAE_CHECK_ARG(aligned16(Wc) && (reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "Wc 16-byte, workspace 256-byte aligned");
    QSAE_CHECK_ARG(workspace_bytes >= qsae_encode_dense_emu_workspace_bytes(B, D), "workspace too small");
    hipStream_t s = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    _Float16* Xc = reinterpret_cast<_Float16*>(ws);
    float* inv = reinterpret_cast<float*>(ws + align_up(static_cast<size_t>(B) * 3 * D * 2, 256));
    hipLaunchKernelGGL(emu_x_prep_kernel, dim3((B + 3) / 4), dim3(256), 0, s, x, B, D, meta2, Xc, inv);
    QSAE_LAUNCH_CHECK();
    const float* xw = reinterpret_cast<const float*>(Xc);
    const float* ww = reinterpret_cast<const float*>(Wc);
    const int Kw = 3 * D / 2;
    if (act == QSAE_ACT_RELU) {
        using Epi = EpiEmuDense<QSAE_ACT_RELU, 256, 256, 4, 2>;
        typename Epi::Args ea{inv, bias, out, out_ld};
        return launch_gemm_dma<Epi, 256, 256, true, 2>(xw, B, ww, H, Kw, ea, s, /*sweep=*/8);
    }
    if (act == QSAE_ACT_SIGMOID) {
        using Epi = EpiEmuDense<QSAE_ACT_SIGMOID, 256, 256, 4, 2>;
        typename Epi::Args ea{inv, bias, out, out_ld};
        return launch_gemm_dma<Epi, 256, 256, true, 2>(xw, B, ww, H, Kw, ea, s, /*sweep=*/8);
    }
    using Epi = EpiEmuDense<QSAE_ACT_NONE, 256, 256, 4, 2>;
    typename Epi::Args ea{inv, bias, out, out_ld};
    return launch_gemm_dma<Epi, 256, 256, true, 2>(xw, B, ww, H, Kw, ea, s, /*sweep=*/8);
}

/* the two-call form of qsae_encode_bits_band (see qsae_prefilter_submit / _finish) */
static int bits_band_args_ok(const char* who, const float* x, const float* W, const void* Wq, const float* meta, int B, int D, int H,
                             const uint32_t* zbits, int64_t words_ld, const void* workspace, size_t workspace_bytes) {
    if (!(x && W && Wq && meta && zbits)) return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: null pointer", who);
    if (words_ld < (H + 31) / 32) return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: words_ld < ceil(H/32)", who);
    if (!bits_band_shape_ok(B, D, H))
        return fail(QSAE_ERR_UNSUPPORTED, "%s: unsupported: shape not covered (D %% 64 == 0, H %% 32 == 0; use qsae_encode_bits)", who);
    if (!(aligned16(x) && aligned16(W) && aligned16(Wq)))
        return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: x, W and Wq must be 16-byte aligned", who);
    if (!(workspace && workspace_bytes >= bits_layout(B, D, H, kBandCap).total))
        return fail(QSAE_ERR_WORKSPACE, "%s: workspace too small", who);
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0)
        return fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: workspace must be 256-byte aligned", who);
    return QSAE_OK;
}

extern "C" int qsae_encode_bits_band_submit(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                                            int B, int D, int H, uint32_t* zbits, int64_t words_ld, void* workspace,
                                            size_t workspace_bytes, int* flagged_host, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    QSAE_CHECK_ARG(flagged_host != nullptr, "flagged_host must point to a host int");
    if (B == 0) { *flagged_host = 0; return QSAE_OK; }
    int rc = bits_band_args_ok(__func__, x, W, Wq, meta, B, D, H, zbits, words_ld, workspace, workspace_bytes);
    if (rc != QSAE_OK) return rc;
    char* ws = static_cast<char*>(workspace);
    rc = bits_band_submit(x, W, bias, static_cast<const _Float16*>(Wq), meta, B, D, H, zbits, words_ld, ws, stream);
    if (rc != QSAE_OK) return rc;
    QSAE_HIP(hipMemcpyAsync(flagged_host, ws + bits_layout(B, D, H, kBandCap).flags, sizeof(int), hipMemcpyDeviceToHost,
                            as_stream(stream)));
    return QSAE_OK;
}

extern "C" int qsae_encode_bits_band_finish(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                                            int B, int D, int H, uint32_t* zbits, int64_t words_ld, void* workspace,
                                            size_t workspace_bytes, int flagged, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    if (B == 0) return QSAE_OK;
    const int rc = bits_band_args_ok(__func__, x, W, Wq, meta, B, D, H, zbits, words_ld, workspace, workspace_bytes);
    if (rc != QSAE_OK) return rc;
    return bits_finish(x, W, bias, B, D, H, zbits, words_ld, static_cast<char*>(workspace), stream, flagged, kBandCap);
}

// topk.hip -- exact per-row top-k of a dense latent (value desc, index asc; NaN first).
//
// Replaces `latent.topk(k, dim=1)` + `zeros_like` + `scatter_` (+ `latent * mask`) of the
// reference (sae/binary.py:94-99, sae/baseline.py:34-40).  HBM-bound by design: one
// 256-thread workgroup owns one row, holds it in registers (<= 128 floats per thread), and
//   1. every thread records its largest element (1 compare per element);
//   2. wave 0 finds the k-th largest of the 256 thread maxima by a 32-step bitwise search on
//      ballots -> a threshold t0 with >= k elements at or above it (a valid lower bound of the
//      k-th largest; for i.i.d. data only ~1.15 k elements pass it);
//   3. elements >= t0 are appended to an LDS candidate list (1 compare per element);
//   4. the candidates are ranked exactly by their 64-bit (value, ~index) keys;
//   5. optionally the row is rewritten with everything below the k-th key zeroed.
// Rows whose candidate list would overflow (massive ties, adversarial layouts) take an exact
// 6 x 8-bit radix select over 48-bit keys instead of steps 2-3.
#include "common.h"

namespace qsae {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTopkThreads = 256;
constexpr int kTopkCap = 1024;   // candidate capacity per row

struct TopkShared {
    unsigned long long cand[kTopkCap];
    uint32_t rep[kTopkThreads];
    uint32_t hist[256];
    unsigned long long kth;      // full key of the k-th largest
    uint32_t t0;                 // mono key threshold
    int count;
    int krem;
    unsigned long long prefix;
};

__device__ __forceinline__ unsigned long long key48(float v, uint32_t idx) {
    return (static_cast<unsigned long long>(mono_key(v)) << 16) | (0xFFFFu - (idx & 0xFFFFu));
}

// Pilot mode (tau_out != nullptr): instead of idx/val the kernel emits, per row, tau = the k-th
// largest value and appends EVERY element >= tau (ties included) as (value bits, index) pairs to
// cand[row][0..cap) with their count in cnt[row] -- the seed of the fused encoder+top-k filter.
struct PilotOut {
    float* tau;
    uint2* cand;
    int* cnt;
    int cap;             // entries of the row's list the seeds may use
    int stride;          // entries between the lists of consecutive rows (>= cap)
    float* dense;        // optional: the row's first H (pilot) columns of the dense latent are zero-filled
    int64_t dense_ld;    //   here, on the way (the sweep zero-fills the rest, the survivors are scattered later)
    const float* margin; // optional [B]: candidates are the elements >= tau - margin[row] (approximate pilots)
};

template <int VPT4>
__global__ void __launch_bounds__(kTopkThreads)
topk_rows_kernel(float* __restrict__ latent, int64_t ld, int H, int k, int32_t* __restrict__ idx_out,
                 float* __restrict__ val_out, int zero_rest, PilotOut pilot) {
    __shared__ TopkShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* row = latent + static_cast<int64_t>(blockIdx.x) * ld;
    const float NEG_INF = -__builtin_huge_valf();

    // ---- 1. load the row, track the per-thread maximum -------------------------------
    f32x4 v[VPT4];
    float best = NEG_INF;
#pragma unroll
    for (int i = 0; i < VPT4; ++i) {
        const int e = (i * kTopkThreads + tid) * 4;
        if (e < H) {
            v[i] = *reinterpret_cast<const f32x4*>(row + e);
        } else {
            v[i] = f32x4{NEG_INF, NEG_INF, NEG_INF, NEG_INF};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) best = (v[i][j] > best) ? v[i][j] : best;
    }
    sh.rep[tid] = mono_key(best);
    if (tid == 0) sh.count = 0;
    __syncthreads();

    // ---- 2. wave 0: k-th largest of the 256 thread maxima ------------------------------
    if (wave == 0) {
        const uint32_t m0 = sh.rep[lane], m1 = sh.rep[lane + 64], m2 = sh.rep[lane + 128], m3 = sh.rep[lane + 192];
        uint32_t T = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = T | (1u << bit);
            const int cnt = __popcll(__ballot(m0 >= trial)) + __popcll(__ballot(m1 >= trial)) +
                            __popcll(__ballot(m2 >= trial)) + __popcll(__ballot(m3 >= trial));
            if (cnt >= k) T = trial;
        }
        if (lane == 0) sh.t0 = T;
    }
    __syncthreads();

    // ---- 3. collect candidates ----------------------------------------------------------
    {
        const uint32_t t0 = sh.t0;
        // float form of the threshold; NaN candidates pass through !(x < t).
        uint32_t u = (t0 & 0x80000000u) ? (t0 & 0x7FFFFFFFu) : ~t0;
        const float tf = (t0 == 0xFFFFFFFFu) ? __builtin_huge_valf() : __uint_as_float(u);
#pragma unroll
        for (int i = 0; i < VPT4; ++i) {
            const int e = (i * kTopkThreads + tid) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = v[i][j];
                if (!(x < tf) && (e + j) < H) {
                    const int pos = atomicAdd(&sh.count, 1);
                    if (pos < kTopkCap) sh.cand[pos] = full_key(x, static_cast<uint32_t>(e + j));
                }
            }
        }
    }
    __syncthreads();
    int n = sh.count;

    // ---- fallback: exact radix select over 48-bit keys when the list overflowed --------
    if (n > kTopkCap) {
        if (tid == 0) { sh.prefix = 0ull; sh.krem = k; }
        for (int pass = 0; pass < 6; ++pass) {
            const int shift = 40 - 8 * pass;
            sh.hist[tid] = 0;
            __syncthreads();
            const unsigned long long prefix = sh.prefix;
#pragma unroll
            for (int i = 0; i < VPT4; ++i) {
                const int e = (i * kTopkThreads + tid) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if ((e + j) < H) {
                        const unsigned long long kk = key48(v[i][j], static_cast<uint32_t>(e + j));
                        if (pass == 0 || (kk >> (shift + 8)) == prefix)
                            atomicAdd(&sh.hist[(kk >> shift) & 0xFFull], 1u);
                    }
                }
            }
            __syncthreads();
            if (tid == 0) {
                int rem = sh.krem, d = 255;
                for (; d > 0; --d) {
                    const int c = static_cast<int>(sh.hist[d]);
                    if (c >= rem) break;
                    rem -= c;
                }
                sh.krem = rem;
                sh.prefix = (prefix << 8) | static_cast<unsigned long long>(d);
            }
            __syncthreads();
        }
        const unsigned long long kth48 = sh.prefix;
        if (tid == 0) sh.count = 0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VPT4; ++i) {
            const int e = (i * kTopkThreads + tid) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if ((e + j) < H && key48(v[i][j], static_cast<uint32_t>(e + j)) >= kth48) {
                    const int pos = atomicAdd(&sh.count, 1);
                    if (pos < kTopkCap) sh.cand[pos] = full_key(v[i][j], static_cast<uint32_t>(e + j));
                }
            }
        }
        __syncthreads();
        n = sh.count;   // == k
    }

    // ---- 4. exact rank among the candidates ---------------------------------------------
    for (int c = tid; c < n; c += kTopkThreads) {
        const unsigned long long mine = sh.cand[c];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (sh.cand[j] > mine) ? 1 : 0;
        if (rank < k) {
            const uint32_t e = key_index(mine);
            if (idx_out) {
                // recovering the value from the register-resident row copy is awkward; re-read it
                const float x = row[e];
                idx_out[static_cast<int64_t>(blockIdx.x) * k + rank] = static_cast<int32_t>(e);
                val_out[static_cast<int64_t>(blockIdx.x) * k + rank] = x;
            }
            if (rank == k - 1) sh.kth = mine;
        }
    }
    if (!zero_rest && pilot.tau == nullptr) return;
    __syncthreads();

    if (pilot.tau != nullptr) {
        const uint32_t km = static_cast<uint32_t>(sh.kth >> 32);
        const uint32_t kb = (km & 0x80000000u) ? (km & 0x7FFFFFFFu) : ~km;
        // a NaN k-th value gives tau = +inf: only NaNs (and +inf) pass !(x < tau)
        const float tf = (km == 0xFFFFFFFFu) ? __builtin_huge_valf() : __uint_as_float(kb);
        if (tid == 0) { pilot.tau[blockIdx.x] = tf; sh.count = 0; }
        __syncthreads();
        const float tcut = pilot.margin ? tf - pilot.margin[blockIdx.x] : tf;
        uint2* list = pilot.cand + static_cast<int64_t>(blockIdx.x) * pilot.stride;
#pragma unroll
        for (int i = 0; i < VPT4; ++i) {
            const int e = (i * kTopkThreads + tid) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = v[i][j];
                if ((e + j) < H && !(x < tcut)) {
                    const int pos = atomicAdd(&sh.count, 1);
                    if (pos < pilot.cap) list[pos] = make_uint2(__float_as_uint(x), static_cast<uint32_t>(e + j));
                }
            }
        }
        if (pilot.dense != nullptr) {
            float* drow = pilot.dense + static_cast<int64_t>(blockIdx.x) * pilot.dense_ld;
            for (int e = tid * 4; e < H; e += kTopkThreads * 4)
                *reinterpret_cast<f32x4*>(drow + e) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        if (tid == 0) pilot.cnt[blockIdx.x] = sh.count;
        return;
    }

    // ---- 5. rewrite the row: keep keys >= kth, zero the rest ----------------------------
    const unsigned long long kth = sh.kth;
    const uint32_t kth_mono = static_cast<uint32_t>(kth >> 32);
    uint32_t ku = (kth_mono & 0x80000000u) ? (kth_mono & 0x7FFFFFFFu) : ~kth_mono;
    const float kf = (kth_mono == 0xFFFFFFFFu) ? __builtin_huge_valf() : __uint_as_float(ku);
#pragma unroll
    for (int i = 0; i < VPT4; ++i) {
        const int e = (i * kTopkThreads + tid) * 4;
        if (e >= H) continue;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = v[i][j];
            bool keep = false;
            if (!(x < kf)) keep = full_key(x, static_cast<uint32_t>(e + j)) >= kth;
            o[j] = keep ? x : 0.0f;
        }
        *reinterpret_cast<f32x4*>(row + e) = o;
    }
}

template <int VPT4>
static int launch_topk(float* latent, int64_t ld, int B, int H, int k, int32_t* idx, float* val, int zero_rest,
                       const PilotOut& pilot, hipStream_t s) {
    hipLaunchKernelGGL(topk_rows_kernel<VPT4>, dim3(B), dim3(kTopkThreads), 0, s, latent, ld, H, k, idx, val,
                       zero_rest, pilot);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

// shared by qsae_topk_rows and the pilot stage of qsae_encode_topk (encode_topk.hip)
int topk_rows_dispatch(float* latent, int64_t ld, int B, int H, int k, int32_t* idx, float* val, int zero_rest,
                       float* tau, uint2* cand, int* cnt, int cap, float* dense, int64_t dense_ld, hipStream_t s,
                       const float* margin, int stride) {
    const PilotOut pilot{tau, cand, cnt, cap, stride > 0 ? stride : cap, dense, dense_ld, margin};
    const int per_thread4 = (H + 1023) / 1024;
    if (per_thread4 <= 1) return launch_topk<1>(latent, ld, B, H, k, idx, val, zero_rest, pilot, s);
    if (per_thread4 <= 2) return launch_topk<2>(latent, ld, B, H, k, idx, val, zero_rest, pilot, s);
    if (per_thread4 <= 4) return launch_topk<4>(latent, ld, B, H, k, idx, val, zero_rest, pilot, s);
    if (per_thread4 <= 8) return launch_topk<8>(latent, ld, B, H, k, idx, val, zero_rest, pilot, s);
    if (per_thread4 <= 16) return launch_topk<16>(latent, ld, B, H, k, idx, val, zero_rest, pilot, s);
    return launch_topk<32>(latent, ld, B, H, k, idx, val, zero_rest, pilot, s);
}

}  // namespace qsae

using namespace qsae;

extern "C" int qsae_topk_rows(float* latent, int64_t ld, int B, int H, int k, int32_t* idx, float* val,
                              int zero_rest, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && H > 0, "B >= 0 and H > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(latent && idx && val, "null pointer");
    QSAE_CHECK_ARG(k >= 1 && k <= H, "1 <= k <= H required");
    QSAE_CHECK_ARG(ld >= H, "ld < H");
    QSAE_CHECK_SUPPORTED(k <= 256, "k <= 256");
    QSAE_CHECK_SUPPORTED(H <= 32768, "H <= 32768");
    QSAE_CHECK_SUPPORTED(H % 4 == 0 && ld % 4 == 0, "H and ld must be multiples of 4");
    QSAE_CHECK_ARG(aligned16(latent), "latent must be 16-byte aligned");
    return topk_rows_dispatch(latent, ld, B, H, k, idx, val, zero_rest, nullptr, nullptr, nullptr, 0, nullptr, 0,
                              as_stream(stream), nullptr, 0);
}

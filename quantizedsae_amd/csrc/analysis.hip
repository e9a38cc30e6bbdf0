// analysis.hip -- consumers of the sparse latent: activation counts and co-activation counts
// (reference: scripts/analysis/dynamic_analysis.py:255-311 compute_activation_stats, :314-440 analyze_dataset).
//
// The reference turns the dense [B, H] latent into a boolean mask and forms mask.sum(0) and mask^T @ mask (a dense
// [H, B] x [B, H] product per batch, 2 B H^2 FLOP, result copied to the host).  From the compact (idx, val) form
// the same integers are k increments / k^2 increments per row: HBM-atomic-bound integer work.
#include "common.h"

namespace qsae {

constexpr int kCoactMaxK = 256;

// counts[idx[b][j]] += 1 for every entry with val > 0 (val == nullptr: every entry)
__global__ void __launch_bounds__(256)
activation_counts_kernel(const int32_t* __restrict__ idx, const float* __restrict__ val, long long total, int H,
                         unsigned long long* __restrict__ counts) {
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int h = idx[i];
        const bool on = val ? (val[i] > 0.0f) : true;
        if (on && h >= 0 && h < H) atomicAdd(&counts[h], 1ull);
    }
}

// counts[bit position] += 1 for every set bit of the packed rows (bit j of word w = unit 32 w + j)
__global__ void __launch_bounds__(256)
activation_counts_bits_kernel(const uint32_t* __restrict__ zbits, int64_t words_ld, int B, int words,
                              unsigned long long* __restrict__ counts) {
    // one thread per word column and row group: threads of a workgroup share the word column range, rows strided
    const int w = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rgroup = threadIdx.x >> 6;                    // 4 row groups per workgroup
    if (w >= words) return;
    unsigned local[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) local[j] = 0u;
    for (int b = blockIdx.y * 4 + rgroup; b < B; b += gridDim.y * 4) {
        const uint32_t v = zbits[static_cast<int64_t>(b) * words_ld + w];
#pragma unroll
        for (int j = 0; j < 32; ++j) local[j] += (v >> j) & 1u;
    }
#pragma unroll
    for (int j = 0; j < 32; ++j)
        if (local[j]) atomicAdd(&counts[static_cast<size_t>(w) * 32 + j], static_cast<unsigned long long>(local[j]));
}

// coact[a][b] += 1 for every ordered pair (a, b) of a row's active units (diagonal included): mask^T @ mask.
// One wave per row: the active indices are compacted into LDS, the m^2 pairs are dealt to the lanes.
__global__ void __launch_bounds__(256)
coactivation_sparse_kernel(const int32_t* __restrict__ idx, const float* __restrict__ val, int B, int k, int H,
                           int* __restrict__ coact, int64_t ld) {
    __shared__ int act[4][kCoactMaxK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    int m = 0;
    for (int j0 = 0; j0 < k; j0 += 64) {
        const int j = j0 + lane;
        int h = -1;
        if (j < k) {
            h = idx[static_cast<int64_t>(b) * k + j];
            const bool on = val ? (val[static_cast<int64_t>(b) * k + j] > 0.0f) : true;
            if (!on || h < 0 || h >= H) h = -1;
        }
        const unsigned long long msk = __ballot(h >= 0);
        if (h >= 0) act[wave][m + __popcll(msk & ((1ull << lane) - 1ull))] = h;
        m += __popcll(msk);
    }
    asm volatile("" ::: "memory");                           // one wave's LDS operations execute in order
    const int pairs = m * m;
    for (int p = lane; p < pairs; p += 64) {
        const int a = act[wave][p / m], c = act[wave][p % m];
        atomicAdd(&coact[static_cast<int64_t>(a) * ld + c], 1);
    }
}

// n-bit activation quantizer of the reference's binary datasets (src/quantized_sae/data/dataset.py:76-102):
//   unsigned (quantize):        q = int(round(clamp((x * sf) * 2 + 2^(n-1), 0, 2^n - 1)))
//   signed   (quantize_signed): q = int(round(clamp(x * sf, -2^(n-1), 2^(n-1) - 1))) & (2^n - 1)
// out[b][d * n + j] = bit j of q (LSB first) as 0.0 / 1.0; every fp32 operation rounded separately, round =
// half-to-even (torch.round).  One thread per element.
__global__ void __launch_bounds__(256)
quantize_bits_kernel(const float* __restrict__ x, int64_t ld, long long total, int D, int n_bits, float sf,
                     int is_signed, float* __restrict__ out) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long b = i / D;
    const int d = static_cast<int>(i - b * D);
    const float v = x[b * ld + d];
    float t = v * sf;
    const float half_range = static_cast<float>(1 << (n_bits - 1));
    float lo, hi;
    if (is_signed) {
        lo = -half_range;
        hi = half_range - 1.0f;
    } else {
        t = t * 2.0f;
        t = t + half_range;
        lo = 0.0f;
        hi = static_cast<float>((1 << n_bits) - 1);
    }
    // torch.clamp: NaN stays NaN (and becomes INT_MIN in .int()); min/max otherwise
    float c = t;
    if (c < lo) c = lo;
    if (c > hi) c = hi;
    const int q = (c != c) ? static_cast<int>(0x80000000u) : static_cast<int>(rintf(c));
    const unsigned u = static_cast<unsigned>(q) & ((1u << n_bits) - 1u);
    float* o = out + i * n_bits;
    for (int j = 0; j < n_bits; ++j) o[j] = static_cast<float>((u >> j) & 1u);
}

}  // namespace qsae

using namespace qsae;

extern "C" int qsae_activation_counts(const int32_t* idx, const float* val, int B, int k, int H,
                                      unsigned long long* counts, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && k >= 1 && H > 0, "B >= 0, k >= 1, H > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(idx && counts, "null pointer");
    const long long total = static_cast<long long>(B) * k;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(activation_counts_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream),
                       idx, val, total, H, counts);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

extern "C" int qsae_activation_counts_bits(const uint32_t* zbits, int64_t words_ld, int B, int nbits,
                                           unsigned long long* counts, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && nbits > 0, "B >= 0, nbits > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(zbits && counts, "null pointer");
    QSAE_CHECK_SUPPORTED(nbits % 32 == 0, "nbits must be a multiple of 32");
    const int words = nbits / 32;
    QSAE_CHECK_ARG(words_ld >= words, "words_ld < nbits / 32");
    int ygroups = (B + 3) / 4;
    if (ygroups > 256) ygroups = 256;
    hipLaunchKernelGGL(activation_counts_bits_kernel, dim3((words + 63) / 64, ygroups), dim3(256), 0, as_stream(stream),
                       zbits, words_ld, B, words, counts);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

extern "C" int qsae_coactivation_sparse(const int32_t* idx, const float* val, int B, int k, int H, int32_t* coact,
                                        int64_t ld, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && k >= 1 && H > 0, "B >= 0, k >= 1, H > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(idx && coact, "null pointer");
    QSAE_CHECK_ARG(ld >= H, "ld < H");
    QSAE_CHECK_SUPPORTED(k <= kCoactMaxK, "k <= 256");
    hipLaunchKernelGGL(coactivation_sparse_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), idx, val, B, k,
                       H, coact, ld);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

extern "C" int qsae_quantize_bits(const float* x, int64_t ld, int B, int D, int n_bits, float scale_factor,
                                  int is_signed, float* bits, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0, "B >= 0, D > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(x && bits, "null pointer");
    QSAE_CHECK_ARG(ld >= D, "ld < D");
    QSAE_CHECK_SUPPORTED(n_bits >= 1 && n_bits <= 16, "1 <= n_bits <= 16");
    const long long total = static_cast<long long>(B) * D;
    hipLaunchKernelGGL(quantize_bits_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), x, ld, total, D, n_bits, scale_factor, is_signed, bits);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

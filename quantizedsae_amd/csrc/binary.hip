// binary.hip -- BinarySAE n-bit two's-complement dictionary: packer, sparse decode, exports.
//
// Reference: binary_decoder in sae/binary.py:10-69.  The reference keeps the dictionary as
// [H, D*n_bits] fp32 logits (256 MiB at 32768x512x4), re-derives int_weights from them on
// every forward and multiplies the 99.8%-zero dense latent with the full [H, D] matrix.
// Here the dictionary is packed once to n-bit fields (8 MiB, row h contiguous = one 256-byte
// gather per selected feature) and the decode touches only the k selected rows.
#include "decode_row.h"

namespace qsae {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- packer -----------------------------------------------------------------------------
// One thread per output dword: 32/fw fields, each from n_bits logits.
// soft_gap (optional): max over all (h, d) of |soft - hard|, soft = sum_b sigmoid(logit_b) bw_b (what the reference's
// forward multiplies with, sae/binary.py:26-35), hard = the packed two's-complement integer (sae/binary.py:49-58).
// It is how far the reference forward is from a hard-bit decode of this checkpoint, in units of one integer step:
// ~1e-12 for +-30 logits, 0.5 for logits near 0.  NaN logits give +inf.
__global__ void __launch_bounds__(256)
pack_binary_kernel(const float* __restrict__ logits, int H, int D, int n, int fw, int row_dwords,
                   uint32_t* __restrict__ packed, double* __restrict__ polarize_sum, unsigned* __restrict__ soft_gap) {
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const long long total = static_cast<long long>(H) * row_dwords;
    double psum = 0.0;
    float gap = 0.0f;
    const bool stats = polarize_sum != nullptr || soft_gap != nullptr;
    if (gid < total) {
        const int h = static_cast<int>(gid / row_dwords), c = static_cast<int>(gid % row_dwords);
        const int F = 32 / fw;
        const float* lrow = logits + static_cast<long long>(h) * D * n;
        uint32_t word = 0;
        for (int f = 0; f < F; ++f) {
            const int d = c * F + f;
            if (d >= D) break;
            uint32_t code = 0;
            float soft = 0.0f;
            for (int b = 0; b < n; ++b) {
                const float w = lrow[d * n + b];
                code |= (sig_gt_half(w) ? 1u : 0u) << b;
                if (stats) {
                    const float p = 1.0f / (1.0f + expf(-w));
                    psum += static_cast<double>(p * (1.0f - p) * static_cast<float>(1u << b));
                    soft = soft + p * ((b == n - 1) ? -static_cast<float>(1u << b) : static_cast<float>(1u << b));
                }
            }
            if (stats) {
                const float hard = static_cast<float>(sbfe_i32(static_cast<int>(code), 0, n));
                const float g = fabsf(soft - hard);
                gap = (g > gap || g != g) ? (g != g ? __builtin_huge_valf() : g) : gap;
            }
            word |= code << (f * fw);
        }
        packed[gid] = word;
    }
    if (stats) {
        // wave reduce, then one atomic per wave
        for (int off = 32; off > 0; off >>= 1) {
            psum += __shfl_down(psum, off, 64);
            gap = fmaxf(gap, __shfl_down(gap, off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            if (polarize_sum) atomicAdd(polarize_sum, psum);
            if (soft_gap) atomicMax(soft_gap, __float_as_uint(gap));    // non-negative floats order like their bit patterns
        }
    }
}

__global__ void __launch_bounds__(256)
unpack_binary_kernel(const uint32_t* __restrict__ packed, int H, int D, int n, int fw, int row_dwords,
                     float* __restrict__ out) {
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= static_cast<long long>(H) * D) return;
    const int h = static_cast<int>(gid / D), d = static_cast<int>(gid % D);
    const int F = 32 / fw;
    const uint32_t word = packed[static_cast<long long>(h) * row_dwords + d / F];
    out[gid] = static_cast<float>(sbfe_i32(static_cast<int>(word), (d % F) * fw, n));
}

// table[h][d] = sum_b sigmoid(logit[h][d*n+b]) * bw[b], bw = [1,2,..,-2^(n-1)]  (binary.py:26-35)
__global__ void __launch_bounds__(256)
soft_table_kernel(const float* __restrict__ logits, int H, int D, int n, float* __restrict__ table) {
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= static_cast<long long>(H) * D) return;
    const float* l = logits + gid * n;
    float acc = 0.0f;
    for (int b = 0; b < n; ++b) {
        const float p = 1.0f / (1.0f + expf(-l[b]));
        const float bw = (b == n - 1) ? -static_cast<float>(1u << b) : static_cast<float>(1u << b);
        acc = acc + p * bw;
    }
    table[gid] = acc;
}

// ---- sparse decode -------------------------------------------------------------------------
// One wave per activation row; 4 rows per workgroup.  The k (idx,val) pairs are sorted by idx
// in LDS (rank counting) so that the fmaf chain runs in ascending feature index exactly like the
// oracle; then each lane owns one dword of the packed row (32/fw output dims) per sweep.
constexpr int kDecWaves = 4;
constexpr int kDecMaxK = 256;

struct DecShared {
    int idx[kDecWaves][kDecMaxK];
    float val[kDecWaves][kDecMaxK];
};

// All threads of the workgroup call this (it contains workgroup barriers); waves whose row is
// out of range pass active = false and only take part in the barriers.
__device__ __forceinline__ void sort_pairs_by_index(bool active, const int32_t* __restrict__ idx,
                                                    const float* __restrict__ val, int k, int H, int lane,
                                                    int* s_idx, float* s_val, int* t_idx) {
    if (active) {
        for (int j = lane; j < k; j += 64) {
            int h = idx[j];
            h = h < 0 ? 0 : (h >= H ? H - 1 : h);   // never index outside the dictionary
            t_idx[j] = h;
        }
    }
    __syncthreads();
    if (active) {
        for (int j = lane; j < k; j += 64) {
            const int mine = t_idx[j];
            int rank = 0;
            for (int i = 0; i < k; ++i) {
                const int o = t_idx[i];
                rank += (o < mine || (o == mine && i < j)) ? 1 : 0;
            }
            s_idx[rank] = mine;
            s_val[rank] = val[j];
        }
    }
    __syncthreads();
}

// rows != nullptr: row b of this launch is activation row rows[b] (the flagged rows of the prefilter pipeline)
template <int FW>
__global__ void __launch_bounds__(64 * kDecWaves)
decode_binary_sparse_kernel(const int32_t* __restrict__ idx, const float* __restrict__ val, int B, int k, int H,
                            RowDecode d, const int* __restrict__ rows) {
    __shared__ DecShared sh;
    __shared__ int tmp_idx[kDecWaves][kDecMaxK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = blockIdx.x * kDecWaves + wave;
    const bool active = slot < B;
    const long long b = active ? (rows ? rows[slot] : slot) : 0;
    int* s_idx = sh.idx[wave];
    float* s_val = sh.val[wave];
    const long long off = b * k;
    sort_pairs_by_index(active, idx + off, val + off, k, H, lane, s_idx, s_val, tmp_idx[wave]);
    if (!active) return;
    // 4- and 8-bit fields: all of a chunk's dictionary rows in flight (and the 13-instruction nibble form), as in the refinement
    if (FW == 4) decode_row_sorted_wide4(s_idx, s_val, k, d, b, lane);
    else if (FW == 8) decode_row_sorted_wide<8>(s_idx, s_val, k, d, b, lane);
    else decode_row_sorted<FW>(s_idx, s_val, k, d, b, lane);
}

// rows != nullptr: row b of this launch is activation row rows[b] (as above)
__global__ void __launch_bounds__(64 * kDecWaves)
decode_table_sparse_kernel(const int32_t* __restrict__ idx, const float* __restrict__ val, int B, int k, int H,
                           RowDecode d, const int* __restrict__ rows) {
    __shared__ DecShared sh;
    __shared__ int tmp_idx[kDecWaves][kDecMaxK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = blockIdx.x * kDecWaves + wave;
    const bool active = slot < B;
    const long long b = active ? (rows ? rows[slot] : slot) : 0;
    int* s_idx = sh.idx[wave];
    float* s_val = sh.val[wave];
    const long long off = b * k;
    sort_pairs_by_index(active, idx + off, val + off, k, H, lane, s_idx, s_val, tmp_idx[wave]);
    if (!active) return;
    decode_row_table(s_idx, s_val, k, d, b, lane);
}

}  // namespace qsae

using namespace qsae;

extern "C" int qsae_binary_row_bytes(int D, int n_bits) {
    if (D <= 0 || n_bits < 1 || n_bits > 8) return QSAE_ERR_INVALID_ARG;
    return ((D * field_width(n_bits) + 31) / 32) * 4;
}

extern "C" int qsae_pack_binary(const float* logits, int H, int D, int n_bits, uint8_t* packed,
                                double* polarize_sum, float* soft_gap, qsae_stream_t stream) {
    QSAE_CHECK_ARG(H > 0 && D > 0, "H > 0 and D > 0 required");
    QSAE_CHECK_ARG(n_bits >= 1 && n_bits <= 8, "1 <= n_bits <= 8 required");
    QSAE_CHECK_ARG(logits && packed, "null pointer");
    QSAE_CHECK_ARG((reinterpret_cast<uintptr_t>(packed) & 3u) == 0, "packed must be 4-byte aligned");
    const int fw = field_width(n_bits), row_dwords = qsae_binary_row_bytes(D, n_bits) / 4;
    const long long total = static_cast<long long>(H) * row_dwords;
    hipStream_t s = as_stream(stream);
    if (polarize_sum) QSAE_HIP(hipMemsetAsync(polarize_sum, 0, sizeof(double), s));
    if (soft_gap) QSAE_HIP(hipMemsetAsync(soft_gap, 0, sizeof(float), s));
    hipLaunchKernelGGL(pack_binary_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, logits,
                       H, D, n_bits, fw, row_dwords, reinterpret_cast<uint32_t*>(packed), polarize_sum,
                       reinterpret_cast<unsigned*>(soft_gap));
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

extern "C" int qsae_unpack_binary(const uint8_t* packed, int H, int D, int n_bits, float* int_weights,
                                  qsae_stream_t stream) {
    QSAE_CHECK_ARG(H > 0 && D > 0, "H > 0 and D > 0 required");
    QSAE_CHECK_ARG(n_bits >= 1 && n_bits <= 8, "1 <= n_bits <= 8 required");
    QSAE_CHECK_ARG(packed && int_weights, "null pointer");
    const int fw = field_width(n_bits), row_dwords = qsae_binary_row_bytes(D, n_bits) / 4;
    const long long total = static_cast<long long>(H) * D;
    hipLaunchKernelGGL(unpack_binary_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), reinterpret_cast<const uint32_t*>(packed), H, D, n_bits, fw, row_dwords,
                       int_weights);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

extern "C" int qsae_binary_soft_table(const float* logits, int H, int D, int n_bits, float* table,
                                      qsae_stream_t stream) {
    QSAE_CHECK_ARG(H > 0 && D > 0, "H > 0 and D > 0 required");
    QSAE_CHECK_ARG(n_bits >= 1 && n_bits <= 8, "1 <= n_bits <= 8 required");
    QSAE_CHECK_ARG(logits && table, "null pointer");
    const long long total = static_cast<long long>(H) * D;
    hipLaunchKernelGGL(soft_table_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), logits, H, D, n_bits, table);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

namespace qsae {
// rows == nullptr: all B rows; otherwise the nrows rows listed in `rows` (device array)
int decode_binary_sparse_rows(const int* rows, int nrows, const int32_t* idx, const float* val, int k, int H,
                              const RowDecode& d, hipStream_t s) {
    if (nrows <= 0) return QSAE_OK;
    const dim3 grid((nrows + kDecWaves - 1) / kDecWaves), block(64 * kDecWaves);
    if (d.table) {          // fp32 dictionary rows (Baseline, BinarySAE soft integers)
        hipLaunchKernelGGL(decode_table_sparse_kernel, grid, block, 0, s, idx, val, nrows, k, H, d, rows);
        QSAE_LAUNCH_CHECK();
        return QSAE_OK;
    }
    switch (d.fw) {
        case 1: hipLaunchKernelGGL(decode_binary_sparse_kernel<1>, grid, block, 0, s, idx, val, nrows, k, H, d, rows); break;
        case 2: hipLaunchKernelGGL(decode_binary_sparse_kernel<2>, grid, block, 0, s, idx, val, nrows, k, H, d, rows); break;
        case 4: hipLaunchKernelGGL(decode_binary_sparse_kernel<4>, grid, block, 0, s, idx, val, nrows, k, H, d, rows); break;
        default: hipLaunchKernelGGL(decode_binary_sparse_kernel<8>, grid, block, 0, s, idx, val, nrows, k, H, d, rows); break;
    }
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}
}  // namespace qsae

extern "C" int qsae_decode_binary_sparse(const int32_t* idx, const float* val, int B, int k,
                                         const uint8_t* packed, int H, int D, int n_bits, float step,
                                         const float* bias, float* recon, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && H > 0 && D > 0, "B >= 0, H > 0, D > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(n_bits >= 1 && n_bits <= 8, "1 <= n_bits <= 8 required");
    QSAE_CHECK_ARG(idx && val && packed && recon, "null pointer");
    QSAE_CHECK_ARG(k >= 1, "k >= 1 required");
    QSAE_CHECK_SUPPORTED(k <= kDecMaxK, "k <= 256");
    const RowDecode d{reinterpret_cast<const uint32_t*>(packed), qsae_binary_row_bytes(D, n_bits) / 4, n_bits,
                      field_width(n_bits), D, step, bias, recon, nullptr};
    return decode_binary_sparse_rows(nullptr, B, idx, val, k, H, d, as_stream(stream));
}

extern "C" int qsae_decode_table_sparse(const int32_t* idx, const float* val, int B, int k, const float* table,
                                        int H, int D, float scale, const float* bias, float* recon,
                                        qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && H > 0 && D > 0, "B >= 0, H > 0, D > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(idx && val && table && recon, "null pointer");
    QSAE_CHECK_ARG(k >= 1, "k >= 1 required");
    QSAE_CHECK_SUPPORTED(k <= kDecMaxK, "k <= 256");
    QSAE_CHECK_SUPPORTED(D % 4 == 0, "D must be a multiple of 4");
    QSAE_CHECK_ARG(aligned16(table) && aligned16(recon), "table and recon must be 16-byte aligned");
    const RowDecode d{nullptr, 0, 0, 0, D, scale, bias, recon, table};
    return decode_binary_sparse_rows(nullptr, B, idx, val, k, H, d, as_stream(stream));
}

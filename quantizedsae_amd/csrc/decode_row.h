// decode_row.h -- one activation row of the BinarySAE sparse decode (reference sae/binary.py:38 evaluated on the k
// kept entries): shared by the stand-alone decode kernel (binary.hip) and the refinement kernel of the prefilter
// pipeline (encode_topk.hip), which decodes a row as soon as it has ranked it.
#pragma once

#include "common.h"

namespace qsae {

struct RowDecode {
    const uint32_t* packed;   // [H][row_dwords] n-bit two's-complement fields (qsae_pack_binary); nullptr = no packed dictionary
    int row_dwords, n, fw, D;
    float step;               // packed: quantization step; table: scale (1 = no multiply)
    const float* bias;        // [D] or nullptr
    float* recon;             // [B][D]
    const float* table;       // [H][D] fp32 dictionary rows (Baseline decoder.weight^T, BinarySAE soft integers) instead of
                              // `packed`; D % 4 == 0, 16-byte aligned.  Both nullptr = no decode.
    __host__ __device__ bool active() const { return packed != nullptr || table != nullptr; }
};

typedef float dec_f32x4 __attribute__((ext_vector_type(4)));

// The fp32-table form of the row decode (qsae_decode_table_sparse): lane l owns output columns 4 (l + 64 i) .. + 3, four
// dictionary rows in flight, the fmaf chain in ascending feature index; `* scale` (skipped when scale == 1) and `+ bias`
// rounded separately (sae/binary.py:38; sae/baseline.py:29 has no scale).
__device__ __forceinline__ void decode_row_table(const int* s_idx, const float* s_val, int k, const RowDecode& d,
                                                 long long b, int lane) {
    const int D = d.D, D4 = D / 4;
    const bool mul = (d.step != 1.0f);
    for (int c = lane; c < D4; c += 64) {
        dec_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int j = 0;
        for (; j + 4 <= k; j += 4) {
            dec_f32x4 w[4];
            float a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w[u] = *reinterpret_cast<const dec_f32x4*>(d.table + static_cast<long long>(s_idx[j + u]) * D + 4 * c);
                a[u] = s_val[j + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(a[u], w[u][e], acc[e]);
        }
        for (; j < k; ++j) {
            const dec_f32x4 w = *reinterpret_cast<const dec_f32x4*>(d.table + static_cast<long long>(s_idx[j]) * D + 4 * c);
            const float a = s_val[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(a, w[e], acc[e]);
        }
        dec_f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float r = mul ? d.step * acc[e] : acc[e];
            r = r + (d.bias ? d.bias[4 * c + e] : 0.0f);
            o[e] = r;
        }
        *reinterpret_cast<dec_f32x4*>(d.recon + b * D + 4 * c) = o;
    }
}

// s_idx ascending, s_val the matching values (LDS or any memory the whole wave can read).  Every lane owns one dword
// of the packed row (32/FW output columns) per sweep; the fmaf chain runs in ascending feature index like the
// oracle's; `* step` and `+ bias` are rounded separately (binary.py:38).
template <int FW, int U = 4>   // U: dictionary rows in flight
__device__ __forceinline__ void decode_row_sorted(const int* s_idx, const float* s_val, int k, const RowDecode& d,
                                                  long long b, int lane) {
    constexpr int F = 32 / FW;
    for (int c = lane; c < d.row_dwords; c += 64) {
        float acc[F];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = 0.0f;
        int j = 0;
        for (; j + U <= k; j += U) {
            uint32_t w[U];
            float a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                w[u] = d.packed[static_cast<long long>(s_idx[j + u]) * d.row_dwords + c];
                a[u] = s_val[j + u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int f = 0; f < F; ++f)
                    acc[f] = fmaf(a[u], static_cast<float>(sbfe_i32(static_cast<int>(w[u]), f * FW, d.n)), acc[f]);
        }
        for (; j < k; ++j) {
            const uint32_t w = d.packed[static_cast<long long>(s_idx[j]) * d.row_dwords + c];
            const float a = s_val[j];
#pragma unroll
            for (int f = 0; f < F; ++f)
                acc[f] = fmaf(a, static_cast<float>(sbfe_i32(static_cast<int>(w), f * FW, d.n)), acc[f]);
        }
        float* out = d.recon + b * d.D + c * F;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int col = c * F + f;
            if (col < d.D) {
                float r = d.step * acc[f];          // rounded multiply, then rounded add (binary.py:38)
                r = r + (d.bias ? d.bias[col] : 0.0f);
                out[f] = r;
            }
        }
    }
}

template <int U>
__device__ __forceinline__ void decode_row_sorted_any(const int* s_idx, const float* s_val, int k, const RowDecode& d,
                                                      long long b, int lane) {
    if (d.table) { decode_row_table(s_idx, s_val, k, d, b, lane); return; }     // wave-uniform
    switch (d.fw) {                                  // wave-uniform
        case 1: decode_row_sorted<1, U>(s_idx, s_val, k, d, b, lane); break;
        case 2: decode_row_sorted<2, U>(s_idx, s_val, k, d, b, lane); break;
        case 4: decode_row_sorted<4, U>(s_idx, s_val, k, d, b, lane); break;
        default: decode_row_sorted<8, U>(s_idx, s_val, k, d, b, lane); break;
    }
}

}  // namespace qsae

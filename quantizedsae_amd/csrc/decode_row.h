// decode_row.h -- one activation row of the BinarySAE sparse decode (reference sae/binary.py:38 evaluated on the k
// kept entries): shared by the stand-alone decode kernel (binary.hip) and the refinement kernel of the prefilter
// pipeline (encode_topk.hip), which decodes a row as soon as it has ranked it.
#pragma once

#include "common.h"

namespace qsae {

struct RowDecode {
    const uint32_t* packed;   // [H][row_dwords] n-bit two's-complement fields (qsae_pack_binary); nullptr = no decode
    int row_dwords, n, fw, D;
    float step;
    const float* bias;        // [D] or nullptr
    float* recon;             // [B][D]
};

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
    switch (d.fw) {                                  // wave-uniform
        case 1: decode_row_sorted<1, U>(s_idx, s_val, k, d, b, lane); break;
        case 2: decode_row_sorted<2, U>(s_idx, s_val, k, d, b, lane); break;
        case 4: decode_row_sorted<4, U>(s_idx, s_val, k, d, b, lane); break;
        default: decode_row_sorted<8, U>(s_idx, s_val, k, d, b, lane); break;
    }
}

}  // namespace qsae

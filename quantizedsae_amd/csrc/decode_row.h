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

// The same row decode with up to 64 dictionary rows in flight per lane (the refinement kernel's form for 4- and 8-bit fields:
// by the time it decodes, the registers of its chain staging are free).  With four rows in flight a row of k = 64 is sixteen
// dependent round trips to L2 / the memory-side cache -- a third of the refinement's wave time by its phase stamps.  Here lane l
// holds entry j0 + l of the sorted list, the row addresses are formed on the scalar side (v_readlane -> SGPR base, lane offset in
// one VGPR), the values enter the fmaf as SGPR operands, and a chunk's 32 + 32 loads are issued before the first is consumed.
// Padding entries (beyond k) re-read the chunk's first row with value 0: fmaf(0, w, acc) leaves acc as it is (acc is never
// -0: it starts at +0 and (+0) + (-0) = +0), so the chain is the oracle's, bit for bit.
template <int FW>
__device__ __forceinline__ void decode_row_sorted_wide(const int* s_idx, const float* s_val, int k, const RowDecode& d,
                                                       long long b, int lane) {
    constexpr int F = 32 / FW;
    static_assert(F <= 8, "the unrolled chunk is 64 F multiply-adds: fields of 4 bits or more only");
    // EVERY lane runs every round of this loop (lanes past the row's last dword re-read it and store nothing): the list
    // entries travel between lanes through v_readlane, which reads a lane's register whether or not that lane is active --
    // a lane that had left the loop would hand out whatever its register held
    for (int c0 = 0; c0 < d.row_dwords; c0 += 64) {
        const bool live = c0 + lane < d.row_dwords;
        const int c = live ? c0 + lane : d.row_dwords - 1;
        float acc[F];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = 0.0f;
        for (int j0 = 0; j0 < k; j0 += 64) {
            const int jl = j0 + lane;
            const int myi = s_idx[jl < k ? jl : j0];
            const float mya = jl < k ? s_val[jl] : 0.0f;
            const bool two = (k - j0) > 32;                              // wave-uniform
            uint32_t w0[32], w1[32];
#pragma unroll
            for (int u = 0; u < 32; ++u)
                w0[u] = d.packed[static_cast<long long>(__builtin_amdgcn_readlane(myi, u)) * d.row_dwords + c];
            if (two) {
#pragma unroll
                for (int u = 0; u < 32; ++u)
                    w1[u] = d.packed[static_cast<long long>(__builtin_amdgcn_readlane(myi, 32 + u)) * d.row_dwords + c];
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mya), u));
#pragma unroll
                for (int f = 0; f < F; ++f)
                    acc[f] = fmaf(a, static_cast<float>(sbfe_i32(static_cast<int>(w0[u]), f * FW, d.n)), acc[f]);
            }
            if (two) {
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mya), 32 + u));
#pragma unroll
                    for (int f = 0; f < F; ++f)
                        acc[f] = fmaf(a, static_cast<float>(sbfe_i32(static_cast<int>(w1[u]), f * FW, d.n)), acc[f]);
                }
            }
        }
        float* out = d.recon + b * d.D + c * F;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int col = c * F + f;
            if (live && col < d.D) {
                float r = d.step * acc[f];          // rounded multiply, then rounded add (binary.py:38)
                r = r + (d.bias ? d.bias[col] : 0.0f);
                out[f] = r;
            }
        }
    }
}

// 4-bit fields holding 4-bit values (n = 4): the same chain with 13 instead of 24 VALU instructions per packed dword.
// v_cvt_off_f32_i4 turns the low nibble of a byte (SDWA byte select) into nibble / 16 in ONE instruction (no bit-field extract,
// no integer convert); one shift brings the odd nibbles into that position.  The factor comes back through the other operand:
// fmaf(16 a, nibble / 16, acc) has the same exact product as fmaf(a, nibble, acc), hence the same rounded sum (16 a is exact
// unless it overflows: the caller keeps rows with |a| >= 2^123 on the plain form).  Columns 2p, 2p + 1 of a dword share one
// v_pk_fma_f32 (two independent IEEE fp32 FMAs per instruction).
typedef float dec_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float cvt_off_nibble(uint32_t w, int byte) {        // low nibble of byte `byte`, as signed / 16
    float r;
    if (byte == 0) asm("v_cvt_off_f32_i4_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(r) : "v"(w));
    else if (byte == 1) asm("v_cvt_off_f32_i4_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(r) : "v"(w));
    else if (byte == 2) asm("v_cvt_off_f32_i4_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(r) : "v"(w));
    else asm("v_cvt_off_f32_i4_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(r) : "v"(w));
    return r;
}

__device__ __forceinline__ void decode_row_sorted_wide_i4(const int* s_idx, const float* s_val, int k, const RowDecode& d,
                                                          long long b, int lane) {
    for (int c0 = 0; c0 < d.row_dwords; c0 += 64) {               // every lane runs every round (see decode_row_sorted_wide)
        const bool live = c0 + lane < d.row_dwords;
        const int c = live ? c0 + lane : d.row_dwords - 1;
        dec_f32x2 acc[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = dec_f32x2{0.0f, 0.0f};
        for (int j0 = 0; j0 < k; j0 += 64) {
            const int jl = j0 + lane;
            const int myi = s_idx[jl < k ? jl : j0];
            const float mya = jl < k ? 16.0f * s_val[jl] : 0.0f;
            const int rem = k - j0;                                      // wave-uniform; entries of this chunk: min(rem, 64)
            // eight groups of eight entries, each group and each entry behind a wave-uniform guard (k = 65 leaves one entry for the
            // second chunk: one load and 13 instructions, not 32 padded entries); all loads of the chunk are issued before the first
            // is consumed
            uint32_t w[64];
#pragma unroll
            for (int g = 0; g < 8; ++g)
                if (rem > 8 * g) {
#pragma unroll
                    for (int u = 8 * g; u < 8 * g + 8; ++u)
                        if (rem > u) w[u] = d.packed[static_cast<long long>(__builtin_amdgcn_readlane(myi, u)) * d.row_dwords + c];
                }
#pragma unroll
            for (int g = 0; g < 8; ++g)
                if (rem > 8 * g) {
#pragma unroll
                    for (int u = 8 * g; u < 8 * g + 8; ++u) {
                        if (rem <= u) continue;                           // (wave-uniform: no padded entry is loaded or multiplied)
                        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mya), u));
                        const dec_f32x2 a2 = {a, a};
                        const uint32_t odd = w[u] >> 4;
#pragma unroll
                        for (int p = 0; p < 4; ++p)
                            acc[p] = __builtin_elementwise_fma(a2, dec_f32x2{cvt_off_nibble(w[u], p), cvt_off_nibble(odd, p)}, acc[p]);
                    }
                }
        }
        float* out = d.recon + b * d.D + c * 8;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int col = c * 8 + f;
            if (live && col < d.D) {
                float r = d.step * acc[f >> 1][f & 1];          // rounded multiply, then rounded add (binary.py:38)
                r = r + (d.bias ? d.bias[col] : 0.0f);
                out[f] = r;
            }
        }
    }
}

// k winners of one row, 4-bit fields: the 13-instruction form where it is exact (4-bit values, no |a| near overflow), else the plain one
__device__ __forceinline__ void decode_row_sorted_wide4(const int* s_idx, const float* s_val, int k, const RowDecode& d,
                                                        long long b, int lane) {
    bool big = false;
    for (int j = lane; j < k; j += 64) big |= !(fabsf(s_val[j]) < 0x1p123f);      // (also true for NaN)
    if (d.n == 4 && !__any(big)) decode_row_sorted_wide_i4(s_idx, s_val, k, d, b, lane);
    else decode_row_sorted_wide<4>(s_idx, s_val, k, d, b, lane);
}

template <int U>
__device__ __forceinline__ void decode_row_sorted_any_wide(const int* s_idx, const float* s_val, int k, const RowDecode& d,
                                                           long long b, int lane) {
    if (d.table) { decode_row_table(s_idx, s_val, k, d, b, lane); return; }     // wave-uniform
    switch (d.fw) {                                  // wave-uniform
        case 1: decode_row_sorted<1, U>(s_idx, s_val, k, d, b, lane); break;
        case 2: decode_row_sorted<2, U>(s_idx, s_val, k, d, b, lane); break;
        case 4: decode_row_sorted_wide4(s_idx, s_val, k, d, b, lane); break;
        default: decode_row_sorted_wide<8>(s_idx, s_val, k, d, b, lane); break;
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

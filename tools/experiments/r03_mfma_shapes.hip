// Bare MFMA loops (operands in registers): FLOP/s of the 32x32 and 16x16 shapes of the fp16 / bf16 / fp32 matrix instructions at
// one and two waves per SIMD on every CU, and the accumulation order of v_mfma_f32_16x16x4_f32 against fmaf chains.
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/r03_mfma_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ void __launch_bounds__(256) loop_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 ah, bh;
    bf16x8 ab, bb;
    for (int e = 0; e < 8; ++e) {
        ah[e] = static_cast<_Float16>(0.001f * (lane + e)); bh[e] = static_cast<_Float16>(0.002f * (lane - e));
        ab[e] = static_cast<__bf16>(0.001f * (lane + e)); bb[e] = static_cast<__bf16>(0.002f * (lane - e));
    }
    const float af = 0.001f * lane, bf = 0.002f * lane;
    f32x16 c32[8];
    f32x4 c16[32];
    for (int t = 0; t < 8; ++t) for (int e = 0; e < 16; ++e) c32[t][e] = 0.f;
    for (int t = 0; t < 32; ++t) for (int e = 0; e < 4; ++e) c16[t][e] = 0.f;
    for (int i = 0; i < iters; ++i) {
        if (SHAPE == 0) { _Pragma("unroll") for (int t = 0; t < 8; ++t) c32[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c32[t], 0, 0, 0); }
        if (SHAPE == 1) { _Pragma("unroll") for (int t = 0; t < 32; ++t) c16[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c16[t], 0, 0, 0); }
        if (SHAPE == 2) { _Pragma("unroll") for (int t = 0; t < 8; ++t) c32[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c32[t], 0, 0, 0); }
        if (SHAPE == 3) { _Pragma("unroll") for (int t = 0; t < 32; ++t) c16[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c16[t], 0, 0, 0); }
        if (SHAPE == 4) { _Pragma("unroll") for (int t = 0; t < 8; ++t) c32[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, c32[t], 0, 0, 0); }
        if (SHAPE == 5) { _Pragma("unroll") for (int t = 0; t < 32; ++t) c16[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c16[t], 0, 0, 0); }
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int e = 0; e < 16; ++e) s += c32[t][e];
    for (int t = 0; t < 32; ++t) for (int e = 0; e < 4; ++e) s += c16[t][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// GEMM-like loops on RANDOM data with the A fragments re-read from LDS (conflict-free 16-byte slots per lane) and the B
// fragments stationary in registers, two waves per SIMD: the candidate sweep's structure (fp16: one W fragment read per
// 32x32x16 MFMA, or per two 16x16x32 MFMAs) and an fp32 GEMM's (one read per four 32x32x2 / 16x16x4 MFMAs).
template <int SHAPE>
__global__ void __launch_bounds__(512) lds_loop_kernel(const float* seed, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];           // 64 KiB of random bits
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = seed[(blockIdx.x * 16384 + i) & 0xFFFFF];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 xb[8];
    for (int j = 0; j < 8; ++j) xb[j] = *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + 256 * j + 64 * (threadIdx.x >> 6)) & 16380));
    f32x16 c32[2];
    f32x4 c16[8];
    for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) c32[t][e] = 0.f;
    for (int t = 0; t < 8; ++t) for (int e = 0; e < 4; ++e) c16[t][e] = 0.f;
    int off = lane * 4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(lds + ((off + 256 * j) & 16380));
            if (SHAPE == 0) {          // fp16 32x32x16: one fragment read per MFMA
                c32[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, xb[j]), c32[j & 1], 0, 0, 0);
            } else if (SHAPE == 1) {   // fp16 16x16x32: one fragment read per two MFMAs (two 16-row groups of the stationary operand)
                c16[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, xb[j]), c16[j & 3], 0, 0, 0);
                c16[4 + (j & 3)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, xb[j ^ 1]), c16[4 + (j & 3)], 0, 0, 0);
            } else if (SHAPE == 2) {   // fp32 32x32x2: four MFMAs per fragment read
#pragma unroll
                for (int t = 0; t < 4; ++t) c32[j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], xb[j][t], c32[j & 1], 0, 0, 0);
            } else {                   // fp32 16x16x4: eight MFMAs per fragment read (same FLOPs per read as above: 2 x half the tile)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    c16[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], xb[j][t], c16[j & 3], 0, 0, 0);
                    c16[4 + (j & 3)] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], xb[j ^ 1][t], c16[4 + (j & 3)], 0, 0, 0);
                }
            }
        }
        off += 2048;
    }
    float s = 0.f;
    for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) s += c32[t][e];
    for (int t = 0; t < 8; ++t) for (int e = 0; e < 4; ++e) s += c16[t][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE>
static double run_lds(double flop_per_iter_per_wave) {
    const int iters = 4000, blocks = 256;
    float *seed, *out;
    hipMalloc(&seed, (1 << 20) * 4);
    hipMalloc(&out, blocks * 512 * 4);
    std::vector<float> h(1 << 20);
    unsigned s = 99;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (static_cast<int>(s >> 9) % 2001 - 1000) * 1e-3f; }
    hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(lds_loop_kernel<SHAPE>, dim3(blocks), dim3(512), 0, 0, seed, out, 100);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(lds_loop_kernel<SHAPE>, dim3(blocks), dim3(512), 0, 0, seed, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipFree(seed); hipFree(out);
    return flop_per_iter_per_wave * iters * blocks * 8.0 / (ms * 1e-3) / 1e12;
}

// one wave: C[i][j] = sum_k A[i][k] B[k][j], K = 4, through v_mfma_f32_16x16x4_f32 (lane l: row / column l & 15, k = l >> 4)
__global__ void order_kernel(const float* A, const float* B, float* C, float c0) {
    const int lane = threadIdx.x;
    f32x4 acc = {c0, c0, c0, c0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(lane & 15) * 4 + (lane >> 4)], B[(lane >> 4) * 16 + (lane & 15)], acc, 0, 0, 0);
    for (int e = 0; e < 4; ++e) C[(4 * (lane >> 4) + e) * 16 + (lane & 15)] = acc[e];
}

template <int SHAPE>
static double run(int waves_per_simd, double flop_per_mfma, int mfma_per_iter) {
    const int iters = 20000, blocks = 256 * waves_per_simd;
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    hipLaunchKernelGGL(loop_kernel<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, 200);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(loop_kernel<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipFree(out);
    return flop_per_mfma * mfma_per_iter * iters * blocks * 4.0 / (ms * 1e-3) / 1e12;
}

int main() {
    const char* names[] = {"f16 32x32x16", "f16 16x16x32", "bf16 32x32x16", "bf16 16x16x32", "f32 32x32x2", "f32 16x16x4"};
    for (int w = 1; w <= 2; ++w) {
        double tf[6] = {run<0>(w, 32768, 8), run<1>(w, 16384, 32), run<2>(w, 32768, 8), run<3>(w, 16384, 32), run<4>(w, 4096, 8), run<5>(w, 2048, 32)};
        for (int i = 0; i < 6; ++i) printf("%d wave(s)/SIMD  %-14s %8.1f TFLOP/s\n", w, names[i], tf[i]);
    }
    printf("LDS-fed loops, random data, 2 waves/SIMD:  f16 32x32x16 %.0f TF | f16 16x16x32 %.0f TF | f32 32x32x2 %.1f TF | f32 16x16x4 %.1f TF\n",
           run_lds<0>(8 * 32768.0), run_lds<1>(16 * 16384.0), run_lds<2>(32 * 4096.0), run_lds<3>(64 * 2048.0));
    // accumulation order of the 16x16x4 fp32 instruction
    std::vector<float> A(64), B(64), C(256);
    unsigned seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (static_cast<int>(seed >> 8) % 20001 - 10000) * 1e-4f; };
    int match_up = 0, match_down = 0, match_fma_pairs = 0, total = 0;
    for (int trial = 0; trial < 200; ++trial) {
        for (auto& v : A) v = rnd() * std::ldexp(1.0f, static_cast<int>(seed % 24) - 12);
        for (auto& v : B) v = rnd() * std::ldexp(1.0f, static_cast<int>((seed >> 5) % 24) - 12);
        const float c0 = rnd();
        float *dA, *dB, *dC;
        hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024);
        hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(order_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, c0);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        hipFree(dA); hipFree(dB); hipFree(dC);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float up = c0, down = c0;
                for (int k = 0; k < 4; ++k) up = std::fmaf(A[i * 4 + k], B[k * 16 + j], up);
                for (int k = 3; k >= 0; --k) down = std::fmaf(A[i * 4 + k], B[k * 16 + j], down);
                const float p01 = std::fmaf(A[i * 4 + 1], B[16 + j], A[i * 4] * B[j]);
                const float p23 = std::fmaf(A[i * 4 + 3], B[48 + j], A[i * 4 + 2] * B[32 + j]);
                const float pairs = c0 + (p01 + p23);
                const float got = C[i * 16 + j];
                match_up += std::memcmp(&got, &up, 4) == 0;
                match_down += std::memcmp(&got, &down, 4) == 0;
                match_fma_pairs += std::memcmp(&got, &pairs, 4) == 0;
                ++total;
            }
    }
    printf("v_mfma_f32_16x16x4_f32 accumulation: %d results; equal to the ascending fmaf chain %d, descending %d, pairwise %d\n",
           total, match_up, match_down, match_fma_pairs);
    return 0;
}

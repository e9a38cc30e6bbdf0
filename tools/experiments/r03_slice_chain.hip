// Prototype: the refinement's exact chains regrouped by HIDDEN SLICE, so that a slice of W stays in an XCD's 4 MiB L2.
// Survivor lists per row (sorted by hidden index, so by slice) + per-slice offsets -> one wave task = (slice, block of 128
// rows): expand the block's entries of that slice into batches of 64 (row, entry) pairs, run the 64 exact fmaf chains with
// W rows AND activation rows fetched line-wise and transposed through LDS.  Timing only (synthetic uniform survivor sets).
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/r03_slice_chain.hip -o tools/experiments/r03_slice_chain.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef ABL
#define ABL 0
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kL = 72;          // list slots per row
constexpr int kStride = 36;     // floats per transposed-tile row
constexpr int kXRows = 40;      // distinct activation rows per batch held in the x tile (more -> the batch is cut short)
constexpr int kQueue = 640;     // (row, entry) pairs queued per wave task

// per wave: W tile [64][36] | x tile [kXRows][36] | queue [kQueue] u32 | xrow ids [64] int
constexpr int kLdsPerWave = 64 * kStride * 4 + kXRows * kStride * 4 + kQueue * 4 + 64 * 4;

template <int NXL>   // x line-loads per block (8 rows each)
__device__ __forceinline__ void run_batch(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                          const int* __restrict__ lists, float* __restrict__ vals, int D, int row0, uint32_t my,
                                          bool valid, int lane, float* wt, float* xt, int* xr, int R, int rx) {
    const int rl = my >> 8, ent = my & 255;
    const int b = row0 + rl;
    const int h = lists[static_cast<size_t>(b) * kL + ent];
    float acc = bias[h];
    uint32_t woff[8], xoff[NXL];
#pragma unroll
#if ABL == 3   // every batch gathers from the same 64 rows of W
    for (int i = 0; i < 8; ++i) woff[i] = static_cast<uint32_t>(8 * i + (lane >> 3)) * static_cast<uint32_t>(D * 4) + 16u * (lane & 7);
#else
    for (int i = 0; i < 8; ++i) woff[i] = static_cast<uint32_t>(__shfl(h, 8 * i + (lane >> 3), 64)) * static_cast<uint32_t>(D * 4) + 16u * (lane & 7);
#endif
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
        int tr = 8 * i + (lane >> 3);
        tr = tr < R ? tr : R - 1;
#if ABL == 5   // every batch reads its activation rows from the same 64 rows (what an x-resident form would save)
        xoff[i] = static_cast<uint32_t>(xr[tr] & 63) * static_cast<uint32_t>(D * 4) + 16u * (lane & 7);
#else
        xoff[i] = static_cast<uint32_t>(xr[tr]) * static_cast<uint32_t>(D * 4) + 16u * (lane & 7);
#endif
    }
    const char* wb = reinterpret_cast<const char*>(W);
    const char* xb = reinterpret_cast<const char*>(x);
    const int nblk = D / 32;
    f32x4 ws[2][8], xs[2][NXL];
    auto load = [&](int s, int blk) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ws[s][i] = *reinterpret_cast<const f32x4*>(wb + woff[i] + 128 * blk);
#pragma unroll
        for (int i = 0; i < NXL; ++i) xs[s][i] = *reinterpret_cast<const f32x4*>(xb + xoff[i] + 128 * blk);
    };
    auto consume = [&](int s) {
        f32x4 w[8], xv[8];
#if ABL == 2 || ABL == 4   // no W transposition
#pragma unroll
        for (int q = 0; q < 8; ++q) w[q] = ws[s][q];
#else
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(wt + (8 * i + (lane >> 3)) * kStride + 4 * (lane & 7)) = ws[s][i];
#endif
#if ABL == 1 || ABL == 4   // no activation tile
#pragma unroll
        for (int q = 0; q < 8; ++q) xv[q] = xs[s][q % NXL];
#else
#pragma unroll
        for (int i = 0; i < NXL; ++i)
            if (8 * i + (lane >> 3) < kXRows) *reinterpret_cast<f32x4*>(xt + (8 * i + (lane >> 3)) * kStride + 4 * (lane & 7)) = xs[s][i];
#endif
        asm volatile("" ::: "memory");
        const float* mw = wt + lane * kStride;
        const float* mx = xt + rx * kStride;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#if ABL != 2 && ABL != 4
            w[q] = *reinterpret_cast<const f32x4*>(mw + 4 * q);
#endif
#if ABL != 1 && ABL != 4
            xv[q] = *reinterpret_cast<const f32x4*>(mx + 4 * q);
#endif
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc = fmaf(xv[q][0], w[q][0], acc);
            acc = fmaf(xv[q][1], w[q][1], acc);
            acc = fmaf(xv[q][2], w[q][2], acc);
            acc = fmaf(xv[q][3], w[q][3], acc);
        }
    };
    load(0, 0);
    for (int t = 0; t < nblk; t += 2) {
        if (t + 1 < nblk) load(1, t + 1);
        consume(0);
        if (t + 2 < nblk) load(0, t + 2);
        if (t + 1 < nblk) consume(1);
    }
    if (valid) vals[static_cast<size_t>(b) * kL + ent] = acc;
}

__global__ void __launch_bounds__(256, 2)
slice_chain_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                   const int* __restrict__ lists, const uint8_t* __restrict__ offs /* [S + 1][B] */, float* __restrict__ vals,
                   int B, int D, int S, int* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* base = smem + wave * kLdsPerWave;
    float* wt = reinterpret_cast<float*>(base);
    float* xt = wt + 64 * kStride;
    uint32_t* queue = reinterpret_cast<uint32_t*>(xt + kXRows * kStride);
    int* xr = reinterpret_cast<int*>(queue + kQueue);
    const int g = blockIdx.x, xcd = g & 7, q = g >> 3;
    const int wgs_per_slice = B / 512;                       // 4 waves x 128 rows
    const int slice = xcd + 8 * (q / wgs_per_slice);
    if (slice >= S) return;
    const int row0 = ((q % wgs_per_slice) * 4 + wave) * 128;
    // ---- expand: entries of this slice, rows row0 .. row0 + 127, into the queue; as many rounds as the queue needs ----
    int a[2], left[2];
    for (int half = 0; half < 2; ++half) {
        const int r = row0 + 64 * half + lane;
        a[half] = offs[static_cast<size_t>(slice) * B + r];
        left[half] = offs[static_cast<size_t>(slice + 1) * B + r] - a[half];
    }
    int nb = 0, grand = 0;
    while (__any(left[0] > 0 || left[1] > 0)) {
        int total = 0;
        for (int half = 0; half < 2; ++half) {
            const int cnt = left[half];
            int incl = cnt;
            for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
            const int start = total + incl - cnt;
            int wrote = 0;
            for (int i = 0; i < cnt; ++i)
                if (start + i < kQueue) { queue[start + i] = (static_cast<uint32_t>(64 * half + lane) << 8) | static_cast<uint32_t>(a[half] + i); ++wrote; }
            a[half] += wrote;
            left[half] -= wrote;
            total += __shfl(incl, 63, 64);
        }
        total = total < kQueue ? total : kQueue;
        grand += total;
        asm volatile("" ::: "memory");
        // ---- batches of up to 64 pairs, cut short where the x tile would overflow ----
        int p0 = 0;
        while (p0 < total) {
            const int p = p0 + lane;
            const bool in = p < total;
            const uint32_t my = queue[in ? p : p0];
            const int rl = my >> 8;
            const int prev = __shfl_up(rl, 1, 64);
            const bool head = in && (lane == 0 || prev != rl);
            const unsigned long long hb = __ballot(head);
            int rx = __popcll(hb & ((2ull << lane) - 1ull)) - 1;
            // cut the batch at the first pair whose row would be tile row kXRows
            const unsigned long long over = __ballot(in && rx >= kXRows);
            const int take = over ? __builtin_ctzll(over) : (total - p0 < 64 ? total - p0 : 64);
            const bool valid = lane < take;
            const int R0 = __popcll(hb & (take >= 64 ? ~0ull : ((1ull << take) - 1ull)));
            if (head && valid) xr[rx] = row0 + rl;
            rx = valid ? rx : 0;
            asm volatile("" ::: "memory");
            if (R0 <= 8) run_batch<1>(x, W, bias, lists, vals, D, row0, my, valid, lane, wt, xt, xr, R0, rx);
            else if (R0 <= 16) run_batch<2>(x, W, bias, lists, vals, D, row0, my, valid, lane, wt, xt, xr, R0, rx);
            else if (R0 <= 32) run_batch<4>(x, W, bias, lists, vals, D, row0, my, valid, lane, wt, xt, xr, R0, rx);
            else run_batch<5>(x, W, bias, lists, vals, D, row0, my, valid, lane, wt, xt, xr, R0, rx);
            asm volatile("" ::: "memory");
            p0 += take;
            ++nb;
        }
    }
    const int total = grand;
    if (stats && lane == 0) { atomicAdd(&stats[0], nb); atomicAdd(&stats[1], total); }
}

// reference: the row-major form (one wave per row, W rows gathered from wherever they are), chains only
__global__ void __launch_bounds__(256, 3)
row_chain_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                 const int* __restrict__ lists, const uint8_t* __restrict__ offs, float* __restrict__ vals, int B, int D, int S) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* wt = reinterpret_cast<float*>(smem + wave * 64 * kStride * 4);
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const int m = offs[static_cast<size_t>(S) * B + b];
    typedef const __attribute__((address_space(4))) f32x4* cvec_t;
    cvec_t xrow = (cvec_t)(x + static_cast<size_t>(b) * D);
    for (int j0 = 0; j0 < m; j0 += 64) {
        const int j = j0 + lane;
        const int h = lists[static_cast<size_t>(b) * kL + (j < m ? j : j0)];
        float acc = bias[h];
        uint32_t woff[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) woff[i] = static_cast<uint32_t>(__shfl(h, 8 * i + (lane >> 3), 64)) * static_cast<uint32_t>(D * 4) + 16u * (lane & 7);
        const char* wb = reinterpret_cast<const char*>(W);
        f32x4 ws[3][8];
        auto load = [&](int s, int blk) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ws[s][i] = *reinterpret_cast<const f32x4*>(wb + woff[i] + 128 * blk);
        };
        auto consume = [&](int s, int t) {
            f32x4 xv[8], w[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xv[q] = xrow[8 * t + q];
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(wt + (8 * i + (lane >> 3)) * kStride + 4 * (lane & 7)) = ws[s][i];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = *reinterpret_cast<const f32x4*>(wt + lane * kStride + 4 * q);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc = fmaf(xv[q][0], w[q][0], acc);
                acc = fmaf(xv[q][1], w[q][1], acc);
                acc = fmaf(xv[q][2], w[q][2], acc);
                acc = fmaf(xv[q][3], w[q][3], acc);
            }
        };
        const int nblk = D / 32;
        load(0, 0); load(1, 1);
        for (int t = 0; t < nblk; t += 3) {
            if (t + 2 < nblk) load(2, t + 2);
            consume(0, t);
            if (t + 3 < nblk) load(0, t + 3);
            if (t + 1 < nblk) consume(1, t + 1);
            if (t + 4 < nblk) load(1, t + 4);
            if (t + 2 < nblk) consume(2, t + 2);
        }
        if (j < m) vals[static_cast<size_t>(b) * kL + j] = acc;
    }
}

int main(int argc, char** argv) {
    const int B = 65536, D = 512, H = 32768;
    const int M = argc > 1 ? atoi(argv[1]) : 69;       // survivors per row
    const int extra_lds = argc > 2 ? atoi(argv[2]) : 0;   // bytes of unused LDS per workgroup (occupancy experiments)
    std::mt19937 rng(7);
    std::vector<float> hx(static_cast<size_t>(B) * D), hW(static_cast<size_t>(H) * D), hb(H);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : hx) v = nd(rng);
    for (auto& v : hW) v = 0.05f * nd(rng);
    for (auto& v : hb) v = 0.1f * nd(rng);
    float *dx, *dW, *db, *dv, *dv2;
    int *dl, *dstats;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMalloc(&dv, static_cast<size_t>(B) * kL * 4)); CK(hipMalloc(&dv2, static_cast<size_t>(B) * kL * 4));
    CK(hipMalloc(&dl, static_cast<size_t>(B) * kL * 4)); CK(hipMalloc(&dstats, 8));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    std::vector<int> lists(static_cast<size_t>(B) * kL, 0);
    for (int b = 0; b < B; ++b) {
        int* l = &lists[static_cast<size_t>(b) * kL];
        int n = 0;
        while (n < M) {
            const int h = rng() % H;
            bool dup = false;
            for (int i = 0; i < n; ++i) dup |= l[i] == h;
            if (!dup) l[n++] = h;
        }
        std::sort(l, l + M);
    }
    CK(hipMemcpy(dl, lists.data(), lists.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(slice_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kLdsPerWave + extra_lds));
    for (int S : {16}) {
        std::vector<uint8_t> offs(static_cast<size_t>(S + 1) * B);
        const int per = H / S;
        for (int b = 0; b < B; ++b) {
            const int* l = &lists[static_cast<size_t>(b) * kL];
            int j = 0;
            for (int s = 0; s <= S; ++s) {
                while (j < M && l[j] < s * per) ++j;
                offs[static_cast<size_t>(s) * B + b] = static_cast<uint8_t>(s == S ? M : j);
            }
        }
        uint8_t* doffs;
        CK(hipMalloc(&doffs, offs.size()));
        CK(hipMemcpy(doffs, offs.data(), offs.size(), hipMemcpyHostToDevice));
        const int grid = 8 * (S / 8) * (B / 512);
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemset(dstats, 0, 8));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(slice_chain_kernel, dim3(grid), dim3(256), 4 * kLdsPerWave + extra_lds, 0, dx, dW, db, dl, doffs, dv, B, D, S, dstats);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        int st[2];
        CK(hipMemcpy(st, dstats, 8, hipMemcpyDeviceToHost));
        printf("slice-major  S = %2d  M = %d: %.3f ms   (%d batches, %.1f pairs per batch)\n", S, M, best, st[0], double(st[1]) / st[0]);
        if (S == 16) {
            float rbest = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(row_chain_kernel, dim3(B / 4), dim3(256), 4 * 64 * kStride * 4, 0, dx, dW, db, dl, doffs, dv2, B, D, S);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                rbest = ms < rbest ? ms : rbest;
            }
            printf("row-major (one wave per row, chains only)  M = %d: %.3f ms\n", M, rbest);
            std::vector<float> v1(static_cast<size_t>(B) * kL), v2(v1.size());
            CK(hipMemcpy(v1.data(), dv, v1.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(v2.data(), dv2, v2.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (int b = 0; b < B; ++b)
                for (int j = 0; j < M; ++j) bad += v1[static_cast<size_t>(b) * kL + j] != v2[static_cast<size_t>(b) * kL + j];
            // spot check against a host chain
            int hbad = 0;
            for (int b = 0; b < B; b += 4099)
                for (int j = 0; j < M; j += 7) {
                    const int h = lists[static_cast<size_t>(b) * kL + j];
                    float acc = hb[h];
                    for (int k = 0; k < D; ++k) acc = fmaf(hx[static_cast<size_t>(b) * D + k], hW[static_cast<size_t>(h) * D + k], acc);
                    hbad += acc != v1[static_cast<size_t>(b) * kL + j];
                }
            printf("values: %zu differ between the two forms, %d of the host spot checks differ\n", bad, hbad);
        }
        CK(hipFree(doffs));
    }
    return 0;
}

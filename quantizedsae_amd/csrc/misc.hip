// misc.hip -- library plumbing, dense-latent materialisation and the recon-MSE reduction.
#include <map>

#include "common.h"

namespace qsae {

char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

// Helper objects of this host thread, one set per device it has used (common.h).  They live as long as the thread;
// the handful of streams / events / pinned words is not reclaimed at thread exit (the runtime may already be gone).
int thread_device_ctx(ThreadDeviceCtx** out) {
    static thread_local std::map<int, ThreadDeviceCtx> table;
    int dev = -1;
    QSAE_HIP(hipGetDevice(&dev));
    auto it = table.find(dev);
    if (it == table.end()) {
        // built in a local object and entered into the table only when all five parts exist: a failure half-way (event
        // creation, page-locked memory under pressure) leaves no entry behind that later calls would take for complete
        ThreadDeviceCtx c{};
        auto undo = [&c]() {
            if (c.pinned) (void)hipHostFree(c.pinned);
            if (c.ev_copied) (void)hipEventDestroy(c.ev_copied);
            if (c.ev_join) (void)hipEventDestroy(c.ev_join);
            if (c.ev_fork) (void)hipEventDestroy(c.ev_fork);
            if (c.side) (void)hipStreamDestroy(c.side);
        };
        hipError_t e = hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c.ev_copied, hipEventDisableTiming);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c.pinned), sizeof(int), hipHostMallocDefault);
        if (e != hipSuccess) {
            undo();
            snprintf(last_error_buf(), 512, "%s: creating the per-thread helper objects failed: %s", __func__, hipGetErrorString(e));
            return QSAE_ERR_HIP;
        }
        it = table.emplace(dev, c).first;
    }
    *out = &it->second;
    return QSAE_OK;
}

static thread_local SweepProfile t_sweep_profile;

SweepProfile take_sweep_profile() {
    const SweepProfile p = t_sweep_profile;
    t_sweep_profile = SweepProfile{};
    return p;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// dense[b][idx[b][j]] = val[b][j]  (after the rows were zeroed): reference zeros_like + scatter_
// (sae/baseline.py:38-40) and latent*mask (sae/binary.py:96-99).
__global__ void __launch_bounds__(256)
scatter_rows_kernel(const int32_t* __restrict__ idx, const float* __restrict__ val, long long total, int k,
                    int H, float* __restrict__ dense, int64_t ld) {
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const long long b = gid / k;
    const int h = idx[gid];
    if (h >= 0 && h < H) dense[b * ld + h] = val[gid];
}

// sum of (float)((r - x)^2) in double: per-thread fp32 square, double accumulation, one atomic per
// wave (scripts/analysis/dynamic_analysis.py:86-100 accumulates a fp32 per-batch sum into a
// Python float; double accumulation is at least as accurate).
__global__ void __launch_bounds__(256)
sq_err_kernel(const float* __restrict__ r, const float* __restrict__ x, size_t n, double* __restrict__ sum) {
    const size_t n4 = n / 4;
    double acc = 0.0;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    // a workgroup walks 16-KiB tiles (4 x 256 chunks of 16 bytes): four independent loads per operand in flight
    // per thread, all inside one contiguous tile
    const size_t tiles = n4 / 1024;
    double acc1 = 0.0;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t i0 = t * 1024 + threadIdx.x;
        f32x4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = reinterpret_cast<const f32x4*>(r)[i0 + 256 * u];
            b[u] = reinterpret_cast<const f32x4*>(x)[i0 + 256 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = a[u][j] - b[u][j];
                if (u & 1) acc1 += static_cast<double>(d * d);
                else acc += static_cast<double>(d * d);
            }
    }
    for (size_t i = tiles * 1024 + static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 a = reinterpret_cast<const f32x4*>(r)[i];
        const f32x4 b = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = a[j] - b[j];
            acc += static_cast<double>(d * d);
        }
    }
    acc += acc1;
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        const float d = r[i] - x[i];
        acc += static_cast<double>(d * d);
    }
    // one atomic per workgroup: tens of thousands of fp64 atomics on one address serialise (~12 ns each)
    __shared__ double wave_sum[4];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
        if (t != 0.0) atomicAdd(sum, t);
    }
}

// dst[r][8g + j/2 + 4*(j&1)] = src[r][8g + j]: every 8 consecutive k stored as [k0 k2 k4 k6 k1 k3 k5 k7],
// the order in which v_mfma_f32_32x32x2_f32 wants them in LDS (gemm_mfma_f32.h).  One thread per group.
__global__ void __launch_bounds__(256)
kperm_rows_kernel(const float* __restrict__ src, long long groups, float* __restrict__ dst) {
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= groups) return;
    const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * gid];
    const f32x4 b = reinterpret_cast<const f32x4*>(src)[2 * gid + 1];
    reinterpret_cast<f32x4*>(dst)[2 * gid] = f32x4{a[0], a[2], b[0], b[2]};
    reinterpret_cast<f32x4*>(dst)[2 * gid + 1] = f32x4{a[1], a[3], b[1], b[3]};
}

// one wave per 64 words: lane l builds word (w0 + l) of a row from 32 consecutive floats
__global__ void __launch_bounds__(256)
pack_bits_gt_kernel(const float* __restrict__ dense, int64_t ld, int B, int H, float thr,
                    uint32_t* __restrict__ zbits, int64_t words_ld, int words) {
    const long long gid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= static_cast<long long>(B) * words) return;
    const long long b = gid / words;
    const int wi = static_cast<int>(gid % words);
    const float* p = dense + b * ld + static_cast<long long>(wi) * 32;
    uint32_t word = 0;
    for (int j = 0; j < 32; ++j) {
        const int h = wi * 32 + j;
        if (h < H && p[j] > thr) word |= 1u << j;
    }
    zbits[b * words_ld + wi] = word;
}

// ---- small elementwise steps of the forward paths (one pass over HBM each; separately rounded operations, as the
// reference's ATen sequence rounds them) ----------------------------------------------------------------------
// mode 0: out = (a - b) * scale           residual = (residual - reconstruction) * 2, sae/residual_quantized.py:67
// mode 1: out = a >= cutoff ? 1 : 0       binary latent of BinaryLatentSAE (sigmoid(pre) >= 0.5), sae/binary_latent.py:21-24
// mode 2: out = scale * a + bias[i % D]   reconstruction = step * latent.matmul(int_w) + bias, sae/binary.py:38
__global__ void __launch_bounds__(256)
elementwise_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, int mode, float scale, int D,
                   float* __restrict__ out) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v;
        if (mode == 0) {
            v = a[i] - b[i];
            v = v * scale;
        } else if (mode == 1) {
            v = a[i] >= scale ? 1.0f : 0.0f;
        } else {
            v = scale * a[i];
            v = v + (b ? b[i % static_cast<size_t>(D)] : 0.0f);
        }
        out[i] = v;
    }
}

static int launch_elementwise(const float* a, const float* b, size_t n, int mode, float scale, int D, float* out,
                              hipStream_t s) {
    if (n == 0) return QSAE_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(elementwise_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, a, b, n, mode, scale, D, out);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

// scatter without the preceding zero fill (the fused encoder zero-fills the dense latent itself)
int scatter_rows(const int32_t* idx, const float* val, int B, int k, int H, float* dense, int64_t ld, hipStream_t s);

// zeros + the k survivors of every row.  Measured on the [65536, 32768] latent: the runtime's linear fill
// (6.9 TB/s) + the scatter kernel take 1.63 ms; a fused kernel that writes whole rows per workgroup (zeros,
// vmcnt(0) + barrier, survivors) took 2.1 ms -- 2048 concurrent rows 128 KiB apart lose the DRAM page
// locality of a linear sweep.
int densify_rows(const int32_t* idx, const float* val, int B, int k, int H, float* dense, int64_t ld, hipStream_t s) {
    if (B == 0) return QSAE_OK;
    if (ld == H) {
        QSAE_HIP(hipMemsetAsync(dense, 0, static_cast<size_t>(B) * H * sizeof(float), s));
    } else {
        QSAE_HIP(hipMemset2DAsync(dense, static_cast<size_t>(ld) * sizeof(float), 0,
                                  static_cast<size_t>(H) * sizeof(float), B, s));
    }
    return scatter_rows(idx, val, B, k, H, dense, ld, s);
}

int scatter_rows(const int32_t* idx, const float* val, int B, int k, int H, float* dense, int64_t ld, hipStream_t s) {
    const long long total = static_cast<long long>(B) * k;
    if (total == 0) return QSAE_OK;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, idx,
                       val, total, k, H, dense, ld);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

}  // namespace qsae

using namespace qsae;

extern "C" int qsae_kperm_rows(const float* src, int rows, int K, float* dst, qsae_stream_t stream) {
    QSAE_CHECK_ARG(rows >= 0 && K > 0, "rows >= 0 and K > 0 required");
    if (rows == 0) return QSAE_OK;
    QSAE_CHECK_ARG(src && dst && src != dst, "null or aliasing pointers");
    QSAE_CHECK_SUPPORTED(K % 8 == 0, "K must be a multiple of 8");
    QSAE_CHECK_ARG(aligned16(src) && aligned16(dst), "src and dst must be 16-byte aligned");
    const long long groups = static_cast<long long>(rows) * (K / 8);
    hipLaunchKernelGGL(kperm_rows_kernel, dim3(static_cast<unsigned>((groups + 255) / 256)), dim3(256), 0,
                       as_stream(stream), src, groups, dst);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

extern "C" int qsae_pack_bits_gt(const float* dense, int64_t ld, int B, int H, float thr, uint32_t* zbits,
                                 int64_t words_ld, qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && H > 0, "B >= 0 and H > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(dense && zbits, "null pointer");
    const int words = (H + 31) / 32;
    QSAE_CHECK_ARG(ld >= H && words_ld >= words, "leading dimension too small");
    hipStream_t s = as_stream(stream);
    if (words_ld > words)
        QSAE_HIP(hipMemset2DAsync(zbits + words, words_ld * 4, 0, (words_ld - words) * 4, B, s));
    const long long total = static_cast<long long>(B) * words;
    hipLaunchKernelGGL(pack_bits_gt_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, dense,
                       ld, B, H, thr, zbits, words_ld, words);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

extern "C" int qsae_residual_update(const float* residual, const float* recon, size_t n, float scale, float* out,
                                    qsae_stream_t stream) {
    if (n == 0) return QSAE_OK;
    QSAE_CHECK_ARG(residual && recon && out, "null pointer");
    return launch_elementwise(residual, recon, n, 0, scale, 1, out, as_stream(stream));
}

extern "C" int qsae_threshold_ge(const float* pre, size_t n, float cutoff, float* out, qsae_stream_t stream) {
    if (n == 0) return QSAE_OK;
    QSAE_CHECK_ARG(pre && out, "null pointer");
    return launch_elementwise(pre, nullptr, n, 1, cutoff, 1, out, as_stream(stream));
}

extern "C" int qsae_scale_bias_rows(const float* acc, int B, int D, float scale, const float* bias, float* out,
                                    qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0, "B >= 0 and D > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(acc && out, "null pointer");
    return launch_elementwise(acc, bias, static_cast<size_t>(B) * D, 2, scale, D, out, as_stream(stream));
}

extern "C" int qsae_abi_version(void) { return QSAE_ABI_VERSION; }

extern "C" int qsae_profile_sweep_events(void* ev_begin, void* ev_end) {
    t_sweep_profile.begin = static_cast<hipEvent_t>(ev_begin);
    t_sweep_profile.end = static_cast<hipEvent_t>(ev_end);
    return QSAE_OK;
}

// Timing events created and read through THIS library's HIP runtime (the one that records them): a caller that dlopens
// a HIP runtime of its own may get a second copy whose events the recording side does not know.
extern "C" int qsae_profile_event_create(void** ev) {
    QSAE_CHECK_ARG(ev != nullptr, "null pointer");
    hipEvent_t e = nullptr;
    QSAE_HIP(hipEventCreate(&e));
    *ev = e;
    return QSAE_OK;
}

extern "C" int qsae_profile_event_destroy(void* ev) {
    if (ev) QSAE_HIP(hipEventDestroy(static_cast<hipEvent_t>(ev)));
    return QSAE_OK;
}

extern "C" int qsae_profile_event_elapsed_ms(void* ev_begin, void* ev_end, float* ms) {
    QSAE_CHECK_ARG(ev_begin && ev_end && ms, "null pointer");
    QSAE_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(ev_begin), static_cast<hipEvent_t>(ev_end)));
    return QSAE_OK;
}

extern "C" const char* qsae_last_error(void) { return last_error_buf(); }

extern "C" int qsae_device_info(int* cu_count, char* arch, int arch_len) {
    int dev = 0;
    QSAE_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    QSAE_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, static_cast<size_t>(arch_len) - 1);
        arch[arch_len - 1] = 0;
    }
    return QSAE_OK;
}

extern "C" int qsae_densify(const int32_t* idx, const float* val, int B, int k, int H, float* dense, int64_t ld,
                            qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && H > 0 && k >= 1, "B >= 0, H > 0, k >= 1 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(idx && val && dense, "null pointer");
    QSAE_CHECK_ARG(ld >= H, "ld < H");
    return densify_rows(idx, val, B, k, H, dense, ld, as_stream(stream));
}



extern "C" int qsae_sq_err_sum(const float* recon, const float* x, size_t n, double* sum, qsae_stream_t stream) {
    if (n == 0) return QSAE_OK;
    QSAE_CHECK_ARG(recon && x && sum, "null pointer");
    QSAE_CHECK_ARG(aligned16(recon) && aligned16(x), "recon and x must be 16-byte aligned");
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(sq_err_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream), recon, x,
                       n, sum);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}

#ifdef QSAE_DEBUG_BUILD
// experiment support (debug library only): which XCD / CU a stream's workgroups land on -- out[2 i] = XCC_ID register,
// out[2 i + 1] = HW_ID register of workgroup i; every workgroup spins ~20 us so that the grid spreads over the CUs
namespace qsae {
__global__ void cu_probe_kernel(unsigned* out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000ull) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
}
}  // namespace qsae
extern "C" int qsae_debug_cu_probe(unsigned* out, int workgroups, qsae_stream_t stream) {
    using namespace qsae;
    hipLaunchKernelGGL(cu_probe_kernel, dim3(workgroups), dim3(1024), 0, as_stream(stream), out);
    QSAE_LAUNCH_CHECK();
    return QSAE_OK;
}
#endif

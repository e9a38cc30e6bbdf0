// common.h -- shared helpers for the gfx950 quantized-SAE kernels (host + device).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/qsae.h"

namespace qsae {

// ---- error plumbing -----------------------------------------------------------------
char* last_error_buf();   // thread-local, 512 bytes (defined in api_misc.hip)

inline int fail(int code, const char* fmt, const char* a = "", long long b = 0, long long c = 0) {
    snprintf(last_error_buf(), 512, fmt, a, b, c);
    return code;
}

#define QSAE_CHECK_ARG(cond, what)                                                         \
    do {                                                                                   \
        if (!(cond)) return ::qsae::fail(QSAE_ERR_INVALID_ARG, "%s: invalid argument: " what, __func__); \
    } while (0)

#define QSAE_CHECK_SUPPORTED(cond, what)                                                   \
    do {                                                                                   \
        if (!(cond)) return ::qsae::fail(QSAE_ERR_UNSUPPORTED, "%s: unsupported: " what, __func__); \
    } while (0)

#define QSAE_HIP(call)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            snprintf(::qsae::last_error_buf(), 512, "%s: %s failed: %s", __func__, #call,  \
                     hipGetErrorString(e_));                                               \
            return QSAE_ERR_HIP;                                                           \
        }                                                                                  \
    } while (0)

#define QSAE_LAUNCH_CHECK() QSAE_HIP(hipGetLastError())

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline hipStream_t as_stream(qsae_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- host-side state ------------------------------------------------------------------
// The library keeps no state that one call hands to another.  What it does keep is (a) "this kernel's dynamic-LDS
// limit has been raised on device d" -- a per-device fact (hipFuncSetAttribute acts on the current device's copy of
// the code object), set once per device and instantiation -- and (b) helper objects (side stream, events, a pinned
// word) owned by one host thread for one device (ThreadDeviceCtx below), so that two host threads, or one thread
// walking over several devices, never share them.  Callers make the device of their pointers current, as for any HIP
// library; the current device is what keys both tables.
constexpr int kMaxDevices = 64;

struct PerDeviceOnce {
    std::atomic<unsigned char> done[kMaxDevices];
};

// Runs f() (a HIP attribute call returning hipError_t) once per device; two threads racing on the same device both
// run it, which is harmless (the call is idempotent).
template <class F>
inline hipError_t once_per_device(PerDeviceOnce& o, F&& f) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool tracked = dev >= 0 && dev < kMaxDevices;
    if (tracked && o.done[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = f();
    if (e == hipSuccess && tracked) o.done[dev].store(1, std::memory_order_release);
    return e;
}

#define QSAE_SET_MAX_LDS_ONCE(kern, bytes)                                                        \
    do {                                                                                          \
        static ::qsae::PerDeviceOnce once_;                                                       \
        QSAE_HIP(::qsae::once_per_device(once_, [&]() {                                           \
            return hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                       \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)); \
        }));                                                                                      \
    } while (0)

// Helper objects of the calling host thread on the current device (created on first use, never shared):
//   side, ev_fork, ev_join : the stream the co-resident zero-fill kernel runs on, forked from / joined into the caller's
//   pinned, ev_copied      : page-locked word + event for the one 4-byte read-back of the blocking entry points
//   prof_begin / prof_end  : one-shot profiling events (qsae_profile_sweep_events)
struct ThreadDeviceCtx {
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_copied = nullptr;
    int* pinned = nullptr;
};
int thread_device_ctx(ThreadDeviceCtx** out);     // misc.hip; QSAE_OK or an error code

struct SweepProfile {
    hipEvent_t begin = nullptr, end = nullptr;
};
SweepProfile take_sweep_profile();                // misc.hip: the calling thread's one-shot event pair (then cleared)

// ---- device helpers -----------------------------------------------------------------
// fp32 cutoffs of the reference's sigmoid (see oracle/qsae_oracle.c header):
//   sigmoid(w) >  0.5  <=>  w >= 0x33C00001 ;  sigmoid(w) >= 0.5  <=>  w >= 0xB43FFFFE
#define QSAE_SIG_GT_BITS 0x33C00001u
#define QSAE_SIG_GE_BITS 0xB43FFFFEu

__device__ __forceinline__ bool sig_gt_half(float w) { return w >= __uint_as_float(QSAE_SIG_GT_BITS); }
__device__ __forceinline__ bool sig_ge_half(float w) { return w >= __uint_as_float(QSAE_SIG_GE_BITS); }

// Monotone map float -> uint32 (larger float -> larger key); NaN above +inf; -0 == +0.
__device__ __forceinline__ uint32_t mono_key(float v) {
    uint32_t u = __float_as_uint(v);
    if (v != v) return 0xFFFFFFFFu;
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// Total order of the top-k: larger key wins; ties in value go to the smaller index.
__device__ __forceinline__ unsigned long long full_key(float v, uint32_t idx) {
    return (static_cast<unsigned long long>(mono_key(v)) << 32) | static_cast<uint32_t>(~idx);
}
__device__ __forceinline__ uint32_t key_index(unsigned long long key) {
    return ~static_cast<uint32_t>(key & 0xFFFFFFFFull);
}

// Signed bit-field extract (v_bfe_i32): `width` bits at `offset`, sign-extended.  The clang
// builtin is typed unsigned; the cast restores the signed result.
__device__ __forceinline__ int sbfe_i32(int v, int offset, int width) {
    return static_cast<int>(__builtin_amdgcn_sbfe(static_cast<unsigned>(v), static_cast<unsigned>(offset),
                                                  static_cast<unsigned>(width)));
}

__host__ __device__ inline int field_width(int n_bits) {
    return n_bits <= 1 ? 1 : n_bits <= 2 ? 2 : n_bits <= 4 ? 4 : 8;
}

}  // namespace qsae

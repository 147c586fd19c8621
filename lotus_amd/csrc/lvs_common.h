// Shared device/host helpers for liblotus_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lotus_hip.h"

typedef unsigned long long u64;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- result keys -------------------------------------------------------------------------------------------
// key = ord32(score where larger is better) << 32 | (0xFFFFFFFF - id); descending key order is
// (score best-first, id ascending) - the total order the oracle uses (oracle/flat.py pack_keys).
__host__ __device__ inline uint32_t lvs_ord32(float f) {
    f = f + 0.0f;  // fold -0.0 onto +0.0 so equal floats get equal keys
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u >> 31) ? ~u : (u ^ 0x80000000u);
}
__host__ __device__ inline float lvs_unord32(uint32_t o) {
    uint32_t u = (o >> 31) ? (o ^ 0x80000000u) : ~o;
    return __builtin_bit_cast(float, u);
}
__host__ __device__ inline u64 lvs_pack_key(float better, uint32_t id) {
    return ((u64)lvs_ord32(better) << 32) | (u64)(0xFFFFFFFFu - id);
}

// ---- wave64 helpers ----------------------------------------------------------------------------------------
__device__ inline u64 lvs_shfl_xor_u64(u64 v, int mask) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return ((u64)hi << 32) | lo;
}
__device__ inline u64 lvs_shfl_u64(u64 v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((u64)hi << 32) | lo;
}
// Bitonic sort of one u64 per lane across the 64 lanes of a wave, descending in lane order.
__device__ inline u64 lvs_wave_sort_desc(u64 v, int lane) {
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            u64 o = lvs_shfl_xor_u64(v, stride);
            bool up = (lane & size) == 0;  // size == 64: always true -> whole wave descending
            bool lower = (lane & stride) == 0;
            u64 mx = v > o ? v : o, mn = v > o ? o : v;
            v = (lower == up) ? mx : mn;
        }
    }
    return v;
}
// `v` is bitonic across the wave; finish into descending order.
__device__ inline u64 lvs_wave_bitonic_merge_desc(u64 v, int lane) {
#pragma unroll
    for (int stride = 32; stride > 0; stride >>= 1) {
        u64 o = lvs_shfl_xor_u64(v, stride);
        bool lower = (lane & stride) == 0;
        u64 mx = v > o ? v : o, mn = v > o ? o : v;
        v = lower ? mx : mn;
    }
    return v;
}

// ---- host-side error plumbing ------------------------------------------------------------------------------
void lvs_set_error(const char* fmt, ...);
#define LVS_HIP_CHECK(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            lvs_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return LVS_EDEVICE;                                                                  \
        }                                                                                        \
    } while (0)
#define LVS_REQUIRE(cond, ...)      \
    do {                            \
        if (!(cond)) {              \
            lvs_set_error(__VA_ARGS__); \
            return LVS_EINVAL;      \
        }                           \
    } while (0)

static inline int64_t lvs_round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
static inline int64_t lvs_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- tuning / debug knobs ----------------------------------------------------------------------------------
// The shipped library reads NO environment variable: every knob below is a compile-time constant unless the library is
// built with -DLVS_TUNING (`make tuning` -> liblotus_hip_tuning.so, reported by lvs_build_flags()).  Some knobs skip
// work and produce wrong results on purpose (timing ablations); they exist in the tuning build only.
#ifdef LVS_TUNING
#include <stdlib.h>
static inline long long lvs_tune(const char* name, long long dflt) {
    const char* e = getenv(name);
    return e ? atoll(e) : dflt;
}
static inline bool lvs_tune_set(const char* name) { return getenv(name) != nullptr; }
#else
#define lvs_tune(name, dflt) ((long long)(dflt))
#define lvs_tune_set(name) (false)
#endif

// Entry points launch on `stream`; make that stream's device current for the duration of the call (the caller may
// drive several GPUs from one process) and restore the previous one afterwards.
struct LvsDeviceGuard {
    int prev = -1;
    int dev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit LvsDeviceGuard(hipStream_t st) {
        err = hipGetDevice(&prev);
        dev = prev;
        if (err != hipSuccess || st == nullptr) return;
        hipDevice_t sd = 0;
        if (hipStreamGetDevice(st, &sd) != hipSuccess) {
            (void)hipGetLastError();
            return;  // legacy/default-stream handles: keep the current device
        }
        dev = (int)sd;
        if (dev != prev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~LvsDeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define LVS_DEVICE_GUARD(stream)                  \
    LvsDeviceGuard _lvs_guard((hipStream_t)(stream)); \
    LVS_HIP_CHECK(_lvs_guard.err)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember which devices have it (bit per device id).
#include <atomic>
struct LvsPerDeviceOnce {
    std::atomic<unsigned long long> mask{0};
    std::atomic<unsigned long long> bytes[64];
    bool done(int dev, size_t need) const {
        return dev >= 0 && dev < 64 && (mask.load(std::memory_order_acquire) >> dev & 1ull) &&
               bytes[dev].load(std::memory_order_relaxed) >= need;
    }
    void set(int dev, size_t have) {
        if (dev < 0 || dev >= 64) return;
        bytes[dev].store(have, std::memory_order_relaxed);
        mask.fetch_or(1ull << dev, std::memory_order_release);
    }
};

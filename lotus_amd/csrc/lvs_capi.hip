// C-ABI of liblotus_hip (see include/lotus_hip.h) + the small streaming kernels around the tile kernel:
// row packing (fp32/fp16 -> fp16 or fp16 hi|lo), row gather, candidate-list merge, key decoding.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "lvs_common.h"
#include "lvs_tile.h"

// ---------------------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void lvs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* lvs_last_error(void) { return g_err; }
extern "C" int32_t lvs_abi_version(void) { return LVS_ABI_VERSION; }
extern "C" int32_t lvs_build_flags(void) {
    int32_t f = 0;
#ifdef LVS_TUNING
    f |= LVS_BUILD_TUNING;
#endif
#ifdef LVS_COUNT_EVENTS
    f |= LVS_BUILD_COUNT_EVENTS;
#endif
    return f;
}

extern "C" int32_t lvs_device_count(int32_t* out_count) {
    LVS_REQUIRE(out_count, "out_count is NULL");
    int n = 0;
    LVS_HIP_CHECK(hipGetDeviceCount(&n));
    *out_count = n;
    return LVS_OK;
}

extern "C" int32_t lvs_device_info(int32_t device, char* name, int32_t name_cap, int32_t* out_cus,
                                   int64_t* out_hbm_bytes) {
    hipDeviceProp_t p;
    LVS_HIP_CHECK(hipGetDeviceProperties(&p, device));
    if (name && name_cap > 0) {
        snprintf(name, (size_t)name_cap, "%s (%s)", p.name, p.gcnArchName);
    }
    if (out_cus) *out_cus = p.multiProcessorCount;
    if (out_hbm_bytes) *out_hbm_bytes = (int64_t)p.totalGlobalMem;
    return LVS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// timing hook (HIP events on the launch stream around the dominant kernel)
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct TimingState {
    std::mutex mu;
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0;
    int64_t launches = 0;
    int64_t calls = 0;   // entry-point calls that timed at least one launch (a chunked search times several per call)
    int32_t kernel = 0;  // LVS_KERNEL_* of the last timed launch
} g_timing;


struct ScopedKernelTimer {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t s;
    bool on;
    // continuation: a further launch of the SAME search (its next chunk of queries) - counted as a launch, not as a call
    explicit ScopedKernelTimer(hipStream_t st, int32_t kernel = LVS_KERNEL_TILE, bool continuation = false) : s(st) {
        std::lock_guard<std::mutex> lk(g_timing.mu);
        on = g_timing.on;
        if (on) {
            g_timing.kernel = kernel;
            if (!continuation) ++g_timing.calls;
            if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
                on = false;
                return;
            }
            (void)hipEventRecord(e0, s);
        }
    }
    ~ScopedKernelTimer() {
        if (!on) return;
        (void)hipEventRecord(e1, s);
        std::lock_guard<std::mutex> lk(g_timing.mu);
        g_timing.pending.emplace_back(e0, e1);
    }
};
}  // namespace

LvsKernelTimer::LvsKernelTimer(hipStream_t st) : impl(new ScopedKernelTimer(st)) {}
LvsKernelTimer::~LvsKernelTimer() { delete (ScopedKernelTimer*)impl; }

extern "C" int32_t lvs_timing_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_timing.mu);
    g_timing.on = on != 0;
    for (auto& pr : g_timing.pending) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    g_timing.pending.clear();
    g_timing.total_ms = 0;
    g_timing.launches = 0;
    g_timing.calls = 0;
    return LVS_OK;
}

extern "C" int32_t lvs_timing_read(double* out_total_ms, int64_t* out_launches) {
    std::lock_guard<std::mutex> lk(g_timing.mu);
    for (auto& pr : g_timing.pending) {
        LVS_HIP_CHECK(hipEventSynchronize(pr.second));
        float ms = 0;
        LVS_HIP_CHECK(hipEventElapsedTime(&ms, pr.first, pr.second));
        g_timing.total_ms += ms;
        g_timing.launches += 1;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    g_timing.pending.clear();
    if (out_total_ms) *out_total_ms = g_timing.total_ms;
    if (out_launches) *out_launches = g_timing.launches;
    return LVS_OK;
}

extern "C" int32_t lvs_timing_read_calls(double* out_total_ms, int64_t* out_launches, int64_t* out_calls, int32_t* out_kernel) {
    const int32_t rc = lvs_timing_read(out_total_ms, out_launches);
    if (rc != LVS_OK) return rc;
    std::lock_guard<std::mutex> lk(g_timing.mu);
    if (out_calls) *out_calls = g_timing.calls;
    if (out_kernel) *out_kernel = g_timing.kernel;
    return LVS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// packing
// ---------------------------------------------------------------------------------------------------------------
extern "C" int32_t lvs_packed_ld(int32_t d, int32_t pack_mode) {
    if (d <= 0) return LVS_EINVAL;
    int64_t dpad = lvs_round_up(d, LVS_BK);
    if (pack_mode == LVS_PACK_F16) return (int32_t)dpad;
    if (pack_mode == LVS_PACK_SPLIT) return (int32_t)(2 * dpad);
    return LVS_EINVAL;
}

namespace {

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// one wave per row, one element per lane and step: the fallback for d % 8 != 0 (rows not 16-byte aligned)
// flags (nullable device word): LVS_PACK_FLAG_NONFINITE when an input value is inf / NaN, LVS_PACK_FLAG_RANGE when a finite
// value (after the optional normalisation) lies outside fp16's range (|x| > 65504 would be stored as inf)
__device__ inline uint32_t pack_check(float x) {
    const float ax = fabsf(x);
    return !(ax <= 3.4028234663852886e38f) ? (uint32_t)LVS_PACK_FLAG_NONFINITE : (ax > 65504.0f ? (uint32_t)LVS_PACK_FLAG_RANGE : 0u);
}
__device__ inline void pack_report(uint32_t bad, uint32_t* flags) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) bad |= (uint32_t)__shfl_xor((int)bad, m, 64);
    if (bad && flags && (threadIdx.x & 63) == 0) atomicOr(flags, bad);
}

template <typename SrcT>
__global__ __launch_bounds__(256) void pack_rows_kernel(const SrcT* __restrict__ src, long long n, int d, int dpad,
                                                        int split, int normalize, float pw, _Float16* __restrict__ dst,
                                                        float* __restrict__ norms, uint32_t* __restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const SrcT* s = src + row * (long long)d;
    const int ld = split ? 2 * dpad : dpad;
    _Float16* o = dst + row * (long long)ld;
    float scale = 1.0f;
    if (normalize) {
        float ss = 0.f;
        for (int j = lane; j < d; j += 64) {
            float x = (float)s[j];
            ss += x * x;
        }
        ss = wave_sum(ss);
        scale = ss > 0.f ? 1.0f / sqrtf(ss) : 0.f;
    }
    float nn = 0.f;
    uint32_t bad = 0;
    for (int j = lane; j < dpad; j += 64) {
        float x = j < d ? ((float)s[j] * scale) * pw : 0.f;  // pw is a power of two: exact
        bad |= pack_check(x);
        _Float16 hi = (_Float16)x;
        float stored = (float)hi;
        o[j] = hi;
        if (split) {
            _Float16 lo = (_Float16)(x - (float)hi);
            o[dpad + j] = lo;
            stored += (float)lo;
        }
        nn += stored * stored;
    }
    if (norms) {
        nn = wave_sum(nn);
        if (lane == 0) norms[row] = nn;
    }
    pack_report(bad, flags);
}

typedef _Float16 pk_half8 __attribute__((ext_vector_type(8)));
typedef float pk_float4 __attribute__((ext_vector_type(4)));

__device__ inline void pack_load8(const float* s, float (&v)[8]) {
    pk_float4 a = *(const pk_float4*)s, b = *(const pk_float4*)(s + 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        v[t] = a[t];
        v[4 + t] = b[t];
    }
}
__device__ inline void pack_load8(const _Float16* s, float (&v)[8]) {
    pk_half8 a = *(const pk_half8*)s;
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = (float)a[t];
}

// one wave per row, 8 consecutive elements per lane and step (16-byte stores, 16/32-byte loads): d % 8 == 0
template <typename SrcT, int SPLIT>
__global__ __launch_bounds__(256) void pack_rows_vec_kernel(const SrcT* __restrict__ src, long long n, int d, int dpad,
                                                            int normalize, float pw, _Float16* __restrict__ dst,
                                                            float* __restrict__ norms, uint32_t* __restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const SrcT* s = src + row * (long long)d;
    _Float16* o = dst + row * (long long)(SPLIT ? 2 * dpad : dpad);
    float scale = 1.0f;
    if (normalize) {
        float ss = 0.f;
        for (int j = lane * 8; j < d; j += 512) {
            float v[8];
            pack_load8(s + j, v);
#pragma unroll
            for (int t = 0; t < 8; ++t) ss += v[t] * v[t];
        }
        ss = wave_sum(ss);
        scale = ss > 0.f ? 1.0f / sqrtf(ss) : 0.f;
    }
    float nn = 0.f;
    uint32_t bad = 0;
    for (int j = lane * 8; j < dpad; j += 512) {
        float v[8];
        if (j < d) {
            pack_load8(s + j, v);
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = 0.f;
        }
        pk_half8 hi, lo;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float x = (v[t] * scale) * pw;  // pw is a power of two: exact
            bad |= pack_check(x);
            hi[t] = (_Float16)x;
            float stored = (float)hi[t];
            if (SPLIT) {
                lo[t] = (_Float16)(x - (float)hi[t]);
                stored += (float)lo[t];
            }
            nn += stored * stored;
        }
        *(pk_half8*)(o + j) = hi;
        if (SPLIT) *(pk_half8*)(o + dpad + j) = lo;
    }
    if (norms) {
        nn = wave_sum(nn);
        if (lane == 0) norms[row] = nn;
    }
    pack_report(bad, flags);
}

// largest |x| over a matrix as the bit pattern of a non-negative float (atomicMax on the bits; inf / NaN propagate as the
// largest patterns, so the caller sees them)
template <typename SrcT>
__global__ __launch_bounds__(256) void absmax_kernel(const SrcT* __restrict__ src, long long total, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t b = __float_as_uint(fabsf((float)src[i])) & 0x7FFFFFFFu;
        m = b > m ? b : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t v = (uint32_t)__shfl_xor((int)m, o, 64);
        m = v > m ? v : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, long long ld16,
                                                          const long long* __restrict__ ids, long long n_ids,
                                                          uint4* __restrict__ dst) {
    const long long total = n_ids * ld16;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i / ld16, c = i - r * ld16;
        dst[i] = src[ids[r] * ld16 + c];
    }
}

__global__ __launch_bounds__(256) void gather_f32_kernel(const float* __restrict__ src, const long long* __restrict__ ids,
                                                         long long n_ids, float* __restrict__ dst) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_ids) dst[i] = src[ids[i]];
}

// k == 1: one thread per query, the best key over the parts (k-means assignment: 10 M queries, a handful of parts)
__global__ __launch_bounds__(256) void merge_top1_kernel(const u64* __restrict__ parts, int nparts, long long nq,
                                                         u64* __restrict__ out, long long out_ld) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    u64 best = 0;
    for (int p = 0; p < nparts; ++p) {
        u64 v = parts[(long long)p * nq + q];
        best = v > best ? v : best;
    }
    out[q * out_ld] = best;
}

// one wave per query: merge nparts sorted-or-not candidate lists of k keys into the best k (k <= 64)
__global__ __launch_bounds__(256) void merge_keys_kernel(const u64* __restrict__ parts, int nparts, long long nq, int k,
                                                         u64* __restrict__ out, long long out_ld,
                                                         const uint32_t* __restrict__ pred) {
    if (pred && *pred == 0u) return;  // predicated launch, see lvs_flat_search_keys
    const int lane = threadIdx.x & 63;
    const long long q = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= nq) return;
    const long long total = (long long)nparts * k;
    u64 best = 0;
    for (long long c0 = 0; c0 < total; c0 += 64) {
        long long c = c0 + lane;
        u64 v = 0;
        if (c < total) {
            long long p = c / k, j = c - p * k;
            v = parts[(p * nq + q) * k + j];
        }
        v = lvs_wave_sort_desc(v, lane);
        if (c0 == 0) {
            best = v;
        } else {
            u64 w = lvs_shfl_u64(v, 63 - lane);  // ascending copy
            best = best > w ? best : w;          // top-64 of the union, bitonic
            best = lvs_wave_bitonic_merge_desc(best, lane);
        }
    }
    if (lane < k) out[q * out_ld + lane] = best;
}

// the same merge with ONE WORKGROUP of 16 waves per query: every wave merges a sixteenth of the parts (its loads in flight
// next to the other waves'), wave 0 merges the sixteen results.  For the small-batch kernel's 64 .. 256 partial lists of a
// few dozen queries, where one wave per query left the merge (a chain of dependent loads) as long as a tenth of the scan.
__global__ __launch_bounds__(1024) void merge_keys_wide_kernel(const u64* __restrict__ parts, int nparts, long long nq, int k,
                                                               u64* __restrict__ out, long long out_ld) {
    __shared__ u64 sm[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long q = blockIdx.x;
    const int per = (nparts + 15) / 16;
    const int p0 = wave * per, p1 = p0 + per < nparts ? p0 + per : nparts;
    const long long total = p1 > p0 ? (long long)(p1 - p0) * k : 0;
    u64 best = 0;
    for (long long c0 = 0; c0 < total; c0 += 64) {
        const long long c = c0 + lane;
        u64 v = 0;
        if (c < total) {
            const long long p = p0 + c / k, j = c % k;
            v = parts[(p * nq + q) * k + j];
        }
        if (__builtin_amdgcn_ballot_w64(v != 0) == 0ull) continue;  // empty lists (seeded thresholds): nothing to merge
        v = lvs_wave_sort_desc(v, lane);
        const u64 w = lvs_shfl_u64(v, 63 - lane);  // ascending copy
        best = best > w ? best : w;                // top-64 of the union, bitonic
        best = lvs_wave_bitonic_merge_desc(best, lane);
    }
    sm[wave][lane] = best;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll 1
    for (int o = 1; o < 16; ++o) {
        const u64 w = sm[o][63 - lane];
        if (__builtin_amdgcn_ballot_w64(w != 0) == 0ull) continue;
        best = best > w ? best : w;
        best = lvs_wave_bitonic_merge_desc(best, lane);
    }
    if (lane < k) out[q * out_ld + lane] = best;
}

__global__ __launch_bounds__(256) void keys_to_result_kernel(const u64* __restrict__ keys, long long n, int metric,
                                                             const long long* __restrict__ id_map, float unscale,
                                                             float* __restrict__ D, long long* __restrict__ I) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 key = keys[i];
    const float FLT_MAX_ = 3.4028234663852886e38f;
    if (key == 0) {
        D[i] = metric == LVS_METRIC_IP ? -FLT_MAX_ : FLT_MAX_;
        I[i] = -1;
        return;
    }
    float better = lvs_unord32((uint32_t)(key >> 32)) * unscale;  // a power of two (operands packed with a scale): exact
    long long id = (long long)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
    D[i] = metric == LVS_METRIC_IP ? better : (0.0f - better);
    I[i] = id_map ? id_map[id] : id;
}

// certified nearest-row search: combine the slabs' (winner key, runner-up score) pairs of every query
__global__ __launch_bounds__(256) void merge_top2_kernel(const u64* __restrict__ keys, const float* __restrict__ sec,
                                                         int nslab, long long nq, u64* __restrict__ out_keys,
                                                         float* __restrict__ out_sec) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    u64 best = 0;
    for (int p = 0; p < nslab; ++p) {
        const u64 v = keys[(long long)p * nq + q];
        best = v > best ? v : best;
    }
    float s2 = -INFINITY;
    for (int p = 0; p < nslab; ++p) {
        const u64 v = keys[(long long)p * nq + q];
        const float cand = v == best ? sec[(long long)p * nq + q] : (v ? lvs_unord32((uint32_t)(v >> 32)) : -INFINITY);
        s2 = fmaxf(s2, cand);
    }
    out_keys[q] = best;
    out_sec[q] = s2;
}

// one wave per query: the exact score (fp32, hi + lo parts of both operands) of the pair (query q, row named by keys[q])
// replaces the approximate score inside the key
template <int QSPLIT, int BSPLIT>
__global__ __launch_bounds__(256) void rescore_keys_kernel(const _Float16* __restrict__ xb, long long ldb,
                                                           const _Float16* __restrict__ xq, long long ldq, int dpad,
                                                           int metric, const float* __restrict__ bn,
                                                           const float* __restrict__ qn, long long id_offset, long long nq,
                                                           int k, u64* __restrict__ keys) {
    const int lane = threadIdx.x & 63;
    const long long slot = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per (query, rank)
    if (slot >= nq * k) return;
    const long long q = slot / k;
    const u64 key = keys[slot];
    if (key == 0) return;
    const uint32_t id = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
    const long long row = (long long)id - id_offset;
    const _Float16* qr = xq + q * ldq;
    const _Float16* br = xb + row * ldb;
    float acc = 0.f;
    for (int j = lane * 8; j < dpad; j += 512) {
        const pk_half8 qh = *(const pk_half8*)(qr + j), bh = *(const pk_half8*)(br + j);
        pk_half8 ql, bl;
        if (QSPLIT) ql = *(const pk_half8*)(qr + dpad + j);
        if (BSPLIT) bl = *(const pk_half8*)(br + dpad + j);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float a = (float)qh[t] + (QSPLIT ? (float)ql[t] : 0.f);
            const float b = (float)bh[t] + (BSPLIT ? (float)bl[t] : 0.f);
            acc += a * b;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        float better = acc;
        if (metric == LVS_METRIC_L2) better = -fmaxf((qn[q] + bn[row]) - 2.0f * acc, 0.f);
        keys[slot] = lvs_pack_key(better, id);
    }
}

// one wave per query: keys[q][0..k) sorted descending in place (k <= 64)
__global__ __launch_bounds__(256) void sort_keys_kernel(u64* __restrict__ keys, long long nq, int k) {
    const int lane = threadIdx.x & 63;
    const long long q = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= nq) return;
    u64 v = lane < k ? keys[q * k + lane] : 0ull;
    v = lvs_wave_sort_desc(v, lane);
    if (lane < k) keys[q * k + lane] = v;
}

// Certificate of a one-pass ("hi" parts only) top-k search with k1 > k list slots:
//   approx [nq][k1]: the one-pass keys, best first;  exact [nq][k1]: the same candidates rescored exactly and re-sorted.
// A row outside the candidate list has a one-pass score <= the list's last one-pass score s_min, hence an exact score
// <= s_min + bound(q); if the k-th exact score of the candidates is STRICTLY above that, the candidates' exact top k is the
// exact top k.  Queries that fail the test are appended to out_idx.  An unfilled list (fewer than k1 rows) holds every row.
__global__ __launch_bounds__(256) void certify_topk_kernel(const u64* __restrict__ approx, const u64* __restrict__ exact,
                                                           const float* __restrict__ qn, long long nq, int k1, int k,
                                                           float scale, float slack, float band,
                                                           long long* __restrict__ out_idx,
                                                           unsigned long long* __restrict__ out_count) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool open = false;
    if (q < nq) {
        const u64 last = approx[q * k1 + k1 - 1];
        const u64 kth = exact[q * k1 + k - 1];
        const float bound = scale * sqrtf(qn ? qn[q] : 1.0f) + slack;
        if (band > 0.f) {
            // banded list: a row outside it scored below max(last slot, k-th one-pass score - band * bound) when it was
            // turned away (both only rise during the search); an unfilled list no longer means "every row is here"
            const u64 akth = approx[q * k1 + k - 1];
            if (kth != 0 && akth != 0) {
                float s_min = lvs_unord32((uint32_t)(akth >> 32)) - band * bound;
                if (last != 0) s_min = fmaxf(s_min, lvs_unord32((uint32_t)(last >> 32)));
                open = !(lvs_unord32((uint32_t)(kth >> 32)) > s_min + bound);
            }
        } else if (last != 0 && kth != 0) {
            const float s_min = lvs_unord32((uint32_t)(last >> 32));
            const float s_k = lvs_unord32((uint32_t)(kth >> 32));
            open = !(s_k > s_min + bound);
        }
    }
    const unsigned long long m = __ballot(open);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(out_count, (unsigned long long)__popcll(m));
    base = lvs_shfl_u64(base, 0);
    if (open) out_idx[base + __popcll(m & ((1ull << lane) - 1ull))] = q;
}

// queries whose winner is NOT certified by its margin: (best score - runner-up score) <= scale * |q| + slack.
// Their indices are appended to out_idx (order unspecified), *out_count counts them.
// With `stats` (device: [0] = R^2 largest squared row norm, [1] = E^2 largest squared lo-part norm of the corpus rows) the
// bound is (coef[0] E + coef[1] R) |q| + coef[2] + coef[3] R + coef[4] R^2 - the k-means loop keeps those statistics on the
// device (lvs_kmeans_update_centroids) so that no iteration waits for the host; without it: scale |q| + slack.
struct MarginCoef {
    float c[5];
};
__global__ __launch_bounds__(256) void margin_select_kernel(const u64* __restrict__ keys, const float* __restrict__ sec,
                                                            const float* __restrict__ qn, long long nq, float scale,
                                                            float slack, const float* __restrict__ stats, MarginCoef coef,
                                                            long long per_block, long long* __restrict__ out_idx,
                                                            unsigned long long* __restrict__ out_count) {
    if (stats) {
        const float R = sqrtf(stats[0]), E = sqrtf(stats[1]);
        scale = coef.c[0] * E + coef.c[1] * R;
        slack = coef.c[2] + coef.c[3] * R + coef.c[4] * R * R;
    }
    // One workgroup owns `per_block` consecutive queries: it counts its uncertified ones, reserves their output range
    // with ONE atomic (10 M queries at 0.4 % uncertified were 30 000 same-address atomics = 1.2 ms when every wave
    // reserved its own), then writes them in a second sweep over the same 16 bytes per query.
    __shared__ unsigned long long s_base;
    __shared__ unsigned s_count;
    const long long q_begin = (long long)blockIdx.x * per_block;
    const long long q_end = q_begin + per_block < nq ? q_begin + per_block : nq;
    auto is_open = [&](long long q) {
        const u64 kq = keys[q];
        if (!kq) return false;
        const float best = lvs_unord32((uint32_t)(kq >> 32));
        const float bound = scale * sqrtf(qn ? qn[q] : 1.0f) + slack;
        return !((best - sec[q]) > bound);  // also true for NaN / inf - inf: never certify what cannot be compared
    };
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    unsigned mine = 0;
    for (long long q = q_begin + threadIdx.x; q < q_end; q += blockDim.x) mine += is_open(q) ? 1u : 0u;
    if (mine) atomicAdd(&s_count, mine);
    __syncthreads();
    if (s_count == 0) return;
    if (threadIdx.x == 0) {
        s_base = atomicAdd(out_count, (unsigned long long)s_count);
        s_count = 0;
    }
    __syncthreads();
    for (long long q = q_begin + threadIdx.x; q < q_end; q += blockDim.x)
        if (is_open(q)) out_idx[s_base + atomicAdd(&s_count, 1u)] = q;
}

}  // namespace

extern "C" int32_t lvs_pack_rows(const void* src, int32_t src_dtype, int64_t n, int32_t d, int32_t pack_mode,
                                 int32_t normalize, void* dst, float* out_norms_sq, void* stream) {
    return lvs_pack_rows_checked(src, src_dtype, n, d, pack_mode, normalize, 0, dst, out_norms_sq, nullptr, stream);
}

extern "C" int32_t lvs_pack_rows_checked(const void* src, int32_t src_dtype, int64_t n, int32_t d, int32_t pack_mode,
                                         int32_t normalize, int32_t scale_exp, void* dst, float* out_norms_sq,
                                         uint32_t* out_flags, void* stream) {
    LVS_REQUIRE(scale_exp >= -100 && scale_exp <= 100, "scale_exp %d out of range", scale_exp);
    const float pw = ldexpf(1.0f, scale_exp);
    LVS_REQUIRE(n >= 0 && d > 0, "bad shape n=%lld d=%d", (long long)n, d);
    LVS_REQUIRE(pack_mode == LVS_PACK_F16 || pack_mode == LVS_PACK_SPLIT, "bad pack_mode %d", pack_mode);
    LVS_REQUIRE(src_dtype == LVS_DTYPE_F32 || src_dtype == LVS_DTYPE_F16, "bad src_dtype %d", src_dtype);
    if (n == 0) return LVS_OK;
    LVS_REQUIRE(src && dst, "NULL buffer");
    const int dpad = (int)lvs_round_up(d, LVS_BK);
    LVS_DEVICE_GUARD(stream);
    dim3 block(256), grid((unsigned)lvs_ceil_div(n, 4));
    hipStream_t st = (hipStream_t)stream;
    const bool split = pack_mode == LVS_PACK_SPLIT;
    const bool vec = d % 8 == 0 && ((uintptr_t)src & 15) == 0;  // every source row 16-byte aligned
    _Float16* o = (_Float16*)dst;
    if (vec && src_dtype == LVS_DTYPE_F32 && split)
        hipLaunchKernelGGL((pack_rows_vec_kernel<float, 1>), grid, block, 0, st, (const float*)src, (long long)n, d,
                           dpad, normalize, pw, o, out_norms_sq, out_flags);
    else if (vec && src_dtype == LVS_DTYPE_F32)
        hipLaunchKernelGGL((pack_rows_vec_kernel<float, 0>), grid, block, 0, st, (const float*)src, (long long)n, d,
                           dpad, normalize, pw, o, out_norms_sq, out_flags);
    else if (vec && split)
        hipLaunchKernelGGL((pack_rows_vec_kernel<_Float16, 1>), grid, block, 0, st, (const _Float16*)src, (long long)n,
                           d, dpad, normalize, pw, o, out_norms_sq, out_flags);
    else if (vec)
        hipLaunchKernelGGL((pack_rows_vec_kernel<_Float16, 0>), grid, block, 0, st, (const _Float16*)src, (long long)n,
                           d, dpad, normalize, pw, o, out_norms_sq, out_flags);
    else if (src_dtype == LVS_DTYPE_F32)
        hipLaunchKernelGGL(pack_rows_kernel<float>, grid, block, 0, st, (const float*)src, (long long)n, d, dpad,
                           split, normalize, pw, o, out_norms_sq, out_flags);
    else
        hipLaunchKernelGGL(pack_rows_kernel<_Float16>, grid, block, 0, st, (const _Float16*)src, (long long)n, d,
                           dpad, split, normalize, pw, o, out_norms_sq, out_flags);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_absmax(const void* src, int32_t src_dtype, int64_t n, int32_t d, uint32_t* inout_bits, void* stream) {
    LVS_REQUIRE(n >= 0 && d > 0 && inout_bits, "bad arguments");
    LVS_REQUIRE(src_dtype == LVS_DTYPE_F32 || src_dtype == LVS_DTYPE_F16, "bad src_dtype %d", src_dtype);
    if (n == 0) return LVS_OK;
    LVS_REQUIRE(src, "NULL buffer");
    LVS_DEVICE_GUARD(stream);
    const long long total = (long long)n * d;
    const unsigned grid = (unsigned)(lvs_ceil_div(total, 256 * 8) < 4096 ? lvs_ceil_div(total, 256 * 8) : 4096);
    if (src_dtype == LVS_DTYPE_F32)
        hipLaunchKernelGGL(absmax_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)src, total, inout_bits);
    else
        hipLaunchKernelGGL(absmax_kernel<_Float16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, total,
                           inout_bits);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_gather_rows(const void* src, int32_t ld, const int64_t* ids, int64_t n_ids, void* dst,
                                   void* stream) {
    LVS_REQUIRE(ld > 0 && ld % 8 == 0, "ld must be a positive multiple of 8 halfs");
    if (n_ids == 0) return LVS_OK;
    LVS_REQUIRE(src && ids && dst && n_ids > 0, "bad arguments");
    LVS_DEVICE_GUARD(stream);
    long long ld16 = ld / 8;
    long long total = n_ids * ld16;
    unsigned grid = (unsigned)(lvs_ceil_div(total, 256) < 16384 ? lvs_ceil_div(total, 256) : 16384);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, ld16,
                       (const long long*)ids, (long long)n_ids, (uint4*)dst);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_gather_f32(const float* src, const int64_t* ids, int64_t n_ids, float* dst, void* stream) {
    if (n_ids == 0) return LVS_OK;
    LVS_REQUIRE(src && ids && dst && n_ids > 0, "bad arguments");
    LVS_DEVICE_GUARD(stream);
    hipLaunchKernelGGL(gather_f32_kernel, dim3((unsigned)lvs_ceil_div(n_ids, 256)), dim3(256), 0,
                       (hipStream_t)stream, src, (const long long*)ids, (long long)n_ids, dst);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// search
// ---------------------------------------------------------------------------------------------------------------
namespace {
// lvs_flat_search_keys_hi: the same search planned over the fp16 "hi" parts only (one K segment whatever the pack modes)
thread_local bool g_hi_only = false;
struct HiOnlyScope {
    HiOnlyScope() { g_hi_only = true; }
    ~HiOnlyScope() { g_hi_only = false; }
};
// lvs_flat_search_keys_hi_banded: list slots beyond a per-query band below the kc-th best stay empty (LvsTileArgs::kc)
struct BandSpec {
    int kc;
    float bscale, bslack;
};
thread_local BandSpec g_band = {0, 0.f, 0.f};
struct BandScope {
    BandScope(int kc, float bscale, float bslack) { g_band = BandSpec{kc, bscale, bslack}; }
    ~BandScope() { g_band = BandSpec{0, 0.f, 0.f}; }
};

// bytes reserved for the small-batch kernel's candidate lists [ranges][nq][k]
static inline int64_t lvs_stream_parts_bytes(int64_t nq, int k) {
    return lvs_round_up((int64_t)((lvs_tune_set("LVS_STREAM_WGS") ? LVS_STREAM_MAXWG : 256) + 8) * (nq > 0 ? nq : 1) *
                            (k > 0 ? k : 1) * 8, 256);
}

// ... and for the register-resident-queries kernel beyond 256 queries: `groups` query groups x at most 256 / groups corpus
// ranges = at most 65 536 lists (+ slack), and as many sample scores
static inline int64_t lvs_rq_parts_bytes(int64_t nq, int k) {
    if (nq <= LVS_STREAM_MAXQ) return lvs_stream_parts_bytes(nq, k);
    return lvs_round_up((int64_t)(65536 + 8 * nq) * (k > 0 ? k : 1) * 8, 256);
}
static inline int64_t lvs_rq_seed_bytes(int64_t nq) {
    if (nq <= LVS_STREAM_MAXQ) return lvs_round_up((int64_t)(nq > 0 ? nq : 1) * (LVS_STREAM_MAXWG + 8) * 4, 256);
    return lvs_round_up((int64_t)(65536 + 8 * nq) * 4, 256);
}

struct Plan {
    int dpad, nkd, nk, ldb, ldq, nseg;
    int seg_q[3], seg_c[3];
    int ntiles, nqt, nslab, tiles_per_slab;
    int kpass, npass;
    int gq, lead_slabs;
    int v2;  // 1: 256 x 256 geometry (k <= LVS2_KCAP, TOP1 / SCORES / RANGE); 0: 256 x 128 geometry (k > LVS2_KCAP)
    int64_t off_gtau, off_partial, off_pass, off_seed, total;
};

// Seeded thresholds of a list launch.  A launch cuts the corpus into slabs that run concurrently, and a slab whose queries
// have no threshold yet pays a cold start of ~k (1 + ln(slab rows / k)) lock-protected insertions per query: with few query
// tiles EVERY slab starts cold (256 queries x 1 M rows: 0.36 ms of a 1.08 ms launch), with many it is the leading slab of
// every query tile (100 k x 125 k: 1.3 ms of 21.2).  So LVS_MODE_SEED first scores a sample (the first rows of the shard,
// one tile per workgroup, no lists), one wave per query takes the k-th largest of the per-tile maxima as the starting
// threshold, and the list launch only inserts rows that reach it.  Exact for the same reason as the small-batch kernel's
// seeding: every value is a real row's score, computed by the same instructions in the same order as the list mode
// computes it, and at least k rows reach the threshold.  Sample size: a sixteenth of the corpus for up to 64 query tiles
// (the sample pass is negligible there and a tight threshold saves the most); beyond, the pass costs MFMA time in
// proportion to the queries, and the smallest useful sample - max(k, 8) tiles: the threshold is the k-th largest of one
// maximum per tile - already removes the first tiles' insertions, which is most of a cold start (100 k x 1 M, same box:
// 10 / 61 tiles 138.5 / 139.5 ms against 140.0 unseeded; 100 k x 125 k 21.2 -> 20.4 ms, 25 k x 500 k 20.9 -> 20.3;
// tools/seed_big_sweep.py), so it is nb / 512 rows there; never more than a sixteenth of the corpus or 64 tiles.
// Returns the sample tiles or 0: no seeding.
#define LVS_TILE_SEED_MAXTILES 64
#define LVS_TILE_SEED_MAXQT 4096  // 2^20 queries: 256 B of seed workspace per query at most
int tile_seed_tiles(int64_t nq, int64_t nb, int k) {
    if (lvs_tune("LVS_TILE_SEED", 1) == 0 || k < 1 || k > LVS_KPASS) return 0;
    const int64_t nqt = lvs_ceil_div(nq, LVS2_BQ);
    long long maxqt = lvs_tune("LVS_TILE_SEED_MAXQT", LVS_TILE_SEED_MAXQT);
    if (maxqt > LVS_TILE_SEED_MAXQT) maxqt = LVS_TILE_SEED_MAXQT;  // the workspace holds seeds up to here
    if (nqt > maxqt) return 0;
    const int64_t kt = k > 8 ? k : 8;
    if (nb < 16 * (int64_t)LVS_BC * kt) return 0;
    int64_t tiles = nb / LVS_BC / (nqt <= 64 ? lvs_tune("LVS_TILE_SEED_DIV", 16) : lvs_tune("LVS_TILE_SEED_DIV_BIG", 512));
    if (tiles < kt) tiles = kt;
    if (tiles > LVS_TILE_SEED_MAXTILES) tiles = LVS_TILE_SEED_MAXTILES;
    return (int)tiles;
}

int make_plan(int64_t nq, int64_t nb, int32_t d, int32_t xb_pack, int32_t xq_pack, int32_t k, Plan& p,
              bool force_v2 = false, int64_t min_slabs = 0, int64_t max_slabs_cap = 0, bool allow_l2 = true) {
    if (nq < 0 || nb < 0 || d <= 0 || k < 0) return LVS_EINVAL;
    if (xb_pack != LVS_PACK_F16 && xb_pack != LVS_PACK_SPLIT) return LVS_EINVAL;
    if (xq_pack != LVS_PACK_F16 && xq_pack != LVS_PACK_SPLIT) return LVS_EINVAL;
    p.dpad = (int)lvs_round_up(d, LVS_BK);
    p.nkd = p.dpad / LVS_BK;
    p.ldb = xb_pack == LVS_PACK_SPLIT ? 2 * p.dpad : p.dpad;
    p.ldq = xq_pack == LVS_PACK_SPLIT ? 2 * p.dpad : p.dpad;
    // x = hi + lo on either side; keep hi*hi plus the first-order cross terms (lo*lo ~ 2^-22 relative is dropped)
    p.nseg = 0;
    auto seg = [&](int qo, int co) {
        p.seg_q[p.nseg] = qo;
        p.seg_c[p.nseg] = co;
        ++p.nseg;
    };
    p.seg_q[0] = p.seg_q[1] = p.seg_q[2] = p.seg_c[0] = p.seg_c[1] = p.seg_c[2] = 0;
    seg(0, 0);
    if (xb_pack == LVS_PACK_SPLIT && !g_hi_only) seg(0, p.dpad);
    if (xq_pack == LVS_PACK_SPLIT && !g_hi_only) seg(p.dpad, 0);
    p.nk = p.nseg * p.nkd;
    p.ntiles = (int)lvs_ceil_div(nb > 0 ? nb : 1, LVS_BC);
    // geometry: 256 queries per tile with 15 list slots (k <= 15 and every non-top-k mode), else 128 queries / 56 slots
    p.v2 = force_v2 || k <= LVS2_KCAP;
    // a single, at most half-full query tile: the 128-query geometry does half the (padding) MFMA work
    if (!force_v2 && k > 1 && nq <= LVS3_BQ && lvs_tune("LVS_SMALLQ", 1) != 0) p.v2 = 0;
    p.nqt = (int)lvs_ceil_div(nq > 0 ? nq : 1, p.v2 ? LVS2_BQ : LVS3_BQ);
    // XCD group = gq query tiles x (32 / gq) slabs resident on one XCD at a time.  Wide groups (one corpus stream per
    // XCD) are fastest (profiles/r01_tuning.md) but must be full: groups are dealt round-robin to the 8 XCDs, so a
    // half-empty group idles half an XCD.  Take the widest gq that wastes < 15 % of its query-tile slots.
    p.gq = 1;
    const int gq_max = 32;
    for (int g = gq_max; g >= 1; g >>= 1) {
        const double fill = (double)p.nqt / (double)(lvs_ceil_div(p.nqt, g) * g);
        if (fill >= 0.85) {
            p.gq = g;
            break;
        }
    }
    // enough (query tile, slab) items to load-balance 256 CUs, slabs kept >= 32 tiles, a multiple of gs slabs
    int64_t want = lvs_ceil_div(4096, p.nqt);
    // Long corpus streams: 8 query tiles x 4 slabs per XCD share BOTH operands through the 4 MB L2 (hit rate 37 % -> 70 %,
    // half the fabric traffic, +3 % clock).  Measured against the 32 x 1 groups (profiles/r02_tuning.md): 100 k x 1 M
    // +9 % on the main loop, 100 k x 500 k 78.1 -> 72.6 ms, 100 k x 250 k 41.5 -> 38.5 ms, no difference at 100 k x 125 k
    // (12 slabs of 40 tiles) - so it is used whenever >= 8 narrow slabs of >= 40 tiles exist.
    int64_t slabs_l2 = 0;
    if (allow_l2 && p.v2 && p.gq > 8 && p.nqt >= 64 && min_slabs == 0) {
        const int64_t l2_min_slabs = lvs_tune("LVS_L2_MIN_SLABS", 8);
        const int64_t n8 = lvs_round_up(want > l2_min_slabs ? want : l2_min_slabs, 4);
        const int64_t min_tiles = lvs_tune("LVS_L2_MIN_TILES", 40);
        if (p.ntiles / n8 >= min_tiles) {
            p.gq = 8;
            slabs_l2 = n8;
        }
    }
    if (lvs_tune_set("LVS_GQ")) {  // -DLVS_TUNING builds only
        const int v = (int)lvs_tune("LVS_GQ", 0);
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32) {
            p.gq = v;
            slabs_l2 = 0;
        }
    }
    const int gs = 32 / p.gq;
    // few query tiles: allow slabs down to 8 tiles so that every CU gets work (cold starts are cheap)
    int64_t max_slabs = lvs_ceil_div(p.ntiles, p.nqt >= 8 ? 32 : (p.nqt >= 2 ? 16 : 8));
    int64_t s = want < 1 ? 1 : want;
    if (s > max_slabs) s = max_slabs;
    s = lvs_round_up(s, gs);
    if (s > p.ntiles) s = p.ntiles;
    if (s < 1) s = 1;
    p.lead_slabs = 0;
    if (slabs_l2 > 0) {  // one leading slab per query tile in wide groups + slabs_l2 slabs in the narrow groups
        p.lead_slabs = lvs_tune("LVS_LEAD", 1) != 0;
        s = slabs_l2 + p.lead_slabs;
    }
    if (min_slabs > 0 && s < min_slabs) {  // the caller needs at least this many per-slab candidate lists
        s = lvs_round_up(min_slabs, gs);
        if (s > p.ntiles) s = p.ntiles;
    }
    if (max_slabs_cap > 0 && s > max_slabs_cap) s = max_slabs_cap;  // ... and at most this many
    // Tail effect.  A launch is nqt * s (query tile, slab) items of equal length; an XCD (32 CUs, one workgroup each) runs
    // 32 of ITS items at a time, so the launch lasts max over the XCDs of ceil(items of that XCD / 32) item-times
    // (lvs_tile_xcd_rounds follows item_of_block's deal exactly, incl. the last groups spread over all XCDs).  Measured
    // with the slab count swept (profiles/r02_tuning.md): 100 k x 1 M at 21 slabs 142.8-147.6 ms, 17 140.5-142.4,
    // 13 139.2-139.6.  So among the slab counts near the heuristic's choice (same group shape, slabs not shorter than the
    // heuristic allows) take the one with the smallest estimated time = rounds x (item length + its fixed cost).
    // (r3: from one full round of items on - launches of a few query tiles run seeded, so an item's fixed cost is the same
    // two tiles there, and e.g. 256 queries x 1 M rows go from 489 slabs x 8 tiles in two rounds to 245 x 16 in one)
    if (min_slabs == 0 && max_slabs_cap == 0 && (int64_t)p.nqt * s >= lvs_tune("LVS_TAIL_MIN_ITEMS", 256) &&
        lvs_tune("LVS_TAIL", 1) != 0) {
        const int64_t lead = p.lead_slabs;
        const int64_t hi_lim = slabs_l2 > 0 ? lead + p.ntiles / lvs_tune("LVS_L2_MIN_TILES", 40) : max_slabs;
        double best_cost = 1e30;
        int64_t best = s;
        for (int64_t c = lead + gs; c <= s + s / 4 && c <= hi_lim && c <= p.ntiles; c += gs) {
            if (10 * c < 3 * s) continue;  // stay within [0.3 s, 1.25 s]
            const int64_t tps = lvs_ceil_div(p.ntiles, c), ns = lvs_ceil_div(p.ntiles, tps);
            const int rounds = lvs_tile_xcd_rounds(p.nqt, (int)ns, p.gq, (int)lead);
            // an item also pays a list cold start and its candidate write-out: about two tiles' worth.  (Round 4 swept a larger
            // fixed cost - 6, 12, 20 tiles, i.e. fewer and longer slabs - over ten shapes: slower everywhere, by up to 19 % at
            // 100 k x 125 k with 3 x 163 instead of 9 x 55 tiles; fine items balance the XCDs better than the round count
            // says.  profiles/r05d_plan_sweep.log)
            const double cost = (double)rounds * (double)(tps + 2);
            const int64_t dist = c > s ? c - s : s - c, bdist = best > s ? best - s : s - best;
            if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && dist < bdist)) {
                best_cost = cost;
                best = c;
            }
        }
        s = best;
    }
    if (lvs_tune_set("LVS_NSLAB")) {  // -DLVS_TUNING builds only
        const int64_t v = lvs_tune("LVS_NSLAB", 0);
        if (v >= 1 && v <= p.ntiles) s = v;
    }
    p.tiles_per_slab = (int)lvs_ceil_div(p.ntiles, s);
    p.nslab = (int)lvs_ceil_div(p.ntiles, p.tiles_per_slab);
    // With around a hundred query tiles the item count pushes the 8 x 4 plan to 45-49 slabs of 40-90 tiles, and there the wide
    // groups win (same box, profiles/r05d_plan_sweep.log: 20 k x 1 M 31.1 -> 28.5 ms, 25 k x 1 M 36.7 -> 35.6, 25 k x 500 k 19.4 ->
    // 18.6); with 196 / 391 query tiles the 8 x 4 groups stay ahead even on 55-109-tile slabs (100 k x 125 k 20.0 vs 20.5 ms,
    // 100 k x 250 k 37.4 vs 38.6), and from 118 query tiles on the slabs are long anyway.
    if (slabs_l2 > 0 && p.nqt <= lvs_tune("LVS_L2_SHORT_NQT", 128) && p.tiles_per_slab < lvs_tune("LVS_L2_MIN_FINAL", 120) &&
        !lvs_tune_set("LVS_GQ") && !lvs_tune_set("LVS_NSLAB"))
        return make_plan(nq, nb, d, xb_pack, xq_pack, k, p, force_v2, min_slabs, max_slabs_cap, false);
#ifdef LVS_TUNING
    if (lvs_tune("LVS_PLAN_PRINT", 0) != 0)
        fprintf(stderr, "lvs plan: nq %lld nb %lld k %d -> nqt %d ntiles %d gq %d lead %d slabs %d x %d tiles, items %lld, xcd rounds %d\n",
                (long long)nq, (long long)nb, (int)k, p.nqt, p.ntiles, p.gq, p.lead_slabs, p.nslab, p.tiles_per_slab,
                (long long)p.nqt * p.nslab, lvs_tile_xcd_rounds(p.nqt, p.nslab, p.gq, p.lead_slabs));
#endif
    p.kpass = k < LVS_KPASS ? (k > 0 ? k : 1) : LVS_KPASS;  // k <= 15 -> one pass on the 256-query geometry
    p.npass = k > 0 ? (int)lvs_ceil_div(k, p.kpass) : 0;
    int64_t off = 256;  // the first 256 bytes hold status words (the large-k overflow flag) in every layout
    p.off_gtau = off;
    off += lvs_round_up(nq * 4, 256);
    p.off_partial = off;
    off += lvs_round_up((int64_t)p.nslab * nq * p.kpass * 8, 256);
    p.off_pass = off;  // [nq][kpass] merged keys of one pass (multi-pass only)
    off += p.npass > 1 ? lvs_round_up(nq * p.kpass * 8, 256) : 0;
    // the small-batch kernel: one k-list per (corpus range, query) written from off_partial onwards, then the per-range best
    // scores of the sample that seeds its thresholds
    if (nq <= LVS_STREAM_MAXQ && k <= LVS_KPASS)
        off += lvs_stream_parts_bytes(nq, k) + lvs_round_up((int64_t)(nq > 0 ? nq : 1) * (LVS_STREAM_MAXWG + 8) * 4, 256);
    else if (k <= LVS_RQ_KMAX) {  // lvs_rq_kernel in query groups, beyond LVS_RQ_MAXQ queries one chunk of that many at a time
        int64_t cmax = lvs_tune("LVS_RQ_CHUNK", LVS_RQ_CHUNK_DEFAULT);
        if (cmax < LVS_RQ_MAXQ) cmax = LVS_RQ_MAXQ;
        if (cmax > LVS_RQ_CHUNK_MAX) cmax = LVS_RQ_CHUNK_MAX;
        const int64_t cq = nq <= cmax ? nq : cmax;
        off += lvs_rq_parts_bytes(cq, k) + lvs_rq_seed_bytes(cq);
    }
    p.off_seed = off;  // [sample tiles][nq] per-tile best scores of a seeded list launch (tile_seed_tiles)
    if (p.npass == 1) off += lvs_round_up((int64_t)(nq > 0 ? nq : 1) * tile_seed_tiles(nq, nb, k) * 4, 256);
    p.total = off;
    return LVS_OK;
}

// One workgroup per query: n candidate keys src[q * stride_q + (i / inner) * stride_outer + (i % inner)], i < n, are
// sorted descending (bitonic, in LDS, P2 = power of two >= n slots) and the k-th largest (kth_out[q]; 0 when n < k)
// and/or the k largest (topk_out[q * out_ld + 0..k)) are written.  With counts != NULL, n = min(counts[q], n_max) and
// *overflow is set when some counts[q] > n_max (candidates were dropped by the producer).
__global__ __launch_bounds__(256) void select_keys_kernel(const u64* __restrict__ src, long long stride_q, int inner,
                                                          long long stride_outer, int n_max,
                                                          const uint32_t* __restrict__ counts, int k, int P2,
                                                          u64* __restrict__ kth_out, u64* __restrict__ topk_out,
                                                          long long out_ld, uint32_t* __restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) char sel_smem[];
    u64* sk = (u64*)sel_smem;
    const long long q = blockIdx.x;
    int n = n_max;
    if (counts) {
        uint32_t c = counts[q];
        if (c > (uint32_t)n_max) {
            if (threadIdx.x == 0 && overflow) atomicOr(overflow, 1u);
            c = (uint32_t)n_max;
        }
        n = (int)c;
    }
    const u64* base = src + q * stride_q;
    for (int i = threadIdx.x; i < P2; i += 256) {
        u64 v = 0;
        if (i < n) v = base[(long long)(i / inner) * stride_outer + (i % inner)];
        sk[i] = v;
    }
    __syncthreads();
    for (int size = 2; size <= P2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < P2; i += 256) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool desc = (i & size) == 0;  // the last merge (size == P2) is descending everywhere
                    const u64 x = sk[i], y = sk[j];
                    if ((x < y) == desc) {
                        sk[i] = y;
                        sk[j] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (kth_out && threadIdx.x == 0) kth_out[q] = k <= P2 ? sk[k - 1] : 0;
    if (topk_out)
        for (int j = threadIdx.x; j < k; j += 256) topk_out[q * out_ld + j] = j < P2 ? sk[j] : 0;
}

int pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}
constexpr int LVS_SELECT_MAX = 4096;  // most candidates per query select_keys_kernel sorts (32 KB of LDS)

hipError_t launch_select(const u64* src, long long stride_q, int inner, long long stride_outer, int n_max,
                         const uint32_t* counts, int k, u64* kth_out, u64* topk_out, long long out_ld, uint32_t* overflow,
                         int64_t nq, hipStream_t st) {
    const int P2 = pow2_ceil(n_max < 2 ? 2 : n_max);
    hipLaunchKernelGGL(select_keys_kernel, dim3((unsigned)nq), dim3(256), (size_t)P2 * 8, st, src, stride_q, inner,
                       stride_outer, n_max, counts, k, P2, kth_out, topk_out, out_ld, overflow);
    return hipGetLastError();
}

// One workgroup per query: out[q][0..k) = the k largest of {acc[q][0..k) if acc} U {parts[p][q][0..k) : p0 <= p < p0 + np}
// (bitonic sort of P2 >= (np + (acc != 0)) * k slots in LDS).  acc may alias out: every input of the query is in LDS
// before the first result is written.
__global__ __launch_bounds__(256) void merge_long_kernel(const u64* __restrict__ parts, int p0, int np,
                                                         const u64* acc, long long nq, int k, int P2, u64* out) {
    extern __shared__ __attribute__((aligned(16))) char sel_smem[];
    u64* sk = (u64*)sel_smem;
    const long long q = blockIdx.x;
    const int nacc = acc ? k : 0;
    const int n = nacc + np * k;
    for (int i = threadIdx.x; i < P2; i += 256) {
        u64 v = 0;
        if (i < nacc) {
            v = acc[q * k + i];
        } else if (i < n) {
            const int c = i - nacc, p = c / k, j = c - p * k;
            v = parts[((long long)(p0 + p) * nq + q) * k + j];
        }
        sk[i] = v;
    }
    __syncthreads();
    for (int size = 2; size <= P2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < P2; i += 256) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool desc = (i & size) == 0;
                    const u64 x = sk[i], y = sk[j];
                    if ((x < y) == desc) {
                        sk[i] = y;
                        sk[j] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int j = threadIdx.x; j < k; j += 256) out[q * k + j] = sk[j];
}

hipError_t launch_merge_long(const u64* parts, int p0, int np, const u64* acc, int64_t nq, int k, u64* out,
                             hipStream_t st) {
    const int n = (np + (acc ? 1 : 0)) * k;
    const int P2 = pow2_ceil(n < 2 ? 2 : n);
    if (P2 > LVS_SELECT_MAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(merge_long_kernel, dim3((unsigned)nq), dim3(256), (size_t)P2 * 8, st, parts, p0, np, acc,
                       (long long)nq, k, P2, out);
    return hipGetLastError();
}

// ---- large k in two phases (k > LVS_KPASS) ----------------------------------------------------------------------
// Phase A is a 15-per-slab top-k pass with at least 3k/15 slabs that do NOT share thresholds, so every slab's list is
// its own exact top 15.  The lists hold 15 * nslab DISTINCT rows,
// so the k-th largest of all their keys is a valid lower bound T of the query's true k-th best key - and a tight one
// unless one slab holds more than 15 of the true top k.  Phase B (LVS_MODE_COLLECT) streams the corpus once more and
// drops every key >= T into the query's bucket (>= k keys by construction); phase C sorts the bucket and keeps k.
// Exact; two corpus passes whatever k is.  If a bucket overflows (heavily tied or slab-clustered data) the caller falls
// back to ceil(k / LVS_KPASS) selection passes.
struct TwoPhasePlan {
    Plan a;             // phase A geometry: 15-slot lists (256 x 256) for moderate k, 56-slot lists (256 x 128) beyond
    Plan b;             // phase B (collect) geometry: always 256 x 256
    int slots;          // list slots per (query, slab) in phase A
    int capacity;       // bucket slots per query
    int64_t off_thr, off_cnt, off_bucket, total;
};

bool make_two_phase_plan(int64_t nq, int64_t nb, int32_t d, int32_t xb_pack, int32_t xq_pack, int32_t k,
                         TwoPhasePlan& tp) {
    if (k <= LVS_KPASS || nq <= 0 || nb <= 0) return false;
    if (lvs_tune("LVS_TWO_PHASE", 1) == 0) return false;
    // 15-slot lists on the fast geometry beat 56-slot lists on the 128-query geometry at every k tried (100 .. 1000:
    // 329 vs 402 ms, 545 vs 607 ms at 100 k x 1 M), although they need four times the slabs
    tp.slots = LVS2_KCAP;
    if (lvs_tune("LVS_TWO_PHASE_SLOTS", LVS2_KCAP) == LVS3_KCAP) tp.slots = LVS3_KCAP;
    const int64_t cap_slabs = LVS_SELECT_MAX / tp.slots;        // the selection kernel sorts at most 4096 keys per query
    int64_t want_slabs = lvs_ceil_div(2ll * k, tp.slots);       // ~2k candidates: a tight bound at a moderate slab count
    if (want_slabs > cap_slabs) want_slabs = cap_slabs;
    if (make_plan(nq, nb, d, xb_pack, xq_pack, tp.slots, tp.a, tp.slots == LVS2_KCAP, want_slabs, cap_slabs) != LVS_OK)
        return false;
    if ((int64_t)tp.a.nslab * tp.slots > LVS_SELECT_MAX) return false;
    if ((int64_t)tp.a.nslab * tp.slots < (int64_t)k + k / 4) return false;  // corpus too small for a useful bound
    if (make_plan(nq, nb, d, xb_pack, xq_pack, LVS2_KCAP, tp.b, true) != LVS_OK) return false;
    tp.capacity = pow2_ceil(2 * k < 512 ? 512 : 2 * k);
    if (tp.capacity > LVS_SELECT_MAX) tp.capacity = LVS_SELECT_MAX;
    if (tp.capacity < k) return false;
    int64_t off = tp.a.off_pass;  // gtau and the per-slab lists use the phase-A plan's layout
    tp.off_thr = off;
    off += lvs_round_up(nq * 8, 256);
    tp.off_cnt = off;
    off += lvs_round_up(nq * 4, 256);
    tp.off_bucket = off;
    off += lvs_round_up(nq * (int64_t)tp.capacity * 8, 256);
    tp.total = off;
    if (tp.total > (24ll << 30)) return false;  // lists + buckets beyond 24 GB: not worth it, use the selection passes
    return true;
}

// Seed of the small-batch kernel's shared thresholds: gtau[q] = order key of the k-th largest of the nparts values
// seeds[p][q] (the best score of query q over the sample rows of corpus range p; lvs_stream_kernel<.., SEED>), 0 when fewer
// than k ranges saw a row.  One wave per query: 64 values at a time are sorted across the lanes and folded into the running
// top 64 (k <= 56).  Every value is a real row's score, and the k-th largest of a subset never exceeds the k-th largest of
// all rows: a valid lower bound of the final k-th best score, so the seeded search stays exact.
// Banded lists (kc > 0, qn given): the larger of that and (kc-th largest - the query's band) - equally a lower bound of what
// the kernel's own admission rule ends at.
__global__ __launch_bounds__(256) void seed_kth_kernel(const float* __restrict__ seeds, int nparts, long long nq, int k,
                                                       uint32_t* __restrict__ gtau, int kc = 0, const float* __restrict__ qn = nullptr,
                                                       float bscale = 0.f, float bslack = 0.f) {
    const int lane = threadIdx.x & 63;
    const long long q = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= nq) return;
    u64 best = 0;
    for (int p0 = 0; p0 < nparts; p0 += 64) {
        const int p = p0 + lane;
        u64 v = 0;
        if (p < nparts) {
            const float sc = seeds[(long long)p * nq + q];
            if (sc > -INFINITY) v = (u64)lvs_ord32(sc) << 32;  // -inf: that range saw no row
        }
        v = lvs_wave_sort_desc(v, lane);
        if (p0 == 0) {
            best = v;
        } else {
            const u64 w = lvs_shfl_u64(v, 63 - lane);
            best = best > w ? best : w;
            best = lvs_wave_bitonic_merge_desc(best, lane);
        }
    }
    uint32_t kth = (uint32_t)(lvs_shfl_u64(best, k - 1) >> 32);
    if (kc > 0 && qn) {
        const uint32_t kcth = (uint32_t)(lvs_shfl_u64(best, kc - 1) >> 32);
        if (kcth) {
            const uint32_t ob = lvs_ord32(lvs_unord32(kcth) - (bscale * sqrtf(qn[q]) + bslack));
            kth = ob > kth ? ob : kth;
        }
    }
    if (lane == 0) gtau[q] = kth;
}

__global__ __launch_bounds__(256) void fill_f32_kernel(float* __restrict__ dst, long long n, float v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}

__global__ void copy_pass_kernel(const u64* __restrict__ src, long long nq, int kp, u64* __restrict__ dst, int k,
                                 int col0, const uint32_t* __restrict__ pred) {
    if (pred && *pred == 0u) return;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * kp) return;
    long long q = i / kp;
    int j = (int)(i - q * kp);
    if (col0 + j < k) dst[q * k + col0 + j] = src[i];
}
}  // namespace

extern "C" int64_t lvs_flat_search_workspace_bytes(int64_t nq, int64_t nb, int32_t d, int32_t k, int32_t xb_pack,
                                                   int32_t xq_pack) {
    Plan p;
    if (make_plan(nq, nb, d, xb_pack, xq_pack, k, p) != LVS_OK) return LVS_EINVAL;
    TwoPhasePlan tp;
    if (make_two_phase_plan(nq, nb, d, xb_pack, xq_pack, k, tp) && tp.total > p.total) return tp.total;
    return p.total;
}

// ext_seeds (nullable): [ext_rows][nq] scores of real rows of the SEARCHED SET (any shard of it, e.g. the all-gathered
// sample maxima of every corpus shard of a multi-GPU join, lvs_flat_search_seed_scores): the k-th largest per query replaces
// the launch's own sample pass as its starting threshold.
static int32_t flat_search_impl(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack,
                                int64_t nq, int32_t d, int32_t metric, int32_t k, const float* xb_norms_sq,
                                const float* xq_norms_sq, int64_t id_offset, const uint32_t* row_ids,
                                uint64_t* out_keys, void* workspace, int64_t workspace_bytes, void* stream,
                                const float* ext_seeds, int32_t ext_rows) {
    Plan p;
    LVS_REQUIRE(make_plan(nq, nb, d, xb_pack, xq_pack, k, p) == LVS_OK,
                "bad shape nq=%lld nb=%lld d=%d k=%d pack=%d/%d", (long long)nq, (long long)nb, d, k, xb_pack, xq_pack);
    LVS_REQUIRE(metric == LVS_METRIC_IP || metric == LVS_METRIC_L2, "bad metric %d", metric);
    LVS_REQUIRE(k <= LVS_MAX_K, "k=%d exceeds LVS_MAX_K", k);
    LVS_REQUIRE(id_offset >= 0 && id_offset + nb < 0xFFFFFFFFll, "ids must stay below 2^32-1");
    if (nq == 0 || k == 0) return LVS_OK;
    LVS_REQUIRE(out_keys, "out_keys is NULL");
    LVS_DEVICE_GUARD(stream);
    hipStream_t st = (hipStream_t)stream;
    if (nb == 0) {
        LVS_HIP_CHECK(hipMemsetAsync(out_keys, 0, (size_t)nq * k * 8, st));
        return LVS_OK;
    }
    LVS_REQUIRE(xb && xq, "NULL rows");
    LVS_REQUIRE(metric != LVS_METRIC_L2 || (xb_norms_sq && xq_norms_sq), "L2 needs both norm vectors");
    if (workspace_bytes < p.total || !workspace) {
        lvs_set_error("workspace too small: need %lld bytes, got %lld", (long long)p.total, (long long)workspace_bytes);
        return LVS_ENOMEM;
    }
    char* ws = (char*)workspace;
    uint32_t* gtau = (uint32_t*)(ws + p.off_gtau);
    u64* partial = (u64*)(ws + p.off_partial);
    u64* passbuf = (u64*)(ws + p.off_pass);

    auto fill_args = [&](LvsTileArgs& a, const Plan& pl, uint32_t* gtau_, u64* out_) {
        memset(&a, 0, sizeof(a));
        a.xb = xb;
        a.xq = xq;
        a.bn = xb_norms_sq;
        a.qn = xq_norms_sq;
        a.row_ids = row_ids;
        a.gtau = gtau_;
        a.out = out_;
        a.nb = nb;
        a.nq = nq;
        a.ldb = pl.ldb;
        a.ldq = pl.ldq;
        a.nseg = pl.nseg;
        for (int i = 0; i < 3; ++i) {
            a.seg_q[i] = pl.seg_q[i];
            a.seg_c[i] = pl.seg_c[i];
        }
        a.id_offset = id_offset;
        a.nkd = pl.nkd;
        a.nk = pl.nk;
        a.metric = metric;
        a.ntiles = pl.ntiles;
        a.tiles_per_slab = pl.tiles_per_slab;
        a.nslab = pl.nslab;
        a.nqt = pl.nqt;
        a.bq = pl.v2 ? LVS2_BQ : LVS3_BQ;
        a.gq = pl.gq;
        a.lead_slabs = pl.lead_slabs;
        a.debug_hot = (int)lvs_tune("LVS_DEBUG_HOT", 0);
    };

    // ---- k > LVS_KPASS: two corpus passes whatever k is (see make_two_phase_plan).  A bucket overflow (mass ties, or
    // the top k clustered in one slab) sets a device-side flag; the selection passes below then run PREDICATED on that
    // flag (their kernels return at once when it is clear), so the call never synchronises the stream. ----
    const uint32_t* pred = nullptr;
    {
        TwoPhasePlan tp;
        if (make_two_phase_plan(nq, nb, d, xb_pack, xq_pack, k, tp) && tp.total <= workspace_bytes) {
            uint32_t* gtau2 = (uint32_t*)(ws + tp.a.off_gtau);
            u64* lists = (u64*)(ws + tp.a.off_partial);          // [nslab][nq][slots]
            u64* thr = (u64*)(ws + tp.off_thr);                  // [nq]
            uint32_t* cnt = (uint32_t*)(ws + tp.off_cnt);        // [nq]
            uint32_t* flag = (uint32_t*)ws;                      // status word 0: some bucket overflowed
            u64* bucket = (u64*)(ws + tp.off_bucket);            // [nq][capacity]
            LvsTileArgs ta;
            fill_args(ta, tp.a, gtau2, lists);
            ta.k = tp.slots;
            ta.no_share = 1;  // the bound needs every slab's own top list, not lists pruned by other slabs' thresholds
            LVS_HIP_CHECK(hipMemsetAsync(gtau2, 0, (size_t)nq * 4, st));
            LVS_HIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)nq * 4, st));
            LVS_HIP_CHECK(hipMemsetAsync(flag, 0, 4, st));
            {
                ScopedKernelTimer timer(st);
                LVS_HIP_CHECK(lvs_tile_launch(LVS_MODE_TOPK, ta, st));
            }
            // T[q] = k-th largest of the nslab * slots listed keys (element (slab, j) of query q at (slab * nq + q) * slots + j)
            LVS_HIP_CHECK(launch_select(lists, tp.slots, tp.slots, (long long)nq * tp.slots, tp.a.nslab * tp.slots,
                                        nullptr, k, thr, nullptr, 0, nullptr, nq, st));
            LvsTileArgs tb;
            fill_args(tb, tp.b, gtau2, lists);
            tb.k = 1;
            tb.thr_key = thr;
            tb.bucket_count = cnt;
            tb.bucket = bucket;
            tb.bucket_capacity = tp.capacity;
            LVS_HIP_CHECK(lvs_tile_launch(LVS_MODE_COLLECT, tb, st));
            LVS_HIP_CHECK(launch_select(bucket, tp.capacity, tp.capacity, 0, tp.capacity, cnt, k, nullptr,
                                        (u64*)out_keys, k, flag, nq, st));
            pred = flag;
#ifdef LVS_TUNING
            if (lvs_tune_set("LVS_TWO_PHASE_DEBUG")) {  // bucket statistics of this call (synchronises)
                std::vector<uint32_t> h((size_t)nq);
                uint32_t overflowed = 0;
                LVS_HIP_CHECK(hipStreamSynchronize(st));
                LVS_HIP_CHECK(hipMemcpy(&overflowed, flag, 4, hipMemcpyDeviceToHost));
                LVS_HIP_CHECK(hipMemcpy(h.data(), cnt, (size_t)nq * 4, hipMemcpyDeviceToHost));
                uint32_t mn = ~0u, mx = 0;
                double sum = 0;
                for (uint32_t v : h) {
                    mn = v < mn ? v : mn;
                    mx = v > mx ? v : mx;
                    sum += v;
                }
                fprintf(stderr, "[lvs] two-phase k=%d slots=%d slabs=%d capacity=%d: bucket counts min %u mean %.1f max %u overflow=%u\n",
                        k, tp.slots, tp.a.nslab, tp.capacity, mn, sum / (double)nq, mx, overflowed);
            }
#endif
        }
    }

    LvsTileArgs a;
    fill_args(a, p, gtau, partial);
    a.pred = pred;
    a.dbg = nullptr;
    unsigned long long* dbg_counters = nullptr;
#if defined(LVS_COUNT_EVENTS) && defined(LVS_TUNING)
    if (lvs_tune("LVS_COUNT", 0)) {  // tuning build only: count slow-path events of this call (allocates, synchronises)
        LVS_HIP_CHECK(hipMalloc((void**)&dbg_counters, 8 * sizeof(unsigned long long)));
        LVS_HIP_CHECK(hipMemsetAsync(dbg_counters, 0, 8 * sizeof(unsigned long long), st));
        a.dbg = dbg_counters;
    }
#endif

    // 97 .. 256 queries (and up to 4 096 in groups of 256), one K segment of fp16 k-slices: the queries live in REGISTERS
    // (lvs_rq.hip), the corpus streams through LDS once - between the HBM-bound batches below and the joins the list kernel is
    // built for.  (r6) Beyond 4 096 queries the same launch repeats per CHUNK of 4 096 queries (16 groups x 16 corpus ranges):
    // a query belongs to one chunk, so every chunk is a complete search of its own (own seeds, own lists, own merge).
    const bool rq_ok = lvs_tune("LVS_RQ", 1) != 0 && p.nseg == 1 && p.npass == 1 && !pred && g_band.kc == 0;
    // (a corpus SHARD qualifies from 65 536 rows on: its thresholds may come from the caller - the pooled sample scores of all
    // shards, scored by the list kernel's SEED mode in the same arithmetic - or from its own first rows)
    const bool rq_join = rq_ok && nq > LVS_RQ_MAXQ && lvs_tune("LVS_RQ_JOIN", LVS_RQ_JOIN_DEFAULT) != 0 &&
                         lvs_rq_shape_ok(p.dpad, k) && nb >= lvs_tune("LVS_RQ_JOIN_MINROWS", LVS_RQ_JOIN_MINROWS);
    const bool rq_ext_seeds = ext_seeds && ext_rows >= k;
    if (rq_join && rq_ext_seeds) {  // the k-th largest of the caller's sample scores, for every query of the call at once
        hipLaunchKernelGGL(seed_kth_kernel, dim3((unsigned)lvs_ceil_div(nq, 4)), dim3(256), 0, st, ext_seeds, (int)ext_rows,
                           (long long)nq, k, gtau);
        LVS_HIP_CHECK(hipGetLastError());
    }
    if (rq_join || (rq_ok && lvs_rq_fits(nq, nb, p.dpad, k))) {
        // a chunk is 4 096 queries (16 groups x 16 corpus ranges) or 8 192 x 2^i (32 / 64 / 128 / 256 groups x 8 / 4 / 2 / 1 ranges:
        // a multiple of 32 groups - lvs_rq_item); what is left at the end goes out in launches of 8 192, 4 096 and one of <= 4 096 queries
        int64_t chunk = lvs_tune("LVS_RQ_CHUNK", LVS_RQ_CHUNK_DEFAULT);
        {
            int64_t c = LVS_RQ_MAXQ;
            while (2 * c <= chunk && 2 * c <= LVS_RQ_CHUNK_MAX) c *= 2;
            chunk = c;
        }
        for (int64_t c0 = 0, cn = 0; c0 < nq; c0 += cn) {
            const int64_t left = nq - c0;
            cn = left >= chunk ? chunk : (left >= 8192 ? 8192 : (left > LVS_RQ_MAXQ ? LVS_RQ_MAXQ : left));
            LvsRqArgs ra;
            memset(&ra, 0, sizeof(ra));
            ra.xb = xb;
            ra.xq = (const _Float16*)xq + c0 * p.ldq;
            ra.bn = xb_norms_sq;
            ra.qn = xq_norms_sq ? xq_norms_sq + c0 : nullptr;
            ra.row_ids = row_ids;
            ra.gtau = gtau + c0;
            ra.out = partial;
            ra.nb = nb;
            ra.ldb = p.ldb;
            ra.ldq = p.ldq;
            ra.id_offset = id_offset;
            ra.nq = (int)cn;
            ra.k = k;
            ra.metric = metric;
            // thresholds seeded from a sample of the rows scanned by this kernel's own SEED mode (bit-identical scores; see the
            // streaming path below for why a caller's pooled sample scores are not used)
            int64_t sample = nb / 8 / 1024 * 1024;
            // (eight groups or more: a workgroup sees 30 000+ rows and fills its own lists - a quarter of the sample costs the main
            // pass 1 % and saves 3 % of the call; profiles/r09b_rq_groups_sample_probe.log)
            // (chunks of 16 384 queries or more see the corpus in <= 4 long ranges whose lists fill at once: half of that sample again
            // costs the launches nothing and the sample pass half - 100 k x 1 M: call 126.0-127.8 -> 125.4-126.2 ms, same box,
            // profiles/r12d_knobs.log)
            const int64_t sample_cap = lvs_tune("LVS_RQ_SAMPLE", cn >= 16384 ? LVS_RQ_SEED_ROWS / 8
                                                                 : (cn > 7 * LVS_RQ_GROUPQ ? LVS_RQ_SEED_ROWS / 4 : LVS_RQ_SEED_ROWS));
            if (sample > sample_cap) sample = sample_cap;
            if (rq_join && rq_ext_seeds) {
                // seeded above
            } else if (sample >= 4096 && lvs_tune("LVS_STREAM_SEED", 1) != 0) {
                // (the k-th largest of one maximum per corpus range needs >= k ranges: the sample pass keeps launches of <= 16 groups)
                float* seeds = (float*)((char*)partial + lvs_rq_parts_bytes(cn, k));  // [ranges][<= 4 096]
                for (int64_t s0 = 0; s0 < cn; s0 += LVS_RQ_MAXQ) {
                    const int64_t sn = cn - s0 < LVS_RQ_MAXQ ? cn - s0 : LVS_RQ_MAXQ;
                    LvsRqArgs rs = ra;
                    rs.xq = (const _Float16*)ra.xq + s0 * p.ldq;
                    rs.qn = ra.qn ? ra.qn + s0 : nullptr;
                    rs.gtau = ra.gtau + s0;
                    rs.nq = (int)sn;
                    rs.nb = sample;
                    rs.seed_out = seeds;
                    LVS_HIP_CHECK(lvs_rq_launch(rs, p.dpad, st));
                    hipLaunchKernelGGL(seed_kth_kernel, dim3((unsigned)lvs_ceil_div(sn, 4)), dim3(256), 0, st, (const float*)seeds,
                                       rs.nparts, (long long)sn, k, gtau + c0 + s0);  // writes every gtau[q] of the piece
                    LVS_HIP_CHECK(hipGetLastError());
                }
            } else {
                LVS_HIP_CHECK(hipMemsetAsync(gtau + c0, 0, (size_t)cn * 4, st));
            }
            // (r6) one wave per SIMD, 64 queries per wave (lvs_rj.hip) for the launches it takes: whole 32-row blocks there, the
            // corpus' last nb % 32 rows through lvs_rq_kernel as one more list per query
            if (lvs_tune("LVS_RJ", LVS_RJ_DEFAULT) != 0 && lvs_rj_fits(cn, nb, p.dpad, k, row_ids != nullptr)) {
                {
                    ScopedKernelTimer timer(st, LVS_KERNEL_RJ, c0 > 0);
                    LVS_HIP_CHECK(lvs_rj_launch(ra, p.dpad, st));
                }
                const int64_t nb_full = nb / 32 * 32;
                if (nb_full < nb) {
                    LvsRqArgs rt = ra;
                    rt.xb = (const char*)xb + nb_full * p.ldb * 2;
                    rt.bn = xb_norms_sq ? xb_norms_sq + nb_full : nullptr;
                    rt.nb = nb - nb_full;
                    rt.id_offset = id_offset + nb_full;
                    rt.out = partial + (int64_t)ra.nparts * cn * k;
                    LVS_HIP_CHECK(lvs_rq_launch(rt, p.dpad, st));
                    ra.nparts += rt.nparts;
                }
            } else {
                ScopedKernelTimer timer(st, LVS_KERNEL_RQ, c0 > 0);
                LVS_HIP_CHECK(lvs_rq_launch(ra, p.dpad, st));
            }
            u64* okeys = (u64*)out_keys + c0 * k;
            if (ra.nparts >= 16)
                hipLaunchKernelGGL(merge_keys_wide_kernel, dim3((unsigned)cn), dim3(1024), 0, st, partial, ra.nparts, (long long)cn, k,
                                   okeys, (long long)k);
            else
                hipLaunchKernelGGL(merge_keys_kernel, dim3((unsigned)lvs_ceil_div(cn, 4)), dim3(256), 0, st, partial, ra.nparts,
                                   (long long)cn, k, okeys, (long long)k, (const uint32_t*)nullptr);
            LVS_HIP_CHECK(hipGetLastError());
        }
        return LVS_OK;
    }
    // HBM-bound regime (the literal sem_search: one query per call; small batches up to 256 queries): stream the corpus
    // once, queries resident in LDS
    {
        const int nqseg = (xq_pack == LVS_PACK_SPLIT && !g_hi_only) ? 2 : 1;
        const int jper = p.dpad / 16;
        int kcap = 0, nqb = 0, groups = 0;
        const bool fits = lvs_stream_plan(nq, k, nqseg * jper, &kcap, &nqb, &groups) > 0;
        const bool want = lvs_tune("LVS_STREAM", 1) != 0 && nq <= lvs_tune("LVS_STREAM_MAXQ_RT", LVS_STREAM_MAXQ);
        // several sibling workgroups per corpus range (beyond 96 fp16 queries) leave the HBM-bound regime - the siblings'
        // re-reads are bound by the fabric behind the L2 (0.45 / 0.70 ms at 128 / 256 queries x 1 M rows) - while the seeded
        // list kernel runs such batches near its MFMA rate: the stream kernel keeps the single-group batches
        const bool beyond = groups > lvs_tune("LVS_STREAM_MAXG", 1) && p.npass == 1 && tile_seed_tiles(nq, nb, k) > 0;
        if (want && fits && nb >= 4096 && !beyond) {
            LvsStreamArgs sa;
            memset(&sa, 0, sizeof(sa));
            sa.xb = xb;
            sa.xq = xq;
            sa.bn = xb_norms_sq;
            sa.qn = xq_norms_sq;
            sa.row_ids = row_ids;
            sa.gtau = gtau;
            sa.out = partial;
            sa.nb = nb;
            sa.ldb = p.ldb;
            sa.ldq = p.ldq;
            sa.id_offset = id_offset;
            sa.nq = (int)nq;
            sa.k = k;
            sa.metric = metric;
            sa.jper = jper;
            sa.nseg = p.nseg;
            sa.nj = p.nseg * jper;
            sa.nbfrag = nqseg * jper;
            sa.kcap = kcap;
            sa.nqb = nqb;
            sa.groups = groups;
            for (int i = 0; i < 3; ++i) {
                sa.seg_q[i] = p.seg_q[i];
                sa.seg_c[i] = p.seg_c[i];
                sa.seg_b[i] = p.seg_q[i] == 0 ? 0 : jper;
            }
            // Every workgroup scans its own contiguous range, all at the same time: without help each of them starts with
            // empty lists and pays its own cold start (~k (1 + ln(range / k)) insertions per query and workgroup - with
            // dozens of queries that, not HBM, set the time in round 2).  So with several queries the thresholds are
            // SEEDED: the same kernel first scans a sample (the first nb / 8 rows, at most 32 768) in SEED mode - no lists,
            // every workgroup just keeps the best score per query over its 4 row blocks - and one wave per query takes the
            // k-th largest of those per-range maxima as the starting threshold (~15 + 5 us; round 3's first version scored the
            // sample into a matrix with the tile kernel and radix-selected it: 130-180 us).  A workgroup of the main scan then
            // only inserts rows that beat it (~k * nb / sample per query over the WHOLE launch).  Exact: every value is a
            // real row's score and the k-th largest of a subset never exceeds the k-th largest of all rows; the sample
            // rows themselves are scanned again with the rest.
            int64_t sample = nb / 8 / 1024 * 1024;
            if (sample > LVS_STREAM_SEED_ROWS) sample = LVS_STREAM_SEED_ROWS;
            // A caller's pooled sample scores (ext_seeds) are NOT used here: they come from the tile kernel's SEED mode, whose
            // accumulation order differs from this kernel's, and "the k-th largest of a subset is a lower bound" is only exact
            // when the threshold and the scan see bit-identical scores - one ulp the wrong way could reject a true top-k row on
            // its own shard.  This path seeds itself, with its own arithmetic.
            if (nq >= lvs_tune("LVS_STREAM_SEED_MINQ", 2) && sample >= 4096 && lvs_tune("LVS_STREAM_SEED", 1) != 0) {
                float* seeds = (float*)((char*)partial + lvs_stream_parts_bytes(nq, k));  // [ranges][nq]
                LvsStreamArgs ss = sa;
                ss.nb = sample;
                ss.seed_out = seeds;
                LVS_HIP_CHECK(lvs_stream_launch(ss, st));
                hipLaunchKernelGGL(seed_kth_kernel, dim3((unsigned)lvs_ceil_div(nq, 4)), dim3(256), 0, st, (const float*)seeds,
                                   ss.nparts, (long long)nq, k, gtau);  // writes every gtau[q]
                LVS_HIP_CHECK(hipGetLastError());
            } else {
                LVS_HIP_CHECK(hipMemsetAsync(gtau, 0, (size_t)nq * 4, st));
            }
            {
                ScopedKernelTimer timer(st, LVS_KERNEL_STREAM);
                LVS_HIP_CHECK(lvs_stream_launch(sa, st));
            }
            if (sa.nparts >= 16 && k <= 64)
                hipLaunchKernelGGL(merge_keys_wide_kernel, dim3((unsigned)nq), dim3(1024), 0, st, partial, sa.nparts,
                                   (long long)nq, k, (u64*)out_keys, (long long)k);
            else
                hipLaunchKernelGGL(merge_keys_kernel, dim3((unsigned)lvs_ceil_div(nq, 4)), dim3(256), 0, st, partial,
                                   sa.nparts, (long long)nq, k, (u64*)out_keys, (long long)k, (const uint32_t*)nullptr);
            LVS_HIP_CHECK(hipGetLastError());
            return LVS_OK;
        }
    }
    // k == 1: the list-free per-lane running best (TOP1) wins on short streams (k-means assignment: a few corpus tiles
    // per query, nearly every block improves some lane's best); on long streams the threshold filter of the list
    // kernel skips almost every block, so k = 1 goes through it like any other k.  Same winner either way
    // (best score, lowest row among equals).
    bool use_top1 = k == 1 && !row_ids && p.tiles_per_slab <= 64;
    if (lvs_tune_set("LVS_TOP1")) use_top1 = k == 1 && !row_ids && lvs_tune("LVS_TOP1", 0) != 0;  // -DLVS_TUNING only
    for (int pass = 0; pass < p.npass; ++pass) {
        const int col0 = pass * p.kpass;
        const int kp = (k - col0) < p.kpass ? (k - col0) : p.kpass;
        a.k = kp;
        // pass > 0: only keys strictly below the last key of the previous pass take part
        a.ub = pass == 0 ? nullptr : (const u64*)out_keys + (col0 - 1);
        a.ub_stride = k;
        const bool banded = g_band.kc > 0 && g_band.kc < kp && p.npass == 1 && !pred && !use_top1 && xq_norms_sq;
        a.kc = banded ? g_band.kc : 0;
        a.bscale = banded ? g_band.bscale : 0.f;
        a.bslack = banded ? g_band.bslack : 0.f;
        const bool seedable = pass == 0 && p.npass == 1 && !pred && !use_top1;
        const int seed_tiles = seedable ? tile_seed_tiles(nq, nb, kp) : 0;
        if (seedable && ext_seeds && ext_rows >= kp) {  // the caller's pooled sample scores (every shard's sample)
            hipLaunchKernelGGL(seed_kth_kernel, dim3((unsigned)lvs_ceil_div(nq, 4)), dim3(256), 0, st, ext_seeds, (int)ext_rows,
                               (long long)nq, kp, gtau, a.kc, xq_norms_sq, a.bscale, a.bslack);  // writes every gtau[q]
            LVS_HIP_CHECK(hipGetLastError());
        } else if (seed_tiles > 0) {  // thresholds seeded from a sample instead of cold starts (tile_seed_tiles)
            float* seeds = (float*)(ws + p.off_seed);
            LvsTileArgs sd = a;
            sd.nb = (int64_t)seed_tiles * LVS_BC;
            sd.ntiles = sd.nslab = seed_tiles;
            sd.tiles_per_slab = 1;
            sd.nqt = (int)lvs_ceil_div(nq, LVS2_BQ);
            sd.bq = LVS2_BQ;
            sd.gq = 1;
            sd.lead_slabs = 0;
            sd.k = 1;
            sd.kc = 0;
            sd.ub = nullptr;
            sd.seed_out = seeds;
            LVS_HIP_CHECK(lvs_tile_launch(LVS_MODE_SEED, sd, st));
            hipLaunchKernelGGL(seed_kth_kernel, dim3((unsigned)lvs_ceil_div(nq, 4)), dim3(256), 0, st, (const float*)seeds,
                               sd.nslab, (long long)nq, kp, gtau, a.kc, xq_norms_sq, a.bscale, a.bslack);  // writes every gtau[q]
            LVS_HIP_CHECK(hipGetLastError());
        } else {
            LVS_HIP_CHECK(hipMemsetAsync(gtau, 0, (size_t)nq * 4, st));
        }
        if (pred) {  // predicated fallback pass: not a measurement of the dominant kernel
            LVS_HIP_CHECK(lvs_tile_launch(LVS_MODE_TOPK, a, st));
        } else {
            ScopedKernelTimer timer(st);
            LVS_HIP_CHECK(lvs_tile_launch(use_top1 ? LVS_MODE_TOP1 : LVS_MODE_TOPK, a, st));
        }
        dim3 mgrid((unsigned)lvs_ceil_div(nq, 4)), mblock(256);
        if (p.npass == 1 && kp == 1) {
            hipLaunchKernelGGL(merge_top1_kernel, dim3((unsigned)lvs_ceil_div(nq, 256)), mblock, 0, st, partial,
                               p.nslab, (long long)nq, (u64*)out_keys, (long long)k);
        } else if (p.npass == 1 && !pred && p.nslab >= 16 && nq <= 4096) {
            // few queries, many slabs: one workgroup of 16 waves per query (one wave per query walks hundreds of partial
            // lists as a chain of dependent loads: 150 us of a 0.5 ms call at 256 queries x 489 slabs)
            hipLaunchKernelGGL(merge_keys_wide_kernel, dim3((unsigned)nq), dim3(1024), 0, st, partial, p.nslab, (long long)nq,
                               kp, (u64*)out_keys, (long long)k);
        } else if (p.npass == 1) {
            hipLaunchKernelGGL(merge_keys_kernel, mgrid, mblock, 0, st, partial, p.nslab, (long long)nq, kp,
                               (u64*)out_keys, (long long)k, pred);
        } else {
            // the upper bounds of this pass live in out_keys, so merge into a side buffer first
            hipLaunchKernelGGL(merge_keys_kernel, mgrid, mblock, 0, st, partial, p.nslab, (long long)nq, kp, passbuf,
                               (long long)kp, pred);
            hipLaunchKernelGGL(copy_pass_kernel, dim3((unsigned)lvs_ceil_div(nq * kp, 256)), dim3(256), 0, st, passbuf,
                               (long long)nq, kp, (u64*)out_keys, k, col0, pred);
        }
        LVS_HIP_CHECK(hipGetLastError());
    }
    if (dbg_counters) {
        unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        LVS_HIP_CHECK(hipStreamSynchronize(st));
        LVS_HIP_CHECK(hipMemcpy(h, dbg_counters, sizeof(h), hipMemcpyDeviceToHost));
        (void)hipFree(dbg_counters);
        fprintf(stderr,
                "[lvs] nq=%lld nb=%lld k=%d nslab=%d: wave-tiles %llu, block visits %llu, insertions %llu (%.1f per query); "
                "cycles per wave-tile: filter %.0f, visit loop %.0f (of which insertions %.0f); per visit %.0f, per insertion %.0f\n",
                (long long)nq, (long long)nb, k, p.nslab, h[2], h[0], h[1], (double)h[1] / (double)(nq > 0 ? nq : 1),
                (double)h[3] / (double)(h[2] ? h[2] : 1), (double)h[4] / (double)(h[2] ? h[2] : 1),
                (double)h[5] / (double)(h[2] ? h[2] : 1), (double)(h[4] - h[5]) / (double)(h[0] ? h[0] : 1),
                (double)h[5] / (double)(h[1] ? h[1] : 1));
    }
    return LVS_OK;
}

extern "C" int32_t lvs_flat_search_keys(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack,
                                        int64_t nq, int32_t d, int32_t metric, int32_t k, const float* xb_norms_sq,
                                        const float* xq_norms_sq, int64_t id_offset, const uint32_t* row_ids,
                                        uint64_t* out_keys, void* workspace, int64_t workspace_bytes, void* stream) {
    return flat_search_impl(xb, xb_pack, nb, xq, xq_pack, nq, d, metric, k, xb_norms_sq, xq_norms_sq, id_offset, row_ids,
                            out_keys, workspace, workspace_bytes, stream, nullptr, 0);
}

extern "C" int32_t lvs_flat_search_keys_seeded(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack,
                                               int64_t nq, int32_t d, int32_t metric, int32_t k, const float* xb_norms_sq,
                                               const float* xq_norms_sq, int64_t id_offset, const uint32_t* row_ids,
                                               const float* seed_scores, int32_t seed_rows, uint64_t* out_keys,
                                               void* workspace, int64_t workspace_bytes, void* stream) {
    LVS_REQUIRE(seed_rows >= 0 && (seed_rows == 0 || seed_scores), "bad seed scores");
    return flat_search_impl(xb, xb_pack, nb, xq, xq_pack, nq, d, metric, k, xb_norms_sq, xq_norms_sq, id_offset, row_ids,
                            out_keys, workspace, workspace_bytes, stream, seed_rows > 0 ? seed_scores : nullptr, seed_rows);
}

extern "C" int32_t lvs_flat_search_seed_tiles(int64_t nq, int64_t nb, int32_t k) {
    if (nq < 0 || nb < 0 || k < 1) return LVS_EINVAL;
    int tiles = tile_seed_tiles(nq, nb, k);
    // A shard of a sharded join pools its sample with the other shards', so a larger sample per shard pays: 100 k x 8 x 125 k
    // rows, same box, sample pass + search per shard (profiles/r05y_seed_pool_sweep.log): 10 tiles 0.42 + 19.94 ms, 20 tiles
    // 0.76 + 19.06, 40 tiles 1.51 + 18.22, 61 tiles 2.29 + 17.71 - a twenty-fourth of the shard, at most 24 tiles.
    if (tiles > 0 && lvs_ceil_div(nq, LVS2_BQ) > 64) {
        int64_t t = nb / LVS_BC / 24;
        t = t > 24 ? 24 : t;
        if (t > tiles) tiles = (int)t;
    }
    return tiles;
}

extern "C" int32_t lvs_flat_search_seed_scores(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack,
                                               int64_t nq, int32_t d, int32_t metric, const float* xb_norms_sq,
                                               const float* xq_norms_sq, int32_t tiles, float* out_scores, void* stream) {
    Plan p;
    LVS_REQUIRE(make_plan(nq, nb, d, xb_pack, xq_pack, 1, p, true) == LVS_OK, "bad shape nq=%lld nb=%lld d=%d pack=%d/%d",
                (long long)nq, (long long)nb, d, xb_pack, xq_pack);
    LVS_REQUIRE(metric == LVS_METRIC_IP || metric == LVS_METRIC_L2, "bad metric %d", metric);
    LVS_REQUIRE(tiles >= 0, "bad tile count %d", tiles);
    if (nq == 0 || tiles == 0) return LVS_OK;
    LVS_REQUIRE(out_scores, "out_scores is NULL");
    LVS_DEVICE_GUARD(stream);
    hipStream_t st = (hipStream_t)stream;
    const int have = (int)(nb / LVS_BC < tiles ? nb / LVS_BC : tiles);  // whole tiles of real rows only
    if (have < tiles) {  // -inf: "that tile saw no row" (seed_kth_kernel skips it)
        const long long cnt = (long long)(tiles - have) * nq;
        float* dst = out_scores + (long long)have * nq;
        hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)lvs_ceil_div(cnt, 256)), dim3(256), 0, st, dst, cnt, -INFINITY);
        LVS_HIP_CHECK(hipGetLastError());
    }
    if (have == 0) return LVS_OK;  // an empty or shorter-than-a-tile shard (its buffers may be NULL) still fills its block
    LVS_REQUIRE(xb && xq, "NULL rows");
    LVS_REQUIRE(metric != LVS_METRIC_L2 || (xb_norms_sq && xq_norms_sq), "L2 needs both norm vectors");
    LvsTileArgs a;
    memset(&a, 0, sizeof(a));
    a.xb = xb;
    a.xq = xq;
    a.bn = xb_norms_sq;
    a.qn = xq_norms_sq;
    a.nb = (int64_t)have * LVS_BC;
    a.nq = nq;
    a.ldb = p.ldb;
    a.ldq = p.ldq;
    a.nseg = p.nseg;
    for (int i = 0; i < 3; ++i) {
        a.seg_q[i] = p.seg_q[i];
        a.seg_c[i] = p.seg_c[i];
    }
    a.nkd = p.nkd;
    a.nk = p.nk;
    a.metric = metric;
    a.k = 1;
    a.ntiles = a.nslab = have;
    a.tiles_per_slab = 1;
    a.nqt = (int)lvs_ceil_div(nq, LVS2_BQ);
    a.bq = LVS2_BQ;
    a.gq = 1;
    a.seed_out = out_scores;
    LVS_HIP_CHECK(lvs_tile_launch(LVS_MODE_SEED, a, st));
    return LVS_OK;
}

extern "C" int64_t lvs_nearest_hi_workspace_bytes(int64_t nq, int64_t nb, int32_t d) {
    Plan p;
    if (make_plan(nq, nb, d, LVS_PACK_F16, LVS_PACK_F16, 1, p, true) != LVS_OK) return LVS_EINVAL;
    return 256 + lvs_round_up((int64_t)p.nslab * nq * 8, 256) + lvs_round_up((int64_t)p.nslab * nq * 4, 256);
}

extern "C" int32_t lvs_nearest_hi(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack, int64_t nq,
                                  int32_t d, int32_t metric, const float* xb_norms_sq, const float* xq_norms_sq,
                                  int64_t id_offset, uint64_t* out_keys, float* out_second, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    Plan p;
    LVS_REQUIRE(make_plan(nq, nb, d, xb_pack, xq_pack, 1, p, true) == LVS_OK, "bad shape nq=%lld nb=%lld d=%d pack=%d/%d",
                (long long)nq, (long long)nb, d, xb_pack, xq_pack);
    LVS_REQUIRE(metric == LVS_METRIC_IP || metric == LVS_METRIC_L2, "bad metric %d", metric);
    LVS_REQUIRE(id_offset >= 0 && id_offset + nb < 0xFFFFFFFFll, "ids must stay below 2^32-1");
    if (nq == 0) return LVS_OK;
    LVS_REQUIRE(out_keys && out_second, "NULL output");
    LVS_REQUIRE(nb > 0 && xb && xq, "NULL rows / empty corpus");
    LVS_REQUIRE(metric != LVS_METRIC_L2 || (xb_norms_sq && xq_norms_sq), "L2 needs both norm vectors");
    const int64_t need = lvs_nearest_hi_workspace_bytes(nq, nb, d);
    if (!workspace || workspace_bytes < need) {
        lvs_set_error("workspace too small: need %lld bytes, got %lld", (long long)need, (long long)workspace_bytes);
        return LVS_ENOMEM;
    }
    LVS_DEVICE_GUARD(stream);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace + 256;
    u64* pk = (u64*)ws;
    float* ps = (float*)(ws + lvs_round_up((int64_t)p.nslab * nq * 8, 256));
    LvsTileArgs a;
    memset(&a, 0, sizeof(a));
    a.xb = xb;
    a.xq = xq;
    a.bn = xb_norms_sq;
    a.qn = xq_norms_sq;
    a.nb = nb;
    a.nq = nq;
    a.ldb = p.ldb;  // SPLIT rows keep their leading dimension; only the fp16 "hi" half (columns [0, dpad)) is read
    a.ldq = p.ldq;
    a.nseg = 1;
    a.id_offset = id_offset;
    a.nkd = p.nkd;
    a.nk = p.nkd;
    a.metric = metric;
    a.k = 1;
    a.ntiles = p.ntiles;
    a.tiles_per_slab = p.tiles_per_slab;
    a.nslab = p.nslab;
    a.nqt = p.nqt;
    a.bq = LVS2_BQ;
    a.gq = p.gq;
    a.lead_slabs = p.lead_slabs;
    a.out = p.nslab == 1 ? (u64*)out_keys : pk;
    a.out_second = p.nslab == 1 ? out_second : ps;
    {
        ScopedKernelTimer timer(st);
        LVS_HIP_CHECK(lvs_tile_launch(LVS_MODE_TOP2, a, st));
    }
    if (p.nslab > 1) {
        hipLaunchKernelGGL(merge_top2_kernel, dim3((unsigned)lvs_ceil_div(nq, 256)), dim3(256), 0, st, pk, ps, p.nslab,
                           (long long)nq, (u64*)out_keys, out_second);
        LVS_HIP_CHECK(hipGetLastError());
    }
    return LVS_OK;
}

extern "C" int32_t lvs_rescore_keys(const void* xb, int32_t xb_pack, const void* xq, int32_t xq_pack, int64_t nq, int32_t d,
                                    int32_t metric, const float* xb_norms_sq, const float* xq_norms_sq, int64_t id_offset,
                                    int32_t k, uint64_t* keys, void* stream) {
    LVS_REQUIRE(nq >= 0 && d > 0 && k >= 0, "bad shape");
    if (k == 0) return LVS_OK;
    LVS_REQUIRE(metric == LVS_METRIC_IP || metric == LVS_METRIC_L2, "bad metric %d", metric);
    LVS_REQUIRE((xb_pack == LVS_PACK_F16 || xb_pack == LVS_PACK_SPLIT) && (xq_pack == LVS_PACK_F16 || xq_pack == LVS_PACK_SPLIT),
                "bad pack mode %d/%d", xb_pack, xq_pack);
    if (nq == 0) return LVS_OK;
    LVS_REQUIRE(xb && xq && keys, "NULL buffer");
    LVS_REQUIRE(metric != LVS_METRIC_L2 || (xb_norms_sq && xq_norms_sq), "L2 needs both norm vectors");
    LVS_DEVICE_GUARD(stream);
    const int dpad = (int)lvs_round_up(d, LVS_BK);
    const long long ldb = xb_pack == LVS_PACK_SPLIT ? 2 * dpad : dpad, ldq = xq_pack == LVS_PACK_SPLIT ? 2 * dpad : dpad;
    const dim3 grid((unsigned)lvs_ceil_div(nq * k, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define LVS_RESCORE(QS, BS)                                                                                              \
    hipLaunchKernelGGL((rescore_keys_kernel<QS, BS>), grid, block, 0, st, (const _Float16*)xb, ldb, (const _Float16*)xq, ldq, \
                       dpad, metric, xb_norms_sq, xq_norms_sq, (long long)id_offset, (long long)nq, (int)k, (u64*)keys)
    if (xq_pack == LVS_PACK_SPLIT && xb_pack == LVS_PACK_SPLIT) LVS_RESCORE(1, 1);
    else if (xq_pack == LVS_PACK_SPLIT) LVS_RESCORE(1, 0);
    else if (xb_pack == LVS_PACK_SPLIT) LVS_RESCORE(0, 1);
    else LVS_RESCORE(0, 0);
#undef LVS_RESCORE
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_flat_search_keys_hi(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack,
                                           int64_t nq, int32_t d, int32_t metric, int32_t k, const float* xb_norms_sq,
                                           const float* xq_norms_sq, int64_t id_offset, const uint32_t* row_ids,
                                           uint64_t* out_keys, void* workspace, int64_t workspace_bytes, void* stream) {
    HiOnlyScope hi;
    return lvs_flat_search_keys(xb, xb_pack, nb, xq, xq_pack, nq, d, metric, k, xb_norms_sq, xq_norms_sq, id_offset, row_ids,
                                out_keys, workspace, workspace_bytes, stream);
}

extern "C" int32_t lvs_flat_search_keys_hi_banded(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack,
                                                  int64_t nq, int32_t d, int32_t metric, int32_t k, int32_t kc,
                                                  float band_scale, float band_slack, const float* xb_norms_sq,
                                                  const float* xq_norms_sq, int64_t id_offset, const uint32_t* row_ids,
                                                  uint64_t* out_keys, void* workspace, int64_t workspace_bytes,
                                                  void* stream) {
    LVS_REQUIRE(kc >= 1 && kc <= k, "banded search: 1 <= kc <= k (got kc=%d, k=%d)", kc, k);
    LVS_REQUIRE(band_scale >= 0.f && band_slack >= 0.f, "negative band");
    LVS_REQUIRE(nq == 0 || xq_norms_sq, "banded search needs the query norms");
    HiOnlyScope hi;
    BandScope band(kc, band_scale, band_slack);
    return lvs_flat_search_keys(xb, xb_pack, nb, xq, xq_pack, nq, d, metric, k, xb_norms_sq, xq_norms_sq, id_offset, row_ids,
                                out_keys, workspace, workspace_bytes, stream);
}

extern "C" int32_t lvs_sort_keys_desc(uint64_t* keys, int64_t nq, int32_t k, void* stream) {
    LVS_REQUIRE(nq >= 0 && k >= 0 && k <= 64, "lvs_sort_keys_desc: k must be at most 64 (got %d)", k);
    if (nq == 0 || k <= 1) return LVS_OK;
    LVS_REQUIRE(keys, "NULL buffer");
    LVS_DEVICE_GUARD(stream);
    hipLaunchKernelGGL(sort_keys_kernel, dim3((unsigned)lvs_ceil_div(nq, 4)), dim3(256), 0, (hipStream_t)stream, (u64*)keys,
                       (long long)nq, (int)k);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_certify_topk_banded(const uint64_t* approx_keys, const uint64_t* exact_keys, const float* q_norms_sq,
                                           int64_t nq, int32_t k1, int32_t k, float scale, float slack, float band,
                                           int64_t* out_idx, uint64_t* out_count, void* stream) {
    LVS_REQUIRE(nq >= 0 && k >= 1 && k1 >= k && scale >= 0.f && slack >= 0.f && band >= 0.f, "bad arguments");
    if (nq == 0) return LVS_OK;
    LVS_REQUIRE(approx_keys && exact_keys && out_idx && out_count, "NULL buffer");
    LVS_DEVICE_GUARD(stream);
    hipLaunchKernelGGL(certify_topk_kernel, dim3((unsigned)lvs_ceil_div(nq, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const u64*)approx_keys, (const u64*)exact_keys, q_norms_sq, (long long)nq, (int)k1, (int)k, scale, slack,
                       band, (long long*)out_idx, (unsigned long long*)out_count);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_certify_topk(const uint64_t* approx_keys, const uint64_t* exact_keys, const float* q_norms_sq,
                                    int64_t nq, int32_t k1, int32_t k, float scale, float slack, int64_t* out_idx,
                                    uint64_t* out_count, void* stream) {
    return lvs_certify_topk_banded(approx_keys, exact_keys, q_norms_sq, nq, k1, k, scale, slack, 0.f, out_idx, out_count, stream);
}

static int32_t margin_select_launch(const uint64_t* keys, const float* second, const float* q_norms_sq, int64_t nq, float scale,
                                    float slack, const float* stats, const MarginCoef& coef, int64_t* out_idx,
                                    uint64_t* out_count, void* stream) {
    if (nq == 0) return LVS_OK;
    LVS_REQUIRE(keys && second && out_idx && out_count, "NULL buffer");
    LVS_DEVICE_GUARD(stream);
    // at most 2048 workgroups, each at least 1024 queries
    long long per_block = lvs_ceil_div(nq, 2048);
    per_block = lvs_round_up(per_block < 1024 ? 1024 : per_block, 256);
    hipLaunchKernelGGL(margin_select_kernel, dim3((unsigned)lvs_ceil_div(nq, per_block)), dim3(256), 0, (hipStream_t)stream,
                       (const u64*)keys, second, q_norms_sq, (long long)nq, scale, slack, stats, coef, per_block,
                       (long long*)out_idx, (unsigned long long*)out_count);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_margin_select(const uint64_t* keys, const float* second, const float* q_norms_sq, int64_t nq,
                                     float scale, float slack, int64_t* out_idx, uint64_t* out_count, void* stream) {
    LVS_REQUIRE(nq >= 0 && scale >= 0.f && slack >= 0.f, "bad arguments");
    return margin_select_launch(keys, second, q_norms_sq, nq, scale, slack, nullptr, MarginCoef{{0, 0, 0, 0, 0}}, out_idx,
                                out_count, stream);
}

extern "C" int32_t lvs_margin_select_stats(const uint64_t* keys, const float* second, const float* q_norms_sq, int64_t nq,
                                           const float* corpus_stats, const float* coef5, int64_t* out_idx,
                                           uint64_t* out_count, void* stream) {
    LVS_REQUIRE(nq >= 0 && corpus_stats && coef5, "bad arguments");
    MarginCoef c;
    for (int i = 0; i < 5; ++i) {
        LVS_REQUIRE(coef5[i] >= 0.f, "negative coefficient");
        c.c[i] = coef5[i];
    }
    return margin_select_launch(keys, second, q_norms_sq, nq, 0.f, 0.f, corpus_stats, c, out_idx, out_count, stream);
}

extern "C" int32_t lvs_merge_keys(const uint64_t* parts, int32_t nparts, int64_t nq, int32_t k, uint64_t* out_keys,
                                  void* stream) {
    LVS_REQUIRE(nparts >= 1 && nq >= 0 && k >= 0, "bad arguments");
    if (nq == 0 || k == 0) return LVS_OK;
    LVS_REQUIRE(parts && out_keys, "NULL buffer");
    LVS_REQUIRE(k <= LVS_MAX_K, "k=%d exceeds LVS_MAX_K", k);
    LVS_DEVICE_GUARD(stream);
    hipStream_t st = (hipStream_t)stream;
    if (k > 64) {
        // long lists: sort the candidates of a query in LDS, at most LVS_SELECT_MAX at a time.  More than that (e.g. 8
        // shards x k = 1000) is folded in rounds: out = top k of (out U the next group of parts); one workgroup owns a
        // query and reads everything it needs before it writes, so `out` is updated in place and no scratch is needed.
        const int per_round0 = LVS_SELECT_MAX / k;            // parts in the first round (>= 2 because k <= 2048)
        const int per_round = per_round0 - 1;                 // later rounds: one slot goes to the running result
        int done = nparts < per_round0 ? nparts : per_round0;
        LVS_HIP_CHECK(launch_merge_long((const u64*)parts, 0, done, nullptr, nq, k, (u64*)out_keys, st));
        while (done < nparts) {
            const int take = nparts - done < per_round ? nparts - done : per_round;
            LVS_HIP_CHECK(launch_merge_long((const u64*)parts, done, take, (const u64*)out_keys, nq, k, (u64*)out_keys, st));
            done += take;
        }
        return LVS_OK;
    }
    hipLaunchKernelGGL(merge_keys_kernel, dim3((unsigned)lvs_ceil_div(nq, 4)), dim3(256), 0, st, (const u64*)parts, nparts,
                       (long long)nq, k, (u64*)out_keys, (long long)k, (const uint32_t*)nullptr);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_keys_to_result(const uint64_t* keys, int64_t nq, int32_t k, int32_t metric,
                                      const int64_t* id_map, int32_t score_exp, float* out_D, int64_t* out_I, void* stream) {
    LVS_REQUIRE(nq >= 0 && k >= 0, "bad shape");
    LVS_REQUIRE(score_exp >= -120 && score_exp <= 120, "score_exp %d out of range", score_exp);
    LVS_REQUIRE(metric == LVS_METRIC_IP || metric == LVS_METRIC_L2, "bad metric %d", metric);
    long long n = (long long)nq * k;
    if (n == 0) return LVS_OK;
    LVS_REQUIRE(keys && out_D && out_I, "NULL buffer");
    LVS_DEVICE_GUARD(stream);
    hipLaunchKernelGGL(keys_to_result_kernel, dim3((unsigned)lvs_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const u64*)keys, n, metric, (const long long*)id_map, ldexpf(1.0f, -score_exp), out_D,
                       (long long*)out_I);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_scores(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack, int64_t nq,
                              int32_t d, int32_t metric, const float* xb_norms_sq, const float* xq_norms_sq,
                              int32_t score_exp, float* out, int64_t ld_out, void* stream) {
    Plan p;
    LVS_REQUIRE(score_exp >= -120 && score_exp <= 120, "score_exp %d out of range", score_exp);
    LVS_REQUIRE(make_plan(nq, nb, d, xb_pack, xq_pack, 1, p, true) == LVS_OK, "bad shape");
    LVS_REQUIRE(metric == LVS_METRIC_IP || metric == LVS_METRIC_L2, "bad metric %d", metric);
    if (nq == 0 || nb == 0) return LVS_OK;
    LVS_REQUIRE(xb && xq && out && ld_out >= nb, "bad buffers");
    LVS_REQUIRE(metric != LVS_METRIC_L2 || (xb_norms_sq && xq_norms_sq), "L2 needs both norm vectors");
    LVS_DEVICE_GUARD(stream);
    LvsTileArgs a;
    memset(&a, 0, sizeof(a));
    a.xb = xb;
    a.xq = xq;
    a.bn = xb_norms_sq;
    a.qn = xq_norms_sq;
    a.scores = out;
    a.ld_scores = ld_out;
    a.out_scale = ldexpf(1.0f, -score_exp);
    a.nb = nb;
    a.nq = nq;
    a.ldb = p.ldb;
    a.ldq = p.ldq;
    a.nseg = p.nseg;
    for (int i = 0; i < 3; ++i) {
        a.seg_q[i] = p.seg_q[i];
        a.seg_c[i] = p.seg_c[i];
    }
    a.nkd = p.nkd;
    a.nk = p.nk;
    a.metric = metric;
    a.k = 1;
    a.ntiles = p.ntiles;
    a.tiles_per_slab = p.tiles_per_slab;
    a.nslab = p.nslab;
    a.nqt = p.nqt;
    a.bq = p.v2 ? LVS2_BQ : LVS3_BQ;
    a.gq = p.gq;
    a.lead_slabs = p.lead_slabs;
    LVS_HIP_CHECK(lvs_tile_launch(LVS_MODE_SCORES, a, (hipStream_t)stream));
    return LVS_OK;
}

extern "C" int32_t lvs_range_join(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack, int64_t nq,
                                  int32_t d, int32_t metric, const float* xb_norms_sq, const float* xq_norms_sq,
                                  float threshold, int32_t score_exp, int64_t q_row0, int64_t id_offset,
                                  int32_t qt_stride, int32_t qt_phase, int64_t capacity, int64_t* out_q, int64_t* out_j,
                                  float* out_s, uint64_t* out_count, void* stream) {
    Plan p;
    LVS_REQUIRE(score_exp >= -120 && score_exp <= 120, "score_exp %d out of range", score_exp);
    LVS_REQUIRE(make_plan(nq, nb, d, xb_pack, xq_pack, 1, p, true) == LVS_OK, "bad shape");
    LVS_REQUIRE(metric == LVS_METRIC_IP || metric == LVS_METRIC_L2, "bad metric %d", metric);
    LVS_REQUIRE(capacity >= 0 && out_count, "bad output buffers");
    LVS_REQUIRE(qt_stride >= 1 && qt_phase >= 0 && qt_phase < qt_stride, "bad tile dealing %d/%d", qt_phase, qt_stride);
    if (nq == 0 || nb == 0) return LVS_OK;
    LVS_REQUIRE(xb && xq && (capacity == 0 || (out_q && out_j && out_s)), "NULL buffer");
    LVS_REQUIRE(metric != LVS_METRIC_L2 || (xb_norms_sq && xq_norms_sq), "L2 needs both norm vectors");
    LVS_DEVICE_GUARD(stream);
    // (r6) fp16 rows, inner product, one rank's tiles: the register-resident-queries kernel in RANGE mode (lvs_rj.hip) - chunks of
    // <= 32 768 queries, each against the rows that can still pair with it (a self-join skips the rows below the chunk's first
    // query: j > i), whole 32-row blocks; the corpus' last nb % 32 rows and chunks left with a short corpus go through the list
    // kernel below.  Same pairs (keys are scores of the same MFMA sequence), one shared counter.
    const bool rj_range = lvs_tune("LVS_RJ", LVS_RJ_DEFAULT) != 0 && lvs_tune("LVS_RJ_RANGE", 1) != 0 && xb_pack == LVS_PACK_F16 &&
                          xq_pack == LVS_PACK_F16 && metric == LVS_METRIC_IP && qt_stride == 1 && nq >= 2048 &&
                          lvs_rq_shape_ok(p.dpad, 1);
    // one launch of the list kernel over queries [c0, c0 + cn) x rows [r0, r0 + rn)
    auto tile_launch = [&](int64_t c0, int64_t cn, int64_t r0, int64_t rn, bool continuation) -> int32_t {
        Plan pl;
        LVS_REQUIRE(make_plan(cn, rn, d, xb_pack, xq_pack, 1, pl, true) == LVS_OK, "bad shape");
        LvsTileArgs a;
        memset(&a, 0, sizeof(a));
        a.xb = (const char*)xb + r0 * pl.ldb * 2;
        a.xq = (const char*)xq + c0 * pl.ldq * 2;
        a.bn = xb_norms_sq ? xb_norms_sq + r0 : nullptr;
        a.qn = xq_norms_sq ? xq_norms_sq + c0 : nullptr;
        a.nb = rn;
        a.nq = cn;
        a.ldb = pl.ldb;
        a.ldq = pl.ldq;
        a.nseg = pl.nseg;
        for (int i = 0; i < 3; ++i) {
            a.seg_q[i] = pl.seg_q[i];
            a.seg_c[i] = pl.seg_c[i];
        }
        a.id_offset = id_offset + r0;
        a.nkd = pl.nkd;
        a.nk = pl.nk;
        a.metric = metric;
        a.k = 1;
        a.ntiles = pl.ntiles;
        a.tiles_per_slab = pl.tiles_per_slab;
        a.nslab = pl.nslab;
        a.nqt = pl.nqt;
        a.bq = pl.v2 ? LVS2_BQ : LVS3_BQ;
        a.gq = pl.gq;
        a.lead_slabs = pl.lead_slabs;
        a.pair_q = (long long*)out_q;
        a.pair_j = (long long*)out_j;
        a.pair_s = out_s;
        a.pair_count = (unsigned long long*)out_count;
        a.pair_capacity = capacity;
        a.q_row0 = q_row0 >= 0 ? q_row0 + c0 : -1;
        a.q_base = c0;
        a.threshold = threshold * ldexpf(1.0f, score_exp);
        a.out_scale = ldexpf(1.0f, -score_exp);
        a.qt_stride = 1;
        a.qt_phase = 0;
        ScopedKernelTimer timer((hipStream_t)stream, LVS_KERNEL_TILE, continuation);
        LVS_HIP_CHECK(lvs_tile_launch(LVS_MODE_RANGE, a, (hipStream_t)stream));
        return LVS_OK;
    };
    if (rj_range) {
        const int64_t chunk = 32768;
        bool first = true;
        for (int64_t c0 = 0, cn = 0; c0 < nq; c0 += cn) {
            const int64_t left = nq - c0;
            cn = left >= chunk ? chunk : (left >= 8192 ? 8192 : (left > LVS_RQ_MAXQ ? LVS_RQ_MAXQ : left));
            // rows that can pair with this chunk's queries: all of them, or (self-join) those above the chunk's first query row
            int64_t r0 = 0;
            if (q_row0 >= 0) {
                r0 = (q_row0 + c0 - id_offset) / 32 * 32;
                if (r0 < 0) r0 = 0;
                if (r0 > nb) r0 = nb;
            }
            const int64_t r_full = nb / 32 * 32;  // the register-resident kernel sees whole blocks
            const int64_t rn = r_full - r0;
            const int64_t groups = (cn + LVS_RQ_GROUPQ - 1) / LVS_RQ_GROUPQ;
            const bool take = rn >= LVS_RQ_JOIN_MINROWS && cn > 128 && (groups <= 32 || groups % 32 == 0);
            if (take) {
                LvsRqArgs ra;
                memset(&ra, 0, sizeof(ra));
                ra.xb = (const char*)xb + r0 * p.ldb * 2;
                ra.xq = (const _Float16*)xq + c0 * p.ldq;
                ra.nb = rn;
                ra.ldb = p.ldb;
                ra.ldq = p.ldq;
                ra.id_offset = id_offset + r0;
                ra.nq = (int)cn;
                ra.k = 1;
                ra.metric = metric;
                ra.pair_q = (long long*)out_q;
                ra.pair_j = (long long*)out_j;
                ra.pair_s = out_s;
                ra.pair_count = (unsigned long long*)out_count;
                ra.pair_capacity = capacity;
                ra.q_row0 = q_row0 >= 0 ? q_row0 + c0 : -1;
                ra.q_base = c0;
                ra.threshold = threshold * ldexpf(1.0f, score_exp);
                ra.out_scale = ldexpf(1.0f, -score_exp);
                {
                    ScopedKernelTimer timer((hipStream_t)stream, LVS_KERNEL_RJ, !first);
                    LVS_HIP_CHECK(lvs_rj_range_launch(ra, p.dpad, (hipStream_t)stream));
                }
                first = false;
                if (r_full < nb) {  // the corpus' last nb % 32 rows
                    const int32_t rc = tile_launch(c0, cn, r_full, nb - r_full, true);
                    if (rc != LVS_OK) return rc;
                }
            } else if (nb - r0 > 0) {
                const int32_t rc = tile_launch(c0, cn, r0, nb - r0, !first);
                if (rc != LVS_OK) return rc;
                first = false;
            }
        }
        return LVS_OK;
    }
    LvsTileArgs a;
    memset(&a, 0, sizeof(a));
    a.xb = xb;
    a.xq = xq;
    a.bn = xb_norms_sq;
    a.qn = xq_norms_sq;
    a.nb = nb;
    a.nq = nq;
    a.ldb = p.ldb;
    a.ldq = p.ldq;
    a.nseg = p.nseg;
    for (int i = 0; i < 3; ++i) {
        a.seg_q[i] = p.seg_q[i];
        a.seg_c[i] = p.seg_c[i];
    }
    a.id_offset = id_offset;
    a.nkd = p.nkd;
    a.nk = p.nk;
    a.metric = metric;
    a.k = 1;
    a.ntiles = p.ntiles;
    a.tiles_per_slab = p.tiles_per_slab;
    a.nslab = p.nslab;
    a.nqt = p.nqt;
    a.bq = p.v2 ? LVS2_BQ : LVS3_BQ;
    a.gq = p.gq;
    a.lead_slabs = p.lead_slabs;
    a.pair_q = (long long*)out_q;
    a.pair_j = (long long*)out_j;
    a.pair_s = out_s;
    a.pair_count = (unsigned long long*)out_count;
    a.pair_capacity = capacity;
    a.q_row0 = q_row0;
    a.threshold = threshold * ldexpf(1.0f, score_exp);  // the device scores are 2^score_exp x the caller's (exact)
    a.out_scale = ldexpf(1.0f, -score_exp);
    a.qt_stride = qt_stride;
    a.qt_phase = qt_phase;
    {
        ScopedKernelTimer timer((hipStream_t)stream);
        LVS_HIP_CHECK(lvs_tile_launch(LVS_MODE_RANGE, a, (hipStream_t)stream));
    }
    return LVS_OK;
}

// k-means pieces behind lotus.utils.cluster (lotus/utils.py:61-65 -> faiss Kmeans.train + index.search(x, 1)).
//
// Assignment is the tile kernel in top-1 / squared-L2 mode (lvs_flat_search_keys with k = 1).  This file holds the
// centroid update and the host-side pieces of faiss's Clustering::train that must match bit for bit:
//   * lvs_kmeans_accumulate: rows are bucketed by centroid with a STABLE radix sort on the assignment bits
//     (rocPRIM), then every (centroid, 512-dim chunk) is reduced by one wave that walks its bucket in row
//     order - the accumulation order of faiss compute_centroids, so results are reproducible run to run;
//   * lvs_rand_perm_host / lvs_kmeans_split_clusters_host: std::mt19937-driven subsample/init permutation and
//     empty-cluster re-seeding exactly as faiss.
#include <cstring>
#include <random>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "lvs_common.h"
#include "lvs_tile.h"

namespace {

__global__ __launch_bounds__(256) void km_keys_kernel(const long long* __restrict__ assign, long long n, int k,
                                                      uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long c = assign[i];
    keys[i] = (c < 0 || c >= k) ? (uint32_t)k : (uint32_t)c;  // out-of-range assignments go to an ignored bucket
    vals[i] = (uint32_t)i;
}

// bucket boundaries from the SORTED keys: offsets[c] = first position whose key is >= c, c = 0..k
// (no atomics: position i writes the offsets of every centroid whose bucket starts there; k + 1 writes in total)
__global__ __launch_bounds__(256) void km_bounds_kernel(const uint32_t* __restrict__ sorted_keys, long long n, int k,
                                                        uint32_t* __restrict__ offsets) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const long long lo = i == 0 ? 0 : (long long)sorted_keys[i - 1] + 1;
    const long long hi = i == n ? (long long)k : (long long)sorted_keys[i];
    for (long long c = lo; c <= hi && c <= k; ++c) offsets[c] = (uint32_t)i;
}

typedef _Float16 km_half8 __attribute__((ext_vector_type(8)));

// grid = (k, ceil(dpad / 512)), one wave per workgroup: lane owns 8 consecutive dimensions of centroid blockIdx.x
// and walks the bucket in row order, U rows (16-byte loads) in flight at a time.  Per dimension the additions
// happen in exactly the bucket (= ascending row) order, so the sums do not depend on U or on the launch shape.
template <int SPLIT>
__global__ __launch_bounds__(64) void km_reduce_kernel(const _Float16* __restrict__ x, long long ld, int d, int dpad,
                                                       const uint32_t* __restrict__ rows,
                                                       const uint32_t* __restrict__ offsets,
                                                       float* __restrict__ sums, float* __restrict__ cnt_out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int j0 = (blockIdx.y * 64 + lane) * 8;
    const uint32_t b = offsets[c], e = offsets[c + 1];
    if (blockIdx.y == 0 && lane == 0) cnt_out[c] += (float)(e - b);
    if (j0 >= dpad) return;
    constexpr int U = SPLIT ? 8 : 16;
    float acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = 0.f;
    const _Float16* xc = x + j0;
    uint32_t p = b;
    for (; p + U <= e; p += U) {
        km_half8 hi[U], lo[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const _Float16* row = xc + (long long)rows[p + i] * ld;
            hi[i] = *(const km_half8*)row;
            if (SPLIT) lo[i] = *(const km_half8*)(row + dpad);
        }
#pragma unroll
        for (int i = 0; i < U; ++i)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] += SPLIT ? (float)hi[i][t] + (float)lo[i][t] : (float)hi[i][t];
    }
    for (; p < e; ++p) {
        const _Float16* row = xc + (long long)rows[p] * ld;
        km_half8 h = *(const km_half8*)row, l;
        if (SPLIT) l = *(const km_half8*)(row + dpad);
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] += SPLIT ? (float)h[t] + (float)l[t] : (float)h[t];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (j0 + t < d) sums[(long long)c * d + j0 + t] += acc[t];
}

// centroid update of faiss compute_centroids: c = sum * (1 / count) where the cluster is not empty, unchanged otherwise
// (an empty cluster keeps its previous centroid until split_clusters re-seeds it)
__global__ __launch_bounds__(256) void km_update_kernel(const float* __restrict__ sums, const float* __restrict__ counts,
                                                        long long total, int d, float* __restrict__ centroids) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const float cnt = counts[i / d];
        if (cnt > 0.f) {
            const float norm = 1.0f / cnt;  // IEEE division, then one rounded multiply per element (as faiss)
            centroids[i] = sums[i] * norm;
        }
    }
}

// dst[i][0..d) = float32 value of packed row ids[i] (row i when ids == NULL): hi (+ lo)
__global__ __launch_bounds__(256) void unpack_rows_kernel(const _Float16* __restrict__ src, long long ld, int d, int dpad,
                                                          int split, const long long* __restrict__ ids, long long n,
                                                          float* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const _Float16* row = src + (ids ? ids[i] : i) * ld;
    float* o = dst + i * (long long)d;
    for (int j = lane; j < d; j += 64) {
        float v = (float)row[j];
        if (split) v += (float)row[dpad + j];
        o[j] = v;
    }
}

}  // namespace

extern "C" int32_t lvs_kmeans_update_centroids(const float* sums, const float* counts, int32_t k, int32_t d,
                                               float* centroids, void* stream) {
    LVS_REQUIRE(k > 0 && d > 0, "bad shape k=%d d=%d", k, d);
    LVS_REQUIRE(sums && counts && centroids, "NULL buffer");
    LVS_DEVICE_GUARD(stream);
    const long long total = (long long)k * d;
    const long long blocks = lvs_ceil_div(total, 256);
    hipLaunchKernelGGL(km_update_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       (hipStream_t)stream, sums, counts, total, d, centroids);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_unpack_rows(const void* src, int32_t d, int32_t pack_mode, const int64_t* ids, int64_t n,
                                   float* dst, void* stream) {
    LVS_REQUIRE(d > 0 && n >= 0, "bad shape n=%lld d=%d", (long long)n, d);
    LVS_REQUIRE(pack_mode == LVS_PACK_F16 || pack_mode == LVS_PACK_SPLIT, "bad pack_mode %d", pack_mode);
    if (n == 0) return LVS_OK;
    LVS_REQUIRE(src && dst, "NULL buffer");
    LVS_DEVICE_GUARD(stream);
    const int dpad = (int)lvs_round_up(d, LVS_BK);
    const int split = pack_mode == LVS_PACK_SPLIT;
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)lvs_ceil_div(n, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)src, (long long)(split ? 2 * dpad : dpad), d, dpad, split,
                       (const long long*)ids, (long long)n, dst);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int64_t lvs_kmeans_accumulate_workspace_bytes(int64_t n, int32_t k) {
    if (n < 0 || k <= 0) return LVS_EINVAL;
    size_t tmp = 0;
    uint32_t* nil = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, nil, nil, nil, nil, (size_t)(n > 0 ? n : 1), 0u, 32u);
    int64_t bytes = lvs_round_up((int64_t)tmp, 256);
    bytes += 4 * lvs_round_up(n * 4, 256);             // keys in/out, vals in/out
    bytes += 2 * lvs_round_up((int64_t)(k + 2) * 4, 256);  // bucket offsets (+ spare)
    return bytes;
}

extern "C" int32_t lvs_kmeans_accumulate(const void* x, int64_t n, int32_t d, int32_t pack_mode, const int64_t* assign,
                                         int32_t k, float* sums, float* counts, void* workspace, int64_t workspace_bytes,
                                         void* stream) {
    LVS_REQUIRE(n >= 0 && d > 0 && k > 0, "bad shape n=%lld d=%d k=%d", (long long)n, d, k);
    LVS_REQUIRE(pack_mode == LVS_PACK_F16 || pack_mode == LVS_PACK_SPLIT, "bad pack_mode");
    LVS_REQUIRE(n < 0xFFFFFFFFll, "n must be below 2^32");
    if (n == 0) return LVS_OK;
    LVS_REQUIRE(x && assign && sums && counts && workspace, "NULL buffer");
    const int64_t need = lvs_kmeans_accumulate_workspace_bytes(n, k);
    if (workspace_bytes < need) {
        lvs_set_error("workspace too small: need %lld bytes", (long long)need);
        return LVS_ENOMEM;
    }
    LVS_DEVICE_GUARD(stream);
    hipStream_t st = (hipStream_t)stream;
    size_t tmp = 0;
    uint32_t* nil = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, nil, nil, nil, nil, (size_t)n, 0u, 32u);
    char* w = (char*)workspace;
    void* d_tmp = w;
    w += lvs_round_up((int64_t)tmp, 256);
    uint32_t* keys_in = (uint32_t*)w;
    w += lvs_round_up(n * 4, 256);
    uint32_t* keys_out = (uint32_t*)w;
    w += lvs_round_up(n * 4, 256);
    uint32_t* vals_in = (uint32_t*)w;
    w += lvs_round_up(n * 4, 256);
    uint32_t* vals_out = (uint32_t*)w;
    w += lvs_round_up(n * 4, 256);
    uint32_t* offs = (uint32_t*)w;

    hipLaunchKernelGGL(km_keys_kernel, dim3((unsigned)lvs_ceil_div(n, 256)), dim3(256), 0, st, (const long long*)assign,
                       (long long)n, k, keys_in, vals_in);
    unsigned bits = 1;
    while ((1u << bits) < (unsigned)(k + 1)) ++bits;
    LVS_HIP_CHECK(rocprim::radix_sort_pairs(d_tmp, tmp, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, bits, st));
    hipLaunchKernelGGL(km_bounds_kernel, dim3((unsigned)lvs_ceil_div(n + 1, 256)), dim3(256), 0, st, keys_out,
                       (long long)n, k, offs);
    const int dpad = (int)lvs_round_up(d, LVS_BK);
    const long long ld = pack_mode == LVS_PACK_SPLIT ? 2 * dpad : dpad;
    const dim3 grid((unsigned)k, (unsigned)lvs_ceil_div(dpad, 512));
    if (pack_mode == LVS_PACK_SPLIT)
        hipLaunchKernelGGL(km_reduce_kernel<1>, grid, dim3(64), 0, st, (const _Float16*)x, ld, d, dpad, vals_out, offs,
                           sums, counts);
    else
        hipLaunchKernelGGL(km_reduce_kernel<0>, grid, dim3(64), 0, st, (const _Float16*)x, ld, d, dpad, vals_out, offs,
                           sums, counts);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

// ---- host-side pieces of faiss Clustering (bit-exact: std::mt19937 is the generator faiss uses) -------------
extern "C" int32_t lvs_rand_perm_host(int64_t n, int64_t seed, int64_t* out_perm) {
    LVS_REQUIRE(n >= 0 && (n == 0 || out_perm), "bad arguments");
    std::mt19937 mt((unsigned)seed);
    for (int64_t i = 0; i < n; ++i) out_perm[i] = i;
    for (int64_t i = 0; i + 1 < n; ++i) {
        int64_t i2 = i + (int64_t)(mt() % (uint32_t)(n - i));  // faiss rand_int(max) = mt() % max
        int64_t t = out_perm[i];
        out_perm[i] = out_perm[i2];
        out_perm[i2] = t;
    }
    return LVS_OK;
}

// The first m entries of the same permutation in O(m): step i of the forward Fisher-Yates fixes perm[i] and touches only
// perm[i] and perm[i2 >= i], and both callers (training subsample, initial centroids) read a prefix - of 10 M entries at
// configs[4]'s size, where the full permutation costs 0.17 s of host time per call.  Positions >= m live in a hash map.
extern "C" int32_t lvs_rand_perm_prefix_host(int64_t n, int64_t seed, int64_t m, int64_t* out_prefix) {
    LVS_REQUIRE(n >= 0 && m >= 0 && m <= n && (m == 0 || out_prefix), "bad arguments");
    std::mt19937 mt((unsigned)seed);
    for (int64_t i = 0; i < m; ++i) out_prefix[i] = i;
    // open-addressing table (linear probing), key and value side by side (one cache line per probe), at most one new
    // entry per step: <= 50 % full.  position >= m -> its current content (absent: the position itself)
    struct Slot { int64_t key, val; };
    size_t cap = 16;
    while (cap < (size_t)m * 2) cap <<= 1;
    std::vector<Slot> tab(cap, Slot{-1, 0});
    for (int64_t i = 0; i < m && i + 1 < n; ++i) {
        const int64_t i2 = i + (int64_t)(mt() % (uint32_t)(n - i));
        if (i2 < m) {
            const int64_t t = out_prefix[i];
            out_prefix[i] = out_prefix[i2];
            out_prefix[i2] = t;
        } else {
            size_t h = (size_t)(((uint64_t)i2 * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
            while (tab[h].key != -1 && tab[h].key != i2) h = (h + 1) & (cap - 1);
            const int64_t v = tab[h].key == i2 ? tab[h].val : i2;
            tab[h].key = i2;
            tab[h].val = out_prefix[i];
            out_prefix[i] = v;
        }
    }
    return LVS_OK;
}

extern "C" int32_t lvs_kmeans_split_clusters_host(int32_t d, int32_t k, int64_t n, float* hassign, float* centroids,
                                                  int32_t* out_nsplit) {
    LVS_REQUIRE(d > 0 && k > 0 && n > k && hassign && centroids, "bad arguments");
    const float EPS = 1.0f / 1024.0f;
    std::mt19937 mt(1234u);
    int32_t nsplit = 0;
    for (int32_t ci = 0; ci < k; ++ci) {
        if (hassign[ci] != 0.f) continue;
        int32_t cj;
        for (cj = 0;; cj = (cj + 1) % k) {
            float p = (hassign[cj] - 1.0f) / (float)(n - k);
            float r = (float)mt() / (float)mt.max();  // faiss rand_float()
            if (r < p) break;
        }
        float* a = centroids + (int64_t)ci * d;
        float* b = centroids + (int64_t)cj * d;
        memcpy(a, b, sizeof(float) * (size_t)d);
        for (int32_t j = 0; j < d; ++j) {
            if (j % 2 == 0) {
                a[j] *= 1 + EPS;
                b[j] *= 1 - EPS;
            } else {
                a[j] *= 1 - EPS;
                b[j] *= 1 + EPS;
            }
        }
        hassign[ci] = hassign[cj] / 2;
        hassign[cj] -= hassign[ci];
        ++nsplit;
    }
    if (out_nsplit) *out_nsplit = nsplit;
    return LVS_OK;
}

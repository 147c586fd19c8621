// k-means pieces behind lotus.utils.cluster (lotus/utils.py:61-65 -> faiss Kmeans.train + index.search(x, 1)).
//
// Assignment is the tile kernel in top-1 / squared-L2 mode (lvs_flat_search_keys with k = 1, or the certified one-pass
// lvs_nearest_hi).  This file holds everything else of an iteration, all of it on the device so that the host only
// enqueues launches:
//   * lvs_kmeans_accumulate(_keys): rows are bucketed by centroid with a STABLE counting sort on the centroid ids
//     (per-chunk histogram in LDS -> exclusive scan -> in-order scatter; hand-written - round 2 used rocPRIM's radix
//     sort), then every (centroid, 512-dim chunk) is reduced by one wave that walks its bucket in row order - the
//     accumulation order of faiss compute_centroids, so results are reproducible run to run;
//   * lvs_kmeans_objective: faiss's objective (sum of the assignment distances) from the sums the update needs anyway;
//   * lvs_kmeans_update_centroids: centroid division, faiss's empty-cluster split (std::mt19937 replayed by one device
//     thread: same draws, same decisions as faiss split_clusters), repacking of the centroids as fp16 hi|lo rows and the
//     two norms the one-pass assignment's certificate needs;
//   * lvs_rand_perm_host / lvs_kmeans_split_clusters_host: the host twins (training subsample, initial centroids; the
//     split routine is kept for callers that hold the centroids on the host) - SURVEY.md Appendix A.4.
#include <cstring>
#include <random>
#include <vector>

#include "lvs_common.h"
#include "lvs_tile.h"

namespace {

// ---- stable counting sort of the rows by centroid id -----------------------------------------------------------------
// bins = k + 1 (the last one collects assignments outside [0, k): ignored rows).  A launch works on chunks of KM_CHUNK
// consecutive rows, one workgroup each:
//   km_count_kernel    counts[bin][chunk] = rows of the chunk that go to `bin`          (histogram in LDS)
//   km_scan*_kernel    exclusive scan over counts in (bin-major, chunk-minor) order = first output position of every
//                      (bin, chunk) run; offsets[c] = start of bucket c, c = 0 .. k
//   km_scatter_kernel  rows_out[position] = row, in row order inside a chunk (waves of a tile take turns, lanes rank
//                      themselves among the lanes with the same bin by ballots over the bin's bits)
// Rows keep their order inside a bucket, which is what makes the centroid sums independent of the launch shape.
constexpr int KM_CHUNK = 8192;
constexpr int KM_SCAN_SEG = 2048;   // entries per workgroup of the scan
constexpr int KM_MAX_BINS = 24576;  // (k + 1) * 4 B of LDS per workgroup; larger k sorts by two digits

template <typename KeyT>
__device__ inline uint32_t km_bin_of(KeyT v, long long id_offset, int k);
template <>
__device__ inline uint32_t km_bin_of<long long>(long long c, long long, int k) {
    return (c < 0 || c >= k) ? (uint32_t)k : (uint32_t)c;
}
template <>
__device__ inline uint32_t km_bin_of<u64>(u64 key, long long id_offset, int k) {  // a result key: id in the low word
    if (key == 0) return (uint32_t)k;
    const long long c = (long long)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull)) - id_offset;
    return (c < 0 || c >= k) ? (uint32_t)k : (uint32_t)c;
}

// digit of a row: ((bin >> shift) & mask); rows are read through `order` (nullable: identity) so that a second pass can
// sort the output of the first
template <typename KeyT>
__global__ __launch_bounds__(256) void km_count_kernel(const KeyT* __restrict__ assign, const uint32_t* __restrict__ order,
                                                       long long n, int k, long long id_offset, int shift, uint32_t mask,
                                                       int nbins, int nchunks, uint32_t* __restrict__ counts) {
    extern __shared__ uint32_t km_hist[];
    for (int b = threadIdx.x; b < nbins; b += 256) km_hist[b] = 0;
    __syncthreads();
    const long long r0 = (long long)blockIdx.x * KM_CHUNK;
    const long long r1 = r0 + KM_CHUNK < n ? r0 + KM_CHUNK : n;
    for (long long i = r0 + threadIdx.x; i < r1; i += 256) {
        const long long row = order ? (long long)order[i] : i;
        atomicAdd(&km_hist[(km_bin_of<KeyT>(assign[row], id_offset, k) >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += 256) counts[(long long)b * nchunks + blockIdx.x] = km_hist[b];
}

// exclusive scan, three launches: per-segment scan + segment totals, scan of the totals (one workgroup), add
__global__ __launch_bounds__(256) void km_scan1_kernel(uint32_t* __restrict__ v, long long total, uint32_t* __restrict__ seg_sum) {
    __shared__ uint32_t part[256];
    const long long base = (long long)blockIdx.x * KM_SCAN_SEG + (long long)threadIdx.x * (KM_SCAN_SEG / 256);
    uint32_t loc[KM_SCAN_SEG / 256], sum = 0;
#pragma unroll
    for (int i = 0; i < KM_SCAN_SEG / 256; ++i) {
        loc[i] = base + i < total ? v[base + i] : 0u;
        sum += loc[i];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan of the 256 thread sums
        const uint32_t add = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;  // exclusive prefix of this thread inside the segment
#pragma unroll
    for (int i = 0; i < KM_SCAN_SEG / 256; ++i) {
        if (base + i < total) v[base + i] = run;
        run += loc[i];
    }
    if (threadIdx.x == 255) seg_sum[blockIdx.x] = part[255];
}
__global__ __launch_bounds__(256) void km_scan2_kernel(uint32_t* __restrict__ seg_sum, int nseg) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int s0 = 0; s0 < nseg; s0 += 256) {
        const int i = s0 + threadIdx.x;
        const uint32_t mine = i < nseg ? seg_sum[i] : 0u;
        part[threadIdx.x] = mine;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const uint32_t add = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nseg) seg_sum[i] = carry + part[threadIdx.x] - mine;
        __syncthreads();
        if (threadIdx.x == 255) carry += part[255];
        __syncthreads();
    }
}
// v += its segment's offset; offsets[b] = v[b * nchunks] for b = 0 .. nbuckets (start of every bucket; written only when
// `offsets` is given - the last digit pass)
__global__ __launch_bounds__(256) void km_scan3_kernel(uint32_t* __restrict__ v, long long total,
                                                       const uint32_t* __restrict__ seg_sum, int nchunks,
                                                       uint32_t* __restrict__ offsets, int nbuckets) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) {
        const uint32_t val = v[i] + seg_sum[i / KM_SCAN_SEG];
        v[i] = val;
        if (offsets && i % nchunks == 0 && i / nchunks <= nbuckets) offsets[i / nchunks] = val;
    }
}

// bucket boundaries from the SORTED rows (two-digit sorts only): offsets[c] = first position whose bin is >= c, c = 0..k
// (no atomics: position i writes the offsets of every bucket that starts there; k + 1 writes in total)
template <typename KeyT>
__global__ __launch_bounds__(256) void km_bounds_kernel(const KeyT* __restrict__ assign, const uint32_t* __restrict__ rows,
                                                        long long n, int k, long long id_offset,
                                                        uint32_t* __restrict__ offsets) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const long long lo = i == 0 ? 0 : (long long)km_bin_of<KeyT>(assign[rows[i - 1]], id_offset, k) + 1;
    const long long hi = i == n ? (long long)k : (long long)km_bin_of<KeyT>(assign[rows[i]], id_offset, k);
    for (long long c = lo; c <= hi && c <= k; ++c) offsets[c] = (uint32_t)i;
}

template <typename KeyT>
__global__ __launch_bounds__(256) void km_scatter_kernel(const KeyT* __restrict__ assign, const uint32_t* __restrict__ order,
                                                         long long n, int k, long long id_offset, int shift, uint32_t mask,
                                                         int nbins, int nbits, int nchunks,
                                                         const uint32_t* __restrict__ counts, uint32_t* __restrict__ rows_out) {
    extern __shared__ uint32_t km_pos[];  // next output position of every bin for this chunk
    for (int b = threadIdx.x; b < nbins; b += 256) km_pos[b] = counts[(long long)b * nchunks + blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.x * KM_CHUNK;
    const long long r1 = r0 + KM_CHUNK < n ? r0 + KM_CHUNK : n;
    for (long long t0 = r0; t0 < r1; t0 += 256) {
        const long long i = t0 + threadIdx.x;
        const bool live = i < r1;
        uint32_t row = 0, bin = 0;
        if (live) {
            row = order ? order[i] : (uint32_t)i;
            bin = (km_bin_of<KeyT>(assign[row], id_offset, k) >> shift) & mask;
        }
        // lanes of this wave holding the same bin (dead lanes match nobody)
        u64 peers = __builtin_amdgcn_ballot_w64(live);
        for (int b = 0; b < nbits; ++b) {
            const u64 m = __builtin_amdgcn_ballot_w64(live && ((bin >> b) & 1u));
            peers &= ((bin >> b) & 1u) ? m : ~m;
        }
        const int rank = __popcll(peers & ((1ull << lane) - 1ull));
        const int cnt = __popcll(peers);
        uint32_t base = 0;
        for (int w = 0; w < 4; ++w) {  // waves take turns: rows of wave w come before those of wave w + 1
            if (wave == w && live && rank == 0) {
                base = km_pos[bin];
                km_pos[bin] = base + (uint32_t)cnt;
            }
            __syncthreads();
        }
        // the leader's base -> its peers
        const int leader = live ? __ffsll((long long)peers) - 1 : lane;
        base = __shfl(base, leader, 64);
        if (live) rows_out[base + (uint32_t)rank] = row;
    }
}

// grid = (k, ceil(dpad / (64 * DPL))), one wave per workgroup: a lane owns DPL consecutive dimensions (8, 4 or 2: 16-, 8- or
// 4-byte loads) of centroid blockIdx.x and walks the bucket in row order, U rows in flight at a time.  Per dimension the
// additions happen in exactly the bucket (= ascending row) order, so the sums do not depend on DPL, U or the launch shape -
// faiss's accumulation order (compute_centroids), bit for bit.
// A bucket is one dependent chain of additions per wave, and what a row costs a wave is mostly fixed (row number, 64-bit address,
// load issue) - so the LONGEST bucket sets the kernel's time when the sizes are skewed (configs[4]'s blob rows: up to 3-4 x the
// mean).  Narrower column slices put more waves on every bucket without touching the order of any column's sum.
template <int DPL>
struct KmVec;
template <>
struct KmVec<8> {
    typedef _Float16 type __attribute__((ext_vector_type(8)));
};
template <>
struct KmVec<4> {
    typedef _Float16 type __attribute__((ext_vector_type(4)));
};
template <>
struct KmVec<2> {
    typedef _Float16 type __attribute__((ext_vector_type(2)));
};

template <int SPLIT, int DPL>
__global__ __launch_bounds__(64) void km_reduce_kernel(const _Float16* __restrict__ x, long long ld, int d, int dpad,
                                                       const uint32_t* __restrict__ rows,
                                                       const uint32_t* __restrict__ offsets,
                                                       float* __restrict__ sums, float* __restrict__ cnt_out) {
    typedef typename KmVec<DPL>::type vec_t;
    const int c = blockIdx.x, lane = threadIdx.x;
    const int j0 = (blockIdx.y * 64 + lane) * DPL;
    const uint32_t b = offsets[c], e = offsets[c + 1];
    if (blockIdx.y == 0 && lane == 0) cnt_out[c] += (float)(e - b);
    if (j0 >= dpad) return;
    // rows in flight: 128 data VGPRs' worth (hi parts; half as many rows when the lo parts ride along), at most 64
    constexpr int U = (DPL == 8 ? 32 : 64) / (SPLIT ? 2 : 1);
    // the sums CONTINUE from what is there (zeros before the first rows): a caller may hand the rows over in consecutive ranges
    // - (s + x1) + x2 ... is the same chain of float32 additions whether it is cut into launches or not
    float acc[DPL];
#pragma unroll
    for (int t = 0; t < DPL; ++t) acc[t] = j0 + t < d ? sums[(long long)c * d + j0 + t] : 0.f;
    const _Float16* xc = x + j0;
    // many rows per batch, and the row NUMBERS of the next batch are fetched while this batch is added (they are wave-uniform:
    // scalar loads), so that a batch waits for one latency, not two
    const uint32_t nfull = (e - b) / U;
    uint32_t p = b;
    if (nfull) {
        uint32_t idx[U];
#pragma unroll
        for (int i = 0; i < U; ++i) idx[i] = rows[p + i];
        for (uint32_t kb = 0; kb < nfull; ++kb) {
            vec_t hi[U], lo[U];
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const _Float16* row = xc + (long long)idx[i] * ld;
                hi[i] = *(const vec_t*)row;
                if (SPLIT) lo[i] = *(const vec_t*)(row + dpad);
            }
            if (kb + 1 < nfull) {
#pragma unroll
                for (int i = 0; i < U; ++i) idx[i] = rows[p + U + i];
            }
#pragma unroll
            for (int i = 0; i < U; ++i)
#pragma unroll
                for (int t = 0; t < DPL; ++t) acc[t] += SPLIT ? (float)hi[i][t] + (float)lo[i][t] : (float)hi[i][t];
            p += U;
        }
    }
    for (; p < e; ++p) {
        const _Float16* row = xc + (long long)rows[p] * ld;
        vec_t h = *(const vec_t*)row, l;
        if (SPLIT) l = *(const vec_t*)(row + dpad);
#pragma unroll
        for (int t = 0; t < DPL; ++t) acc[t] += SPLIT ? (float)h[t] + (float)l[t] : (float)h[t];
    }
#pragma unroll
    for (int t = 0; t < DPL; ++t)
        if (j0 + t < d) sums[(long long)c * d + j0 + t] = acc[t];
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
                                                          float unscale, float* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const _Float16* row = src + (ids ? ids[i] : i) * ld;
    float* o = dst + i * (long long)d;
    for (int j = lane; j < d; j += 64) {
        float v = (float)row[j];
        if (split) v += (float)row[dpad + j];
        o[j] = v * unscale;
    }
}

}  // namespace

// ---- objective, empty-cluster split and centroid statistics on the device ------------------------------------------------
namespace {
__device__ inline double km_wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const long long b = __double_as_longlong(v);
        const int lo = __shfl_xor((int)(b & 0xFFFFFFFFll), m, 64), hi = __shfl_xor((int)(b >> 32), m, 64);
        v += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    return v;
}
// part[j] = -2 <c_j, S_j> + n_j |c_j|^2 in float64 (one wave per centroid; fixed summation order)
__global__ __launch_bounds__(256) void km_obj_part_kernel(const float* __restrict__ c, const float* __restrict__ sums,
                                                          const float* __restrict__ counts, int k, int d,
                                                          double* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= k) return;
    double dot = 0.0, nn = 0.0;
    for (int t = lane; t < d; t += 64) {
        const double cv = (double)c[(long long)j * d + t];
        dot += cv * (double)sums[(long long)j * d + t];
        nn += cv * cv;
    }
    dot = km_wave_sum_d(dot);
    nn = km_wave_sum_d(nn);
    if (lane == 0) part[j] = -2.0 * dot + (double)counts[j] * nn;
}
// out[0] = x2 + sum_j part[j]  (one workgroup, fixed order)
__global__ __launch_bounds__(256) void km_obj_sum_kernel(const double* __restrict__ part, int k, const double* __restrict__ x2,
                                                         double* __restrict__ out) {
    __shared__ double sm[256];
    double acc = 0.0;
    for (int j = threadIdx.x; j < k; j += 256) acc += part[j];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (x2 ? x2[0] : 0.0) + sm[0];
}

// faiss split_clusters on the device: ONE wave replays std::mt19937(1234) and takes every decision exactly as the host twin
// (lvs_kmeans_split_clusters_host); all lanes copy / perturb the centroid rows.  hassign (= the counts) is modified in
// place as faiss does.  Nothing happens - not even the generator's set-up - unless some cluster is empty.
//
// A split walks the clusters cyclically and accepts cluster cj with probability (hassign[cj] - 1) / (n - k): ~k draws per
// split, 135 000 for the 132 empty clusters of configs[4]'s first iteration.  One thread doing that against global memory
// took 67 ms; here the 64 lanes test 64 consecutive (draw, cluster) pairs at once - the draws are consumed in faiss's order,
// the first accepting lane wins and the stream position advances by exactly the draws faiss would have used - with the
// sizes in LDS and the generator's 624-word state regenerated by all lanes (three dependency-free segments).
struct KmMtWave {
    uint32_t* mt;  // [624] in LDS
    __device__ void seed(uint32_t s, int lane) {
        if (lane == 0) {
            mt[0] = s;
            for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __device__ static uint32_t mix(uint32_t a, uint32_t b, uint32_t c) {  // next state word from mt[i], mt[i+1], mt[i+397]
        const uint32_t y = (a & 0x80000000u) | (b & 0x7FFFFFFFu);
        return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
    }
    // the standard in-place twist; word i needs OLD mt[i], mt[i+1] and mt[i+397 mod 624], which is old for i < 227 and already
    // NEW for i >= 227: segments [0,227), [227,454), [454,623) are each free of internal dependencies, word 623 comes last
    __device__ void twist(int lane) {
        for (int base = 0; base < 227; base += 64) {
            const int i = base + lane;
            uint32_t v = 0;
            if (i < 227) v = mix(mt[i], mt[i + 1], mt[i + 397]);
            __builtin_amdgcn_wave_barrier();
            if (i < 227) mt[i] = v;
            __builtin_amdgcn_wave_barrier();
        }
        // mt[i + 1] read by word i = 226 was still old when it was computed (all reads of a round precede its writes; rounds
        // run upwards, and word i only ever reads indices > i or, for i >= 227, the finished new words below i - 226)
        for (int seg = 227; seg < 623; seg += 227) {
            const int hi = seg + 227 < 623 ? seg + 227 : 623;
            for (int base = seg; base < hi; base += 64) {
                const int i = base + lane;
                uint32_t v = 0;
                if (i < hi) v = mix(mt[i], mt[i + 1], mt[i - 227]);
                __builtin_amdgcn_wave_barrier();
                if (i < hi) mt[i] = v;
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane == 0) mt[623] = mix(mt[623], mt[0], mt[396]);
        __builtin_amdgcn_wave_barrier();
    }
    __device__ static uint32_t temper(uint32_t y) {
        y ^= y >> 11;
        y ^= (y << 7) & 0x9D2C5680u;
        y ^= (y << 15) & 0xEFC60000u;
        y ^= y >> 18;
        return y;
    }
};
constexpr int KM_SPLIT_LDS_K = 15360;  // cluster sizes kept in LDS up to this k (with the generator's state: < 64 KB); beyond, global memory
__global__ __launch_bounds__(64) void km_split_kernel(int d, int k, long long n, float* hassign_, float* __restrict__ centroids,
                                                      int* __restrict__ out_nsplit) {
    extern __shared__ uint32_t km_split_smem[];
    uint32_t* state = km_split_smem;                 // [624] (+ 16 pad)
    float* hl = (float*)(km_split_smem + 640);       // [k] when k <= KM_SPLIT_LDS_K
    const int lane = threadIdx.x;
    volatile float* hg = hassign_;
    const bool in_lds = k <= KM_SPLIT_LDS_K;
    bool any = false;
    for (int c = lane; c < k; c += 64) {
        const float h = hg[c];
        if (in_lds) hl[c] = h;
        any |= h == 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    if (__builtin_amdgcn_ballot_w64(any) == 0ull || n <= k) {
        if (lane == 0 && out_nsplit) *out_nsplit = 0;
        return;
    }
    auto H = [&](int c) -> float { return in_lds ? hl[c] : hg[c]; };
    auto setH = [&](int c, float v) {
        if (in_lds) hl[c] = v;
        hg[c] = v;
    };
    KmMtWave rng{state};
    rng.seed(1234u, lane);
    int idx = 624;  // next unused word of the current state block (624: regenerate first), wave-uniform
    const float EPS = 1.0f / 1024.0f;
    const float denom = (float)(n - k);
    int nsplit = 0;
    for (int ci = 0; ci < k; ++ci) {
        if (H(ci) != 0.f) continue;  // wave-uniform (every lane reads the same word)
        // walk cj = 0, 1, 2, ... (cyclically) consuming one draw per step until r < p(cj)
        int cj0 = 0, found = -1;
        long long draws = 0;
        // the acceptance probabilities of one round over the clusters sum to 1, so a split takes ~k draws; the cap only
        // guards the device against counts that do not describe n points (faiss itself would spin forever)
        const long long cap = 4096ll * k + (1ll << 20);
        while (found < 0 && draws <= cap) {
            if (idx >= 624) {
                rng.twist(lane);
                idx = 0;
            }
            const int avail = 624 - idx < 64 ? 624 - idx : 64;  // draws this round (one per lane), wave-uniform
            bool acc = false;
            if (lane < avail) {
                const int cj = (cj0 + lane) % k;
                const float p = (H(cj) - 1.0f) / denom;
                const float r = (float)KmMtWave::temper(state[idx + lane]) / 4294967295.0f;  // faiss rand_float(): mt() / float(mt.max())
                acc = r < p;
            }
            const u64 m = __builtin_amdgcn_ballot_w64(acc);
            if (m) {
                const int w = __ffsll((long long)m) - 1;  // the first accepting step: later draws of this round stay unused
                found = (cj0 + w) % k;
                idx += w + 1;
                draws += w + 1;
            } else {
                idx += avail;
                draws += avail;
                cj0 = (cj0 + avail) % k;
            }
        }
        if (found < 0) {  // inconsistent counts: report and stop
            if (lane == 0 && out_nsplit) *out_nsplit = -1;
            return;
        }
        const int cj = found;
        if (lane == 0) {
            const float hj = H(cj);
            const float half = hj / 2;
            setH(ci, half);
            setH(cj, hj - half);
        }
        __builtin_amdgcn_wave_barrier();
        float* a = centroids + (long long)ci * d;
        float* b = centroids + (long long)cj * d;
        for (int j = lane; j < d; j += 64) {  // a lane only ever touches its own columns: later splits see these writes
            const float v = b[j];
            if (j % 2 == 0) {
                a[j] = v * (1 + EPS);
                b[j] = v * (1 - EPS);
            } else {
                a[j] = v * (1 - EPS);
                b[j] = v * (1 + EPS);
            }
        }
        ++nsplit;
    }
    if (lane == 0 && out_nsplit) *out_nsplit = nsplit;
}

// stats[0] = max |row|^2 (of the stored values), stats[1] = max |lo part of a row|^2: what the one-pass assignment's
// certificate needs of the centroids (non-negative floats order like their bit patterns: atomicMax on the bits)
__global__ __launch_bounds__(256) void km_stats_kernel(const _Float16* __restrict__ rows, long long ld, int dpad, int split,
                                                       const float* __restrict__ norms, int k, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= k) return;
    float e = 0.f;
    if (split)
        for (int t = lane; t < dpad; t += 64) {
            const float v = (float)rows[(long long)j * ld + dpad + t];
            e += v * v;
        }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) e += __shfl_xor(e, m, 64);
    if (lane == 0) {
        atomicMax((unsigned int*)&stats[0], __float_as_uint(norms[j]));
        atomicMax((unsigned int*)&stats[1], __float_as_uint(e));
    }
}

int32_t km_pack_and_stats(const float* centroids, int32_t k, int32_t d, int32_t pack_mode, void* packed_out, float* norms_out,
                          float* stats_out, hipStream_t st) {
    int32_t rc = lvs_pack_rows(centroids, LVS_DTYPE_F32, k, d, pack_mode, 0, packed_out, norms_out, st);
    if (rc != LVS_OK) return rc;
    if (stats_out) {
        const int dpad = (int)lvs_round_up(d, LVS_BK);
        const int split = pack_mode == LVS_PACK_SPLIT;
        LVS_HIP_CHECK(hipMemsetAsync(stats_out, 0, 2 * sizeof(float), st));
        hipLaunchKernelGGL(km_stats_kernel, dim3((unsigned)lvs_ceil_div(k, 4)), dim3(256), 0, st, (const _Float16*)packed_out,
                           (long long)(split ? 2 * dpad : dpad), dpad, split, (const float*)norms_out, k, stats_out);
        LVS_HIP_CHECK(hipGetLastError());
    }
    return LVS_OK;
}
}  // namespace

// ---- exact distance bounds across iterations (Hamerly 2010): which rows need no new search -------------------------------
// Per row i: assign[i], ub[i] >= |x_i - c_assign| and lb[i] <= min over the OTHER centroids of |x_i - c_j| (Euclidean, in
// the rows' scaled domain).  When the centroids move by delta_j, ub grows by delta_assign and lb shrinks by the largest
// delta among the others; while ub < lb (with a margin above float32 noise) the row's nearest centroid provably has not
// changed - faiss's exhaustive search would return the same id - and the row is skipped.  Every other row goes through the
// full search again, which also renews its bounds.  Results are identical to the exhaustive iteration; only work is saved.
namespace {
// delta[j] = |c_new_j - c_old_j| (one wave per centroid)
__global__ __launch_bounds__(256) void km_shift_kernel(const float* __restrict__ c_old, const float* __restrict__ c_new, int k,
                                                       int d, float* __restrict__ delta) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= k) return;
    float acc = 0.f;
    for (int t = lane; t < d; t += 64) {
        const float v = c_new[(long long)j * d + t] - c_old[(long long)j * d + t];
        acc += v * v;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) delta[j] = sqrtf(acc) * (1.0f + 1e-6f);  // rounded up
}
// top2[0] = largest delta, top2[1] = its centroid (as float bits of an int), top2[2] = second largest (one workgroup)
__global__ __launch_bounds__(256) void km_shift_top2_kernel(const float* __restrict__ delta, int k, float* __restrict__ top2) {
    __shared__ float m1[256], m2[256];
    __shared__ int a1[256];
    float b1 = 0.f, b2 = 0.f;
    int i1 = -1;
    for (int j = threadIdx.x; j < k; j += 256) {
        const float v = delta[j];
        if (v > b1 || i1 < 0) {
            b2 = i1 < 0 ? 0.f : b1;
            b1 = v;
            i1 = j;
        } else if (v > b2) {
            b2 = v;
        }
    }
    m1[threadIdx.x] = b1;
    m2[threadIdx.x] = b2;
    a1[threadIdx.x] = i1;
    __syncthreads();
    if (threadIdx.x == 0) {
        float g1 = 0.f, g2 = 0.f;
        int gi = -1;
        for (int t = 0; t < 256; ++t) {
            if (a1[t] < 0) continue;
            if (m1[t] > g1 || gi < 0) {
                g2 = gi < 0 ? m2[t] : fmaxf(g1, m2[t]);
                g1 = m1[t];
                gi = a1[t];
            } else {
                g2 = fmaxf(g2, m1[t]);
            }
        }
        top2[0] = g1;
        top2[1] = __int_as_float(gi);
        top2[2] = g2;
    }
}
// bounds of the rows pos[i] (NULL: rows 0 .. m) from a search result: keys[i] = winner key (score = -dist^2), second[i] =
// runner-up score; err_i = (coef[0] E + coef[1] R) |x| + coef[2] + coef[3] R + coef[4] R^2 is added to the winner's squared
// distance and subtracted from the runner-up's before the roots (the search's own error bound)
__global__ __launch_bounds__(256) void km_bounds_set_kernel(const u64* __restrict__ keys, long long key_stride,
                                                            const float* __restrict__ second, const float* __restrict__ qn,
                                                            const long long* __restrict__ pos, long long m,
                                                            const float* __restrict__ stats, float c0, float c1, float c2,
                                                            float c3, float c4, long long id_offset, int* __restrict__ assign,
                                                            float* __restrict__ ub, float* __restrict__ lb) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float R = sqrtf(stats[0]), E = sqrtf(stats[1]);
    const long long row = pos ? pos[i] : i;
    const u64 kq = keys[i * key_stride];
    const float err = (c0 * E + c1 * R) * sqrtf(qn[i]) + c2 + c3 * R + c4 * R * R;
    const float d1 = kq ? fmaxf(-lvs_unord32((uint32_t)(kq >> 32)), 0.f) : 3.0e38f;
    const float s2 = second ? second[i] : lvs_unord32((uint32_t)(keys[i * key_stride + 1] >> 32));
    const float d2 = fmaxf(-s2, 0.f);  // -inf runner-up (a single centroid) -> +inf
    assign[row] = kq ? (int)((long long)(0xFFFFFFFFu - (uint32_t)(kq & 0xFFFFFFFFull)) - id_offset) : -1;
    ub[row] = sqrtf(d1 + err) * (1.0f + 1e-6f);
    lb[row] = sqrtf(fmaxf(d2 - err, 0.f)) * (1.0f - 1e-6f);
}
// rows whose one-pass winner was NOT certified: the exact (k = 1) search decided them.  exact[i] = its key; approx[i] = the
// one-pass key whose bounds km_bounds_set_kernel has just written for row pos[i].  The upper bound is renewed from the exact
// distance; the lower bound stays valid as it is when the winner is the same centroid (every other centroid's one-pass score
// was <= the runner-up's), and otherwise must also stay below the distance to the one-pass winner, now one of "the others".
__global__ __launch_bounds__(256) void km_bounds_fix_kernel(const u64* __restrict__ approx, const u64* __restrict__ exact,
                                                            const float* __restrict__ qn, const long long* __restrict__ pos,
                                                            long long m, const float* __restrict__ stats, float c0, float c1,
                                                            float c2, float c3, float c4, float e1, float e4, long long id_offset,
                                                            int* __restrict__ assign, float* __restrict__ ub,
                                                            float* __restrict__ lb) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float R = sqrtf(stats[0]), E = sqrtf(stats[1]);
    const long long row = pos ? pos[i] : i;
    const u64 ka = approx[i], ke = exact[i];
    if (!ke) return;
    const float qnorm = sqrtf(qn[i]);
    const float err1 = (c0 * E + c1 * R) * qnorm + c2 + c3 * R + c4 * R * R;  // the one-pass search's error bound
    const float err2 = e1 * R * qnorm + e4 * R * R;                           // float32 rounding of the exact distance
    const float d1e = fmaxf(-lvs_unord32((uint32_t)(ke >> 32)), 0.f);
    assign[row] = (int)((long long)(0xFFFFFFFFu - (uint32_t)(ke & 0xFFFFFFFFull)) - id_offset);
    ub[row] = sqrtf(d1e + err2) * (1.0f + 1e-6f);
    if ((uint32_t)ka != (uint32_t)ke) {  // another centroid than the one-pass winner
        const float d1a = ka ? fmaxf(-lvs_unord32((uint32_t)(ka >> 32)), 0.f) : 0.f;
        lb[row] = fminf(lb[row], sqrtf(fmaxf(d1a - err1, 0.f)) * (1.0f - 1e-6f));
    }
}
// one iteration later: move the bounds by the centroid shifts and list the rows whose nearest centroid may have changed
__global__ __launch_bounds__(256) void km_bounds_step_kernel(const int* __restrict__ assign, float* __restrict__ ub,
                                                             float* __restrict__ lb, const float* __restrict__ delta,
                                                             const float* __restrict__ top2, long long n, long long per_block,
                                                             long long* __restrict__ out_idx,
                                                             unsigned long long* __restrict__ out_count) {
    __shared__ unsigned long long s_base;
    __shared__ unsigned s_count;
    const float g1 = top2[0], g2 = top2[2];
    const int gi = __float_as_int(top2[1]);
    const long long q_begin = (long long)blockIdx.x * per_block;
    const long long q_end = q_begin + per_block < n ? q_begin + per_block : n;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    unsigned mine = 0;
    for (long long q = q_begin + threadIdx.x; q < q_end; q += blockDim.x) {
        const int a = assign[q];
        float u = ub[q], l = lb[q];
        if (a >= 0) {
            u += delta[a];
            l -= a == gi ? g2 : g1;
        }
        ub[q] = u;
        lb[q] = l;
        mine += (a < 0 || !(u * (1.0f + 1e-5f) < l)) ? 1u : 0u;
    }
    if (mine) atomicAdd(&s_count, mine);
    __syncthreads();
    if (s_count == 0) return;
    if (threadIdx.x == 0) {
        s_base = atomicAdd(out_count, (unsigned long long)s_count);
        s_count = 0;
    }
    __syncthreads();
    for (long long q = q_begin + threadIdx.x; q < q_end; q += blockDim.x) {
        const int a = assign[q];
        if (a < 0 || !(ub[q] * (1.0f + 1e-5f) < lb[q])) out_idx[s_base + atomicAdd(&s_count, 1u)] = q;
    }
}
}  // namespace

extern "C" int32_t lvs_kmeans_centroid_shift(const float* c_old, const float* c_new, int32_t k, int32_t d, float* out_delta,
                                             float* out_top2, void* stream) {
    LVS_REQUIRE(k > 0 && d > 0 && c_old && c_new && out_delta && out_top2, "bad arguments");
    LVS_DEVICE_GUARD(stream);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(km_shift_kernel, dim3((unsigned)lvs_ceil_div(k, 4)), dim3(256), 0, st, c_old, c_new, k, d, out_delta);
    hipLaunchKernelGGL(km_shift_top2_kernel, dim3(1), dim3(256), 0, st, (const float*)out_delta, k, out_top2);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_kmeans_bounds_set(const uint64_t* keys, int32_t key_stride, const float* second, const float* q_norms_sq,
                                         const int64_t* positions, int64_t m, const float* corpus_stats, const float* coef5,
                                         int64_t id_offset, int32_t* assign, float* ub, float* lb, void* stream) {
    LVS_REQUIRE(m >= 0 && key_stride >= 1 && (second || key_stride >= 2), "bad arguments");
    if (m == 0) return LVS_OK;
    LVS_REQUIRE(keys && q_norms_sq && corpus_stats && coef5 && assign && ub && lb, "NULL buffer");
    LVS_DEVICE_GUARD(stream);
    hipLaunchKernelGGL(km_bounds_set_kernel, dim3((unsigned)lvs_ceil_div(m, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const u64*)keys, (long long)key_stride, second, q_norms_sq, (const long long*)positions, (long long)m,
                       corpus_stats, coef5[0], coef5[1], coef5[2], coef5[3], coef5[4], (long long)id_offset, assign, ub, lb);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_kmeans_bounds_fix(const uint64_t* approx_keys, const uint64_t* exact_keys, const float* q_norms_sq,
                                         const int64_t* positions, int64_t m, const float* corpus_stats, const float* coef5,
                                         const float* exact_coef2, int64_t id_offset, int32_t* assign, float* ub, float* lb,
                                         void* stream) {
    LVS_REQUIRE(m >= 0, "bad arguments");
    if (m == 0) return LVS_OK;
    LVS_REQUIRE(approx_keys && exact_keys && q_norms_sq && corpus_stats && coef5 && exact_coef2 && assign && ub && lb, "NULL buffer");
    LVS_DEVICE_GUARD(stream);
    hipLaunchKernelGGL(km_bounds_fix_kernel, dim3((unsigned)lvs_ceil_div(m, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const u64*)approx_keys, (const u64*)exact_keys, q_norms_sq, (const long long*)positions, (long long)m,
                       corpus_stats, coef5[0], coef5[1], coef5[2], coef5[3], coef5[4], exact_coef2[0], exact_coef2[1],
                       (long long)id_offset, assign, ub, lb);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_kmeans_bounds_step(const int32_t* assign, float* ub, float* lb, const float* delta, const float* top2,
                                          int64_t n, int64_t* out_idx, uint64_t* out_count, void* stream) {
    LVS_REQUIRE(n >= 0, "bad arguments");
    if (n == 0) return LVS_OK;
    LVS_REQUIRE(assign && ub && lb && delta && top2 && out_idx && out_count, "NULL buffer");
    LVS_DEVICE_GUARD(stream);
    long long per_block = lvs_ceil_div(n, 2048);
    per_block = lvs_round_up(per_block < 1024 ? 1024 : per_block, 256);
    hipLaunchKernelGGL(km_bounds_step_kernel, dim3((unsigned)lvs_ceil_div(n, per_block)), dim3(256), 0, (hipStream_t)stream,
                       assign, ub, lb, delta, top2, (long long)n, per_block, (long long*)out_idx,
                       (unsigned long long*)out_count);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int64_t lvs_kmeans_objective_workspace_bytes(int32_t k) { return k > 0 ? lvs_round_up((int64_t)k * 8, 256) : LVS_EINVAL; }

extern "C" int32_t lvs_kmeans_objective(const float* centroids, const float* sums, const float* counts, int32_t k, int32_t d,
                                        const double* x_norms_sq_sum, double* out_obj, void* workspace,
                                        int64_t workspace_bytes, void* stream) {
    LVS_REQUIRE(k > 0 && d > 0, "bad shape k=%d d=%d", k, d);
    LVS_REQUIRE(centroids && sums && counts && out_obj && workspace, "NULL buffer");
    LVS_REQUIRE(workspace_bytes >= lvs_kmeans_objective_workspace_bytes(k), "workspace too small");
    LVS_DEVICE_GUARD(stream);
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)workspace;
    hipLaunchKernelGGL(km_obj_part_kernel, dim3((unsigned)lvs_ceil_div(k, 4)), dim3(256), 0, st, centroids, sums, counts, k, d, part);
    hipLaunchKernelGGL(km_obj_sum_kernel, dim3(1), dim3(256), 0, st, (const double*)part, k, x_norms_sq_sum, out_obj);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_kmeans_pack_centroids(const float* centroids, int32_t k, int32_t d, int32_t pack_mode, void* packed_out,
                                             float* norms_out, float* stats_out, void* stream) {
    LVS_REQUIRE(k > 0 && d > 0, "bad shape k=%d d=%d", k, d);
    LVS_REQUIRE(centroids && packed_out && norms_out, "NULL buffer");
    LVS_DEVICE_GUARD(stream);
    return km_pack_and_stats(centroids, k, d, pack_mode, packed_out, norms_out, stats_out, (hipStream_t)stream);
}

extern "C" int32_t lvs_kmeans_update_centroids(const float* sums, float* counts, int32_t k, int32_t d, int64_t n_train,
                                               float* centroids, int32_t* out_nsplit, int32_t pack_mode, void* packed_out,
                                               float* norms_out, float* stats_out, void* stream) {
    LVS_REQUIRE(k > 0 && d > 0, "bad shape k=%d d=%d", k, d);
    LVS_REQUIRE(sums && counts && centroids, "NULL buffer");
    LVS_DEVICE_GUARD(stream);
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)k * d;
    const long long blocks = lvs_ceil_div(total, 256);
    hipLaunchKernelGGL(km_update_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, sums,
                       (const float*)counts, total, d, centroids);
    if (n_train > 0)  // faiss: split_clusters right after compute_centroids; n_train <= 0 skips it (caller splits on the host)
    {
        const size_t lds = (640 + (size_t)(k <= KM_SPLIT_LDS_K ? k : 0)) * 4;
        hipLaunchKernelGGL(km_split_kernel, dim3(1), dim3(64), lds, st, d, k, (long long)n_train, counts, centroids, out_nsplit);
    }
    LVS_HIP_CHECK(hipGetLastError());
    if (packed_out) {
        LVS_REQUIRE(norms_out, "norms_out is NULL");
        return km_pack_and_stats(centroids, k, d, pack_mode, packed_out, norms_out, stats_out, st);
    }
    return LVS_OK;
}

extern "C" int32_t lvs_unpack_rows(const void* src, int32_t d, int32_t pack_mode, const int64_t* ids, int64_t n,
                                   int32_t scale_exp, float* dst, void* stream) {
    LVS_REQUIRE(d > 0 && n >= 0, "bad shape n=%lld d=%d", (long long)n, d);
    LVS_REQUIRE(scale_exp >= -100 && scale_exp <= 100, "scale_exp %d out of range", scale_exp);
    LVS_REQUIRE(pack_mode == LVS_PACK_F16 || pack_mode == LVS_PACK_SPLIT, "bad pack_mode %d", pack_mode);
    if (n == 0) return LVS_OK;
    LVS_REQUIRE(src && dst, "NULL buffer");
    LVS_DEVICE_GUARD(stream);
    const int dpad = (int)lvs_round_up(d, LVS_BK);
    const int split = pack_mode == LVS_PACK_SPLIT;
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)lvs_ceil_div(n, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)src, (long long)(split ? 2 * dpad : dpad), d, dpad, split,
                       (const long long*)ids, (long long)n, ldexpf(1.0f, -scale_exp), dst);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

namespace {
int km_bits(int nbins) {
    int b = 1;
    while ((1 << b) < nbins) ++b;
    return b;
}
struct KmSortPlan {
    int nchunks, passes, nbins[2], shift[2];
    uint32_t mask[2];
    int64_t off_counts, off_seg, off_rows_a, off_rows_b, off_offs, total;
};
bool km_sort_plan(int64_t n, int32_t k, KmSortPlan& p) {
    if (n < 0 || k <= 0 || n >= 0xFFFFFFFFll) return false;
    p.nchunks = (int)lvs_ceil_div(n > 0 ? n : 1, KM_CHUNK);
    if (k + 1 <= KM_MAX_BINS) {
        p.passes = 1;
        p.nbins[0] = k + 1;
        p.shift[0] = 0;
        p.mask[0] = 0xFFFFFFFFu;
        p.nbins[1] = 0;
        p.shift[1] = 0;
        p.mask[1] = 0;
    } else {  // two stable passes: low 12 bits, then the rest
        p.passes = 2;
        p.nbins[0] = 4096;
        p.shift[0] = 0;
        p.mask[0] = 4095u;
        p.nbins[1] = (k >> 12) + 1;
        p.shift[1] = 12;
        p.mask[1] = 0xFFFFFFFFu;
        if (p.nbins[1] > KM_MAX_BINS) return false;  // k >= 2^26.5: not a k-means anyone runs
    }
    const int64_t maxbins = p.nbins[0] > p.nbins[1] ? p.nbins[0] : p.nbins[1];
    int64_t off = 0;
    p.off_counts = off;
    off += lvs_round_up(maxbins * p.nchunks * 4, 256);
    p.off_seg = off;
    off += lvs_round_up(lvs_ceil_div(maxbins * p.nchunks, KM_SCAN_SEG) * 4 + 4, 256);
    p.off_rows_a = off;
    off += lvs_round_up(n * 4, 256);
    p.off_rows_b = off;
    off += p.passes > 1 ? lvs_round_up(n * 4, 256) : 0;
    p.off_offs = off;
    off += lvs_round_up((int64_t)(k + 2) * 4, 256);
    p.total = off;
    return true;
}

// rows bucketed by centroid: *rows_out [n] = row numbers, bucket after bucket, ascending inside a bucket; *offs_out [k + 1]
template <typename KeyT>
int32_t km_bucket_rows(const KeyT* assign, int64_t n, int32_t k, int64_t id_offset, const KmSortPlan& p, char* w,
                       hipStream_t st, const uint32_t** rows_out, const uint32_t** offs_out) {
    uint32_t* counts = (uint32_t*)(w + p.off_counts);
    uint32_t* seg = (uint32_t*)(w + p.off_seg);
    uint32_t* rows[2] = {(uint32_t*)(w + p.off_rows_a), (uint32_t*)(w + p.off_rows_b)};
    uint32_t* offs = (uint32_t*)(w + p.off_offs);
    const uint32_t* order = nullptr;
    for (int ps = 0; ps < p.passes; ++ps) {
        const int nbins = p.nbins[ps];
        const size_t lds = (size_t)nbins * 4;
        const long long total = (long long)nbins * p.nchunks;
        const int nseg = (int)lvs_ceil_div(total, KM_SCAN_SEG);
        const bool last = ps + 1 == p.passes;
        hipLaunchKernelGGL((km_count_kernel<KeyT>), dim3((unsigned)p.nchunks), dim3(256), lds, st, assign, order,
                           (long long)n, k, (long long)id_offset, p.shift[ps], p.mask[ps], nbins, p.nchunks, counts);
        hipLaunchKernelGGL(km_scan1_kernel, dim3((unsigned)nseg), dim3(256), 0, st, counts, total, seg);
        hipLaunchKernelGGL(km_scan2_kernel, dim3(1), dim3(256), 0, st, seg, nseg);
        hipLaunchKernelGGL(km_scan3_kernel, dim3((unsigned)lvs_ceil_div(total, 256)), dim3(256), 0, st, counts, total,
                           (const uint32_t*)seg, p.nchunks, (p.passes == 1 && last) ? offs : (uint32_t*)nullptr, k);
        hipLaunchKernelGGL((km_scatter_kernel<KeyT>), dim3((unsigned)p.nchunks), dim3(256), lds, st, assign, order,
                           (long long)n, k, (long long)id_offset, p.shift[ps], p.mask[ps], nbins, km_bits(nbins),
                           p.nchunks, (const uint32_t*)counts, rows[ps]);
        order = rows[ps];
    }
    if (p.passes > 1)
        hipLaunchKernelGGL((km_bounds_kernel<KeyT>), dim3((unsigned)lvs_ceil_div(n + 1, 256)), dim3(256), 0, st, assign,
                           order, (long long)n, k, (long long)id_offset, offs);
    LVS_HIP_CHECK(hipGetLastError());
    *rows_out = order;
    *offs_out = offs;
    return LVS_OK;
}

template <typename KeyT>
int32_t km_accumulate(const void* x, int64_t n, int32_t d, int32_t pack_mode, const KeyT* assign, int64_t id_offset,
                      int32_t k, float* sums, float* counts, void* workspace, int64_t workspace_bytes, void* stream) {
    LVS_REQUIRE(n >= 0 && d > 0 && k > 0, "bad shape n=%lld d=%d k=%d", (long long)n, d, k);
    LVS_REQUIRE(pack_mode == LVS_PACK_F16 || pack_mode == LVS_PACK_SPLIT, "bad pack_mode");
    LVS_REQUIRE(n < 0xFFFFFFFFll, "n must be below 2^32");
    if (n == 0) return LVS_OK;
    LVS_REQUIRE(x && assign && sums && counts && workspace, "NULL buffer");
    KmSortPlan p;
    LVS_REQUIRE(km_sort_plan(n, k, p), "k=%d is beyond the bucket sort", k);
    if (workspace_bytes < p.total) {
        lvs_set_error("workspace too small: need %lld bytes", (long long)p.total);
        return LVS_ENOMEM;
    }
    LVS_DEVICE_GUARD(stream);
    hipStream_t st = (hipStream_t)stream;
    static LvsPerDeviceOnce attr_count, attr_scatter;  // > 64 KB of dynamic LDS needs the attribute (per device)
    {
        int dev = 0;
        LVS_HIP_CHECK(hipGetDevice(&dev));
        const size_t lds = (size_t)KM_MAX_BINS * 4;
        if (!attr_count.done(dev, lds)) {
            LVS_HIP_CHECK(hipFuncSetAttribute((const void*)km_count_kernel<KeyT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_count.set(dev, lds);
        }
        if (!attr_scatter.done(dev, lds)) {
            LVS_HIP_CHECK(hipFuncSetAttribute((const void*)km_scatter_kernel<KeyT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_scatter.set(dev, lds);
        }
    }
    const uint32_t *rows = nullptr, *offs = nullptr;
    const int32_t rc = km_bucket_rows<KeyT>(assign, n, k, id_offset, p, (char*)workspace, st, &rows, &offs);
    if (rc != LVS_OK) return rc;
    const int dpad = (int)lvs_round_up(d, LVS_BK);
    const long long ld = pack_mode == LVS_PACK_SPLIT ? 2 * dpad : dpad;
    // column slices per bucket: two dimensions per lane (128 columns per wave; d = 768: six waves per bucket).  Measured neutral
    // against eight dimensions per lane on its own (profiles/r05s_km_reduce_dpl_sweep.log), but at 82 instead of 148 VGPRs a
    // wave of this kernel fits on a SIMD NEXT TO the two waves of the assignment kernel (2 x 199 + 82 <= 512), which is what lets
    // the sums of one half of the rows run under the assignment of the other half (lotus_amd/cluster.py)
    int dpl = 2;
    if (lvs_tune_set("LVS_KM_DPL")) {
        const int v = (int)lvs_tune("LVS_KM_DPL", 0);
        if (v == 2 || v == 4 || v == 8) dpl = v;
    }
    const dim3 grid((unsigned)k, (unsigned)lvs_ceil_div(dpad, 64 * dpl));
#define LVS_KM_REDUCE(SP, DP)                                                                                              \
    hipLaunchKernelGGL((km_reduce_kernel<SP, DP>), grid, dim3(64), 0, st, (const _Float16*)x, ld, d, dpad, rows, offs, sums, \
                       counts)
    if (pack_mode == LVS_PACK_SPLIT) {
        if (dpl == 2) LVS_KM_REDUCE(1, 2);
        else if (dpl == 4) LVS_KM_REDUCE(1, 4);
        else LVS_KM_REDUCE(1, 8);
    } else {
        if (dpl == 2) LVS_KM_REDUCE(0, 2);
        else if (dpl == 4) LVS_KM_REDUCE(0, 4);
        else LVS_KM_REDUCE(0, 8);
    }
#undef LVS_KM_REDUCE
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}
}  // namespace

extern "C" int64_t lvs_kmeans_accumulate_workspace_bytes(int64_t n, int32_t k) {
    KmSortPlan p;
    if (!km_sort_plan(n, k, p)) return LVS_EINVAL;
    return p.total;
}

extern "C" int32_t lvs_kmeans_accumulate(const void* x, int64_t n, int32_t d, int32_t pack_mode, const int64_t* assign,
                                         int32_t k, float* sums, float* counts, void* workspace, int64_t workspace_bytes,
                                         void* stream) {
    return km_accumulate<long long>(x, n, d, pack_mode, (const long long*)assign, 0, k, sums, counts, workspace,
                                    workspace_bytes, stream);
}

extern "C" int32_t lvs_kmeans_accumulate_keys(const void* x, int64_t n, int32_t d, int32_t pack_mode, const uint64_t* keys,
                                              int64_t id_offset, int32_t k, float* sums, float* counts, void* workspace,
                                              int64_t workspace_bytes, void* stream) {
    return km_accumulate<u64>(x, n, d, pack_mode, (const u64*)keys, id_offset, k, sums, counts, workspace, workspace_bytes,
                              stream);
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

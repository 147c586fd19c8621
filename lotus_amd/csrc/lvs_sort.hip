// Full-row ranking for callers that ask for K = N (lotus/sem_ops/sem_dedup.py:45, sem_filter.py:491-497,
// sem_join.py:367, sem_topk.py:787): score rows from the tile kernel (lvs_scores) are ranked best-first, one row per query.
//
// Hand-written segmented LSD radix sort (round 2 called rocPRIM's segmented_radix_sort_keys_desc): four stable counting
// passes over the 8-bit digits of the scores' order keys, largest digit value first.  Ids grow with the column, and a stable
// sort keeps the columns of equal scores in their original order, so sorting the 32-bit order keys alone yields the
// (score best-first, id ascending) order of the 64-bit result keys.  A pass works on chunks of RS_CHUNK consecutive elements
// of ONE row, one workgroup each:
//   rs_count    counts[row][digit][chunk] (histogram in LDS)
//   rs_scan1-3  exclusive scan over counts in exactly that order - which makes every row's output range start at row * nb
//               and the digits descend inside it: the segmentation costs nothing
//   rs_scatter  element -> its position, in column order inside a chunk (lanes rank themselves among the lanes with the
//               same digit by ballots over its bits, the waves of a tile take turns)
// HBM-bound: a pass reads 8 B and writes 8 B per element (+ 4 B for the counting read).
#include <cstring>

#include "lvs_common.h"

namespace {
constexpr int RS_CHUNK = 4096;     // elements per workgroup and pass
constexpr int RS_SCAN_SEG = 2048;  // counters per workgroup of the scan

__device__ inline uint32_t rs_digit(uint32_t ord, int shift) { return 255u - ((ord >> shift) & 255u); }  // descending

// first pass reads the scores themselves (key = order key of the float, value = column), later passes the ping-pong buffers
template <bool FIRST>
__device__ inline void rs_load(const float* __restrict__ scores, long long ld, const uint32_t* __restrict__ keys,
                               const uint32_t* __restrict__ vals, long long row, long long col, long long nb, uint32_t& k,
                               uint32_t& v) {
    if (FIRST) {
        k = lvs_ord32(scores[row * ld + col]);
        v = (uint32_t)col;
    } else {
        k = keys[row * nb + col];
        v = vals[row * nb + col];
    }
}

template <bool FIRST>
__global__ __launch_bounds__(256) void rs_count_kernel(const float* __restrict__ scores, long long ld,
                                                       const uint32_t* __restrict__ keys, long long nb, int cpr, int shift,
                                                       uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const long long row = blockIdx.x / cpr;
    const int piece = blockIdx.x % cpr;
    const long long c0 = (long long)piece * RS_CHUNK, c1 = c0 + RS_CHUNK < nb ? c0 + RS_CHUNK : nb;
    for (long long c = c0 + threadIdx.x; c < c1; c += 256) {
        const uint32_t k = FIRST ? lvs_ord32(scores[row * ld + c]) : keys[row * nb + c];
        atomicAdd(&hist[rs_digit(k, shift)], 1u);
    }
    __syncthreads();
    counts[(row * 256 + threadIdx.x) * cpr + piece] = hist[threadIdx.x];
}

// exclusive scan, three launches: per-segment scan + segment totals, scan of the totals (one workgroup), add
__global__ __launch_bounds__(256) void rs_scan1_kernel(uint32_t* __restrict__ v, long long total, uint32_t* __restrict__ seg_sum) {
    __shared__ uint32_t part[256];
    const long long base = (long long)blockIdx.x * RS_SCAN_SEG + (long long)threadIdx.x * (RS_SCAN_SEG / 256);
    uint32_t loc[RS_SCAN_SEG / 256], sum = 0;
#pragma unroll
    for (int i = 0; i < RS_SCAN_SEG / 256; ++i) {
        loc[i] = base + i < total ? v[base + i] : 0u;
        sum += loc[i];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const uint32_t add = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
#pragma unroll
    for (int i = 0; i < RS_SCAN_SEG / 256; ++i) {
        if (base + i < total) v[base + i] = run;
        run += loc[i];
    }
    if (threadIdx.x == 255) seg_sum[blockIdx.x] = part[255];
}
__global__ __launch_bounds__(256) void rs_scan2_kernel(uint32_t* __restrict__ seg_sum, long long nseg) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long long s0 = 0; s0 < nseg; s0 += 256) {
        const long long i = s0 + threadIdx.x;
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
__global__ __launch_bounds__(256) void rs_scan3_kernel(uint32_t* __restrict__ v, long long total,
                                                       const uint32_t* __restrict__ seg_sum) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) v[i] += seg_sum[i / RS_SCAN_SEG];
}

// LAST pass writes the result keys instead of the ping-pong buffers
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(256) void rs_scatter_kernel(const float* __restrict__ scores, long long ld,
                                                         const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                         long long nb, int cpr, int shift, const uint32_t* __restrict__ counts,
                                                         uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                         u64* __restrict__ result, long long id_offset) {
    __shared__ uint32_t pos[256];  // next output position (inside the whole [nq * nb] array) of every digit for this chunk
    const long long row = blockIdx.x / cpr;
    const int piece = blockIdx.x % cpr;
    pos[threadIdx.x] = counts[(row * 256 + threadIdx.x) * cpr + piece];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long c0 = (long long)piece * RS_CHUNK, c1 = c0 + RS_CHUNK < nb ? c0 + RS_CHUNK : nb;
    for (long long t0 = c0; t0 < c1; t0 += 256) {
        const long long c = t0 + threadIdx.x;
        const bool live = c < c1;
        uint32_t k = 0, v = 0, dgt = 0;
        if (live) {
            rs_load<FIRST>(scores, ld, keys, vals, row, c, nb, k, v);
            dgt = rs_digit(k, shift);
        }
        u64 peers = __builtin_amdgcn_ballot_w64(live);  // lanes of this wave holding the same digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const u64 m = __builtin_amdgcn_ballot_w64(live && ((dgt >> b) & 1u));
            peers &= ((dgt >> b) & 1u) ? m : ~m;
        }
        const int rank = __popcll(peers & ((1ull << lane) - 1ull));
        const int cnt = __popcll(peers);
        uint32_t base = 0;
        for (int w = 0; w < 4; ++w) {  // waves take turns: columns of wave w come before those of wave w + 1
            if (wave == w && live && rank == 0) {
                base = pos[dgt];
                pos[dgt] = base + (uint32_t)cnt;
            }
            __syncthreads();
        }
        const int leader = live ? __ffsll((long long)peers) - 1 : lane;
        base = __shfl(base, leader, 64);
        if (live) {
            const long long o = (long long)base + rank;
            if (LAST) {
                result[o] = ((u64)k << 32) | (u64)(0xFFFFFFFFu - (uint32_t)(v + id_offset));
            } else {
                keys_out[o] = k;
                vals_out[o] = v;
            }
        }
    }
}

struct RsPlan {
    int cpr;            // chunks per row
    long long nchunks;  // workgroups per pass
    long long ncount;   // counters = nq * 256 * cpr
    long long nseg;     // scan segments
    int64_t off_counts, off_seg, off_k0, off_v0, off_k1, off_v1, total;
};
bool rs_plan(int64_t nq, int64_t nb, RsPlan& p) {
    if (nq < 0 || nb < 0 || nq * nb >= 0xFFFFFFFFll) return false;
    p.cpr = (int)lvs_ceil_div(nb > 0 ? nb : 1, RS_CHUNK);
    p.nchunks = nq * p.cpr;
    if (p.nchunks >= 0x7FFFFFFFll) return false;
    p.ncount = nq * 256 * p.cpr;
    p.nseg = lvs_ceil_div(p.ncount > 0 ? p.ncount : 1, RS_SCAN_SEG);
    int64_t off = 0;
    p.off_counts = off;
    off += lvs_round_up(p.ncount * 4, 256);
    p.off_seg = off;
    off += lvs_round_up(p.nseg * 4 + 4, 256);
    const int64_t buf = lvs_round_up(nq * nb * 4, 256);
    p.off_k0 = off;
    off += buf;
    p.off_v0 = off;
    off += buf;
    p.off_k1 = off;
    off += buf;
    p.off_v1 = off;
    off += buf;
    p.total = off;
    return true;
}
}  // namespace

extern "C" int64_t lvs_sort_rows_workspace_bytes(int64_t nq, int64_t nb) {
    RsPlan p;
    if (!rs_plan(nq, nb, p)) return LVS_EINVAL;
    if (nq == 0 || nb == 0) return 0;
    return p.total;
}

extern "C" int32_t lvs_sort_rows_desc(const float* scores, int64_t nq, int64_t nb, int64_t ld, int64_t id_offset,
                                      uint64_t* out_keys, void* workspace, int64_t workspace_bytes, void* stream) {
    LVS_REQUIRE(nq >= 0 && nb >= 0 && ld >= nb, "bad shape");
    LVS_REQUIRE(nq * nb < 0xFFFFFFFFll, "nq * nb must stay below 2^32 (got %lld)", (long long)(nq * nb));
    LVS_REQUIRE(id_offset >= 0 && id_offset + nb < 0xFFFFFFFFll, "ids must stay below 2^32-1");
    if (nq == 0 || nb == 0) return LVS_OK;
    LVS_REQUIRE(scores && out_keys && workspace, "NULL buffer");
    RsPlan p;
    LVS_REQUIRE(rs_plan(nq, nb, p), "shape beyond the row sort");
    if (workspace_bytes < p.total) {
        lvs_set_error("workspace too small: need %lld bytes", (long long)p.total);
        return LVS_ENOMEM;
    }
    LVS_DEVICE_GUARD(stream);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)workspace;
    uint32_t* counts = (uint32_t*)(w + p.off_counts);
    uint32_t* seg = (uint32_t*)(w + p.off_seg);
    uint32_t* kb[2] = {(uint32_t*)(w + p.off_k0), (uint32_t*)(w + p.off_k1)};
    uint32_t* vb[2] = {(uint32_t*)(w + p.off_v0), (uint32_t*)(w + p.off_v1)};
    const dim3 grid((unsigned)p.nchunks), block(256);
    const unsigned sgrid = (unsigned)p.nseg, agrid = (unsigned)lvs_ceil_div(p.ncount, 256);
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 8 * pass;
        const uint32_t *kin = pass ? kb[(pass - 1) & 1] : nullptr, *vin = pass ? vb[(pass - 1) & 1] : nullptr;
        uint32_t *kout = kb[pass & 1], *vout = vb[pass & 1];
        if (pass == 0)
            hipLaunchKernelGGL((rs_count_kernel<true>), grid, block, 0, st, scores, (long long)ld, kin, (long long)nb, p.cpr,
                               shift, counts);
        else
            hipLaunchKernelGGL((rs_count_kernel<false>), grid, block, 0, st, scores, (long long)ld, kin, (long long)nb, p.cpr,
                               shift, counts);
        hipLaunchKernelGGL(rs_scan1_kernel, dim3(sgrid), block, 0, st, counts, (long long)p.ncount, seg);
        hipLaunchKernelGGL(rs_scan2_kernel, dim3(1), block, 0, st, seg, (long long)p.nseg);
        hipLaunchKernelGGL(rs_scan3_kernel, dim3(agrid), block, 0, st, counts, (long long)p.ncount, (const uint32_t*)seg);
        if (pass == 0)
            hipLaunchKernelGGL((rs_scatter_kernel<true, false>), grid, block, 0, st, scores, (long long)ld, kin, vin,
                               (long long)nb, p.cpr, shift, (const uint32_t*)counts, kout, vout, (u64*)nullptr,
                               (long long)id_offset);
        else if (pass < 3)
            hipLaunchKernelGGL((rs_scatter_kernel<false, false>), grid, block, 0, st, scores, (long long)ld, kin, vin,
                               (long long)nb, p.cpr, shift, (const uint32_t*)counts, kout, vout, (u64*)nullptr,
                               (long long)id_offset);
        else
            hipLaunchKernelGGL((rs_scatter_kernel<false, true>), grid, block, 0, st, scores, (long long)ld, kin, vin,
                               (long long)nb, p.cpr, shift, (const uint32_t*)counts, kout, vout, (u64*)out_keys,
                               (long long)id_offset);
    }
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

// Full-row ranking for callers that ask for K = N (lotus/sem_ops/sem_dedup.py:45, sem_filter.py:491-497,
// sem_join.py:367, sem_topk.py:787): score rows from the tile kernel (lvs_scores) are turned into result keys and
// every row is sorted best-first with rocPRIM's segmented radix sort (one segment per query).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "lvs_common.h"

namespace {
__global__ __launch_bounds__(256) void make_keys_kernel(const float* __restrict__ scores, long long nq, long long nb,
                                                        long long ld, long long id_offset, u64* __restrict__ keys,
                                                        unsigned* __restrict__ offsets) {
    const long long total = nq * nb;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long q = i / nb, j = i - q * nb;
        keys[i] = lvs_pack_key(scores[q * ld + j], (uint32_t)(j + id_offset));
    }
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q <= nq;
         q += (long long)gridDim.x * blockDim.x)
        offsets[q] = (unsigned)(q * nb);
}
}  // namespace

static size_t sort_temp_bytes(int64_t nq, int64_t nb) {
    size_t tmp = 0;
    u64* nil = nullptr;
    unsigned* off = nullptr;
    (void)rocprim::segmented_radix_sort_keys_desc(nullptr, tmp, nil, nil, (unsigned)(nq * nb), (unsigned)nq, off,
                                                  off + 1, 0u, 64u);
    return tmp;
}

extern "C" int64_t lvs_sort_rows_workspace_bytes(int64_t nq, int64_t nb) {
    if (nq < 0 || nb < 0 || nq * nb >= 0xFFFFFFFFll) return LVS_EINVAL;
    if (nq == 0 || nb == 0) return 0;
    return lvs_round_up((int64_t)sort_temp_bytes(nq, nb), 256) + lvs_round_up(nq * nb * 8, 256) +
           lvs_round_up((nq + 1) * 4, 256);
}

extern "C" int32_t lvs_sort_rows_desc(const float* scores, int64_t nq, int64_t nb, int64_t ld, int64_t id_offset,
                                      uint64_t* out_keys, void* workspace, int64_t workspace_bytes, void* stream) {
    LVS_REQUIRE(nq >= 0 && nb >= 0 && ld >= nb, "bad shape");
    LVS_REQUIRE(nq * nb < 0xFFFFFFFFll, "nq * nb must stay below 2^32 (got %lld)", (long long)(nq * nb));
    LVS_REQUIRE(id_offset >= 0 && id_offset + nb < 0xFFFFFFFFll, "ids must stay below 2^32-1");
    if (nq == 0 || nb == 0) return LVS_OK;
    LVS_REQUIRE(scores && out_keys && workspace, "NULL buffer");
    const int64_t need = lvs_sort_rows_workspace_bytes(nq, nb);
    if (workspace_bytes < need) {
        lvs_set_error("workspace too small: need %lld bytes", (long long)need);
        return LVS_ENOMEM;
    }
    LVS_DEVICE_GUARD(stream);
    hipStream_t st = (hipStream_t)stream;
    size_t tmp = sort_temp_bytes(nq, nb);
    char* w = (char*)workspace;
    void* d_tmp = w;
    w += lvs_round_up((int64_t)tmp, 256);
    u64* keys_in = (u64*)w;
    w += lvs_round_up(nq * nb * 8, 256);
    unsigned* offs = (unsigned*)w;
    hipLaunchKernelGGL(make_keys_kernel, dim3(2048), dim3(256), 0, st, scores, (long long)nq, (long long)nb,
                       (long long)ld, (long long)id_offset, keys_in, offs);
    LVS_HIP_CHECK(rocprim::segmented_radix_sort_keys_desc(d_tmp, tmp, keys_in, (u64*)out_keys, (unsigned)(nq * nb),
                                                          (unsigned)nq, offs, offs + 1, 0u, 64u, st));
    return LVS_OK;
}

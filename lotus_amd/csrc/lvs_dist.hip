// A row-sharded search with its exchange steps INSIDE the library (`lvs_search` "includes RCCL gather+merge";
// the split of lotus/sem_ops/sem_sim_join.py:132-134 over the GPUs of a node, BASELINE configs[2]).  One rank = one GPU = one
// caller of these entry points; what crosses the ranks is exactly what lotus_amd/vs.py + _dist.py exchange from Python:
//   1. [optional] every shard's sample scores (lvs_flat_search_seed_scores, [tiles][nq] float32)   - one all-gather
//   2. every shard's candidate lists ([nq][k] uint64 keys carrying global ids)                     - one all-gather
// and a per-query merge (lvs_merge_keys) on every rank.  Two flavours:
//   lvs_search_sharded        the all-gather is a caller-supplied function (any transport; a thread barrier in the tests);
//   lvs_search_sharded_rccl   the all-gather is RCCL's ncclAllGather on the caller's ncclComm_t.  librccl is resolved at run
//                             time (dlopen of the process's librccl.so.1), so the library itself keeps loading - and every
//                             single-GPU entry point keeps working - on a host without RCCL.
// Host code only: no kernel lives in this file.
#include <dlfcn.h>
#include <string.h>

#include "lvs_common.h"

namespace {

struct ShardedPlan {
    int tiles;  // sample tiles per shard actually exchanged (0: no pooled thresholds)
    int64_t off_seed_local, off_seed_all, off_keys_local, off_keys_all, off_search, search_bytes, total;
};

bool sharded_plan(int32_t nranks, int64_t nq, int64_t nb_local, int32_t d, int32_t k, int32_t xb_pack, int32_t xq_pack,
                  int32_t seed_tiles, ShardedPlan& p) {
    if (nranks < 1 || nq < 0 || nb_local < 0 || d <= 0 || k < 1 || seed_tiles < 0) return false;
    // pooled thresholds: fp16 rows on both sides, single-pass lists, more than one shard - decided from the ARGUMENTS, so
    // every rank takes the same path (a certified fp32 search reads a short list as "this shard holds no more rows")
    p.tiles = (nranks > 1 && xb_pack == LVS_PACK_F16 && xq_pack == LVS_PACK_F16 && k <= 56 && seed_tiles * (int64_t)nranks >= k)
                  ? seed_tiles : 0;
    p.search_bytes = lvs_flat_search_workspace_bytes(nq, nb_local > 0 ? nb_local : 1, d, k, xb_pack, xq_pack);
    if (p.search_bytes < 0) return false;
    int64_t off = 0;
    p.off_seed_local = off;
    off += lvs_round_up((int64_t)p.tiles * nq * 4, 256);
    p.off_seed_all = off;
    off += lvs_round_up((int64_t)p.tiles * nq * 4 * nranks, 256);
    p.off_keys_local = off;
    off += lvs_round_up(nq * (int64_t)k * 8, 256);
    p.off_keys_all = off;
    off += lvs_round_up(nq * (int64_t)k * 8 * nranks, 256);
    p.off_search = off;
    off += lvs_round_up(p.search_bytes, 256);
    p.total = off + 256;
    return true;
}

// ---- RCCL, resolved at run time ----
typedef int (*nccl_all_gather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*nccl_comm_count_t)(const void*, int*);
typedef const char* (*nccl_error_string_t)(int);
struct Rccl {
    void* handle = nullptr;
    nccl_all_gather_t all_gather = nullptr;
    nccl_comm_count_t comm_count = nullptr;
    nccl_error_string_t error_string = nullptr;
};
Rccl g_bound;  // set by lvs_rccl_bind: the caller's own RCCL entry points take precedence
const Rccl* rccl() {
    if (g_bound.all_gather && g_bound.comm_count) return &g_bound;
    // 1. whatever RCCL the process already exposes globally (a host linked against librccl), 2. the system's librccl.so.1,
    // 3. a bundled librccl.so (PyTorch wheels ship one without a version suffix)
    static Rccl r = [] {
        Rccl x;
        x.all_gather = (nccl_all_gather_t)dlsym(RTLD_DEFAULT, "ncclAllGather");
        x.comm_count = (nccl_comm_count_t)dlsym(RTLD_DEFAULT, "ncclCommCount");
        x.error_string = (nccl_error_string_t)dlsym(RTLD_DEFAULT, "ncclGetErrorString");
        if (x.all_gather && x.comm_count) return x;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            x.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.handle) break;
        }
        if (x.handle) {
            x.all_gather = (nccl_all_gather_t)dlsym(x.handle, "ncclAllGather");
            x.comm_count = (nccl_comm_count_t)dlsym(x.handle, "ncclCommCount");
            x.error_string = (nccl_error_string_t)dlsym(x.handle, "ncclGetErrorString");
        }
        return x;
    }();
    return (r.all_gather && r.comm_count) ? &r : nullptr;
}
int32_t rccl_all_gather(void* comm, const void* send, void* recv, int64_t bytes, void* stream) {
    const Rccl* r = rccl();
    if (!r) return LVS_EDEVICE;
    const int rc = r->all_gather(send, recv, (size_t)bytes, /* ncclInt8 */ 0, comm, (hipStream_t)stream);
    if (rc != 0) {
        lvs_set_error("ncclAllGather failed: %s", r->error_string ? r->error_string(rc) : "?");
        return LVS_EDEVICE;
    }
    return LVS_OK;
}

}  // namespace

extern "C" int64_t lvs_search_sharded_workspace_bytes(int32_t nranks, int64_t nq, int64_t nb_local, int32_t d, int32_t k,
                                                      int32_t xb_pack, int32_t xq_pack, int32_t seed_tiles) {
    ShardedPlan p;
    if (!sharded_plan(nranks, nq, nb_local, d, k, xb_pack, xq_pack, seed_tiles, p)) return LVS_EINVAL;
    return p.total;
}

extern "C" int32_t lvs_search_sharded(lvs_all_gather_fn all_gather, void* all_gather_ctx, int32_t nranks, const void* xb,
                                      int32_t xb_pack, int64_t nb_local, const void* xq, int32_t xq_pack, int64_t nq, int32_t d,
                                      int32_t metric, int32_t k, const float* xb_norms_sq, const float* xq_norms_sq,
                                      int64_t id_offset, int32_t seed_tiles, uint64_t* out_keys, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
    ShardedPlan p;
    LVS_REQUIRE(sharded_plan(nranks, nq, nb_local, d, k, xb_pack, xq_pack, seed_tiles, p),
                "bad shape nranks=%d nq=%lld nb_local=%lld d=%d k=%d pack=%d/%d seed_tiles=%d", nranks, (long long)nq,
                (long long)nb_local, d, k, xb_pack, xq_pack, seed_tiles);
    LVS_REQUIRE(nranks == 1 || all_gather, "an all-gather function is required for more than one rank");
    if (nq == 0) return LVS_OK;
    LVS_REQUIRE(out_keys, "NULL output");
    if (!workspace || workspace_bytes < p.total) {
        lvs_set_error("workspace too small: need %lld bytes, got %lld", (long long)p.total, (long long)workspace_bytes);
        return LVS_ENOMEM;
    }
    // Everything that can fail on ONE rank only is checked before the first exchange: a rank that returned early would leave
    // its peers waiting inside the all-gather.  (An error from the all-gather itself, or from a launch, still leaves the
    // communicator in an undefined collective state - include/lotus_hip.h.)
    LVS_REQUIRE(xq, "NULL queries");
    LVS_REQUIRE(nb_local == 0 || xb, "NULL rows");
    LVS_REQUIRE(metric == LVS_METRIC_IP || metric == LVS_METRIC_L2, "bad metric %d", metric);
    LVS_REQUIRE(metric != LVS_METRIC_L2 || (xq_norms_sq && (nb_local == 0 || xb_norms_sq)), "L2 needs both norm vectors");
    LVS_REQUIRE(k >= 1 && k <= LVS_MAX_K, "k=%d out of range", k);
    LVS_REQUIRE(id_offset >= 0 && id_offset + nb_local < 0xFFFFFFFFll, "ids must stay below 2^32-1");
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    float* seed_all = nullptr;
    if (p.tiles) {  // 1. pooled sample thresholds: every shard scores its first `tiles` tiles, the blocks are all-gathered
        float* seed_local = (float*)(ws + p.off_seed_local);
        seed_all = (float*)(ws + p.off_seed_all);
        int32_t rc = lvs_flat_search_seed_scores(xb, xb_pack, nb_local, xq, xq_pack, nq, d, metric, xb_norms_sq, xq_norms_sq,
                                                 p.tiles, seed_local, stream);
        if (rc != LVS_OK) return rc;
        rc = all_gather(all_gather_ctx, seed_local, seed_all, (int64_t)p.tiles * nq * 4, stream);
        if (rc != LVS_OK) return rc;
    }
    // 2. this shard's lists (global ids through id_offset).  An EMPTY shard (more ranks than row blocks) contributes empty
    //    lists: key 0 = "no candidate", which lvs_merge_keys sorts last
    uint64_t* keys_local = nranks == 1 ? out_keys : (uint64_t*)(ws + p.off_keys_local);
    if (nb_local > 0) {
        const int32_t rc = lvs_flat_search_keys_seeded(xb, xb_pack, nb_local, xq, xq_pack, nq, d, metric, k, xb_norms_sq,
                                                       xq_norms_sq, id_offset, nullptr, seed_all, p.tiles * nranks, keys_local,
                                                       ws + p.off_search, p.search_bytes, stream);
        if (rc != LVS_OK) return rc;
    } else {
        LVS_DEVICE_GUARD(stream);
        LVS_HIP_CHECK(hipMemsetAsync(keys_local, 0, (size_t)nq * k * 8, st));
    }
    if (nranks == 1) return LVS_OK;
    // 3. one all-gather of the [nq][k] key lists, merged per query on every rank
    uint64_t* keys_all = (uint64_t*)(ws + p.off_keys_all);
    int32_t rc = all_gather(all_gather_ctx, keys_local, keys_all, nq * (int64_t)k * 8, stream);
    if (rc != LVS_OK) return rc;
    return lvs_merge_keys(keys_all, nranks, nq, k, out_keys, stream);
}

extern "C" int32_t lvs_rccl_available(void) { return rccl() ? 1 : 0; }

extern "C" int32_t lvs_rccl_bind(void* nccl_all_gather, void* nccl_comm_count, void* nccl_get_error_string) {
    LVS_REQUIRE((nccl_all_gather == nullptr) == (nccl_comm_count == nullptr), "pass both functions, or NULL for both to unbind");
    g_bound.all_gather = (nccl_all_gather_t)nccl_all_gather;
    g_bound.comm_count = (nccl_comm_count_t)nccl_comm_count;
    g_bound.error_string = (nccl_error_string_t)nccl_get_error_string;
    return LVS_OK;
}

extern "C" int32_t lvs_search_sharded_rccl(void* nccl_comm, const void* xb, int32_t xb_pack, int64_t nb_local, const void* xq,
                                           int32_t xq_pack, int64_t nq, int32_t d, int32_t metric, int32_t k,
                                           const float* xb_norms_sq, const float* xq_norms_sq, int64_t id_offset,
                                           int32_t seed_tiles, uint64_t* out_keys, void* workspace, int64_t workspace_bytes,
                                           void* stream) {
    const Rccl* r = rccl();
    if (!r) {
        const char* why = dlerror();  // (a second call would return NULL: the first one clears the message)
        lvs_set_error("librccl.so.1 could not be loaded (or lacks ncclAllGather / ncclCommCount): %s", why ? why : "");
        return LVS_EDEVICE;
    }
    LVS_REQUIRE(nccl_comm, "NULL communicator");
    int nranks = 0;
    const int rc = r->comm_count(nccl_comm, &nranks);
    if (rc != 0 || nranks < 1) {
        lvs_set_error("ncclCommCount failed: %s", r->error_string ? r->error_string(rc) : "?");
        return LVS_EDEVICE;
    }
    return lvs_search_sharded(rccl_all_gather, nccl_comm, nranks, xb, xb_pack, nb_local, xq, xq_pack, nq, d, metric, k,
                              xb_norms_sq, xq_norms_sq, id_offset, seed_tiles, out_keys, workspace, workspace_bytes, stream);
}

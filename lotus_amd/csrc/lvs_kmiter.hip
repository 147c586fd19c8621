// One Lloyd iteration of faiss `Clustering::train` (lotus/utils.py:61-62: `faiss.Kmeans(d, k, niter).train(x)`) as ONE C-ABI
// call, and the same with its reduction over the ranks that hold the rows (SURVEY.md 8(b) `lvs_kmeans`, 8(e) row 3): a host
// in any language runs the k-means of `lotus.utils.cluster` as a loop over `lvs_kmeans_iteration[_rccl]`.
//
// The chain it enqueues is the one lotus_amd/cluster.py runs launch by launch for one range of rows - same entry points, same
// order, same arithmetic, so the centroids, objectives and split counts are bit-identical to that path:
//   lvs_nearest3 (one MFMA pass over the hi parts: best / runner-up key, second / third score per row)
//   -> lvs_nearest3_select (certified | pair | open)  -> [host reads ONE number: how many rows are open]
//   -> lvs_resolve_pairs (two exact dot products per pair)  -> open rows: lvs_gather_rows + the exact lvs_flat_search_keys + scatter
//   -> lvs_kmeans_accumulate_keys (counting sort + in-row-order sums)  -> lvs_kmeans_objective
//   -> [all-reduce of sums | counts (float32) and the objective (float64) over the ranks]
//   -> lvs_kmeans_update_centroids (division, faiss's split_clusters replayed on the device, repack + certificate statistics)
// It is the one entry point of the library that synchronises its stream (once per call, to size the exact search of the open
// rows - 0.8 % of the rows on BASELINE configs[4]'s data); nothing is allocated: every temporary lives in the caller's workspace.
#include <dlfcn.h>
#include <math.h>

#include "lvs_common.h"
#include "lvs_tile.h"

namespace {

__global__ __launch_bounds__(256) void scatter_keys_kernel(u64* __restrict__ dst, const int64_t* __restrict__ idx,
                                                           const u64* __restrict__ src, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[idx[i]] = src[i];
}

constexpr int64_t KM_NEAREST3_BUDGET = 2ll << 30;  // bytes of lvs_nearest3 scratch per call (lotus_amd/backend.py NEAREST3_WS_BUDGET)

struct IterPlan {
    int64_t step;      // rows per lvs_nearest3 call
    int64_t cap_open;  // open rows searched exactly per batch
    int64_t ld;        // halfs per packed row of x
    int64_t off_keys2, off_sec, off_third, off_pair, off_open, off_counts, off_sums, off_cnt, off_n3, off_acc, off_obj, off_sub,
        off_subn, off_subk, off_search, n3_bytes, acc_bytes, obj_bytes, search_bytes, total;
};

bool iter_plan(int64_t n, int32_t d, int32_t k, int32_t x_pack, int32_t c_pack, IterPlan& p) {
    if (n < 0 || d <= 0 || k < 1 || k > LVS_NEAREST3_MAX_ROWS) return false;
    if ((x_pack != LVS_PACK_F16 && x_pack != LVS_PACK_SPLIT) || c_pack != LVS_PACK_SPLIT) return false;
    const int64_t per_q = lvs_nearest3_workspace_bytes(1 << 20, k, d) >> 20;
    p.step = (KM_NEAREST3_BUDGET / (per_q > 0 ? per_q : 1)) >> 16 << 16;
    if (p.step < (1 << 16)) p.step = 1 << 16;
    const int64_t nn = n > 0 ? n : 1;
    if (p.step > nn) p.step = nn;
    p.cap_open = nn / 8 > 65536 ? nn / 8 : (nn < 65536 ? nn : 65536);
    p.ld = lvs_packed_ld(d, x_pack);
    p.n3_bytes = lvs_nearest3_workspace_bytes(p.step, k, d);
    p.acc_bytes = lvs_kmeans_accumulate_workspace_bytes(nn, k);
    p.obj_bytes = lvs_kmeans_objective_workspace_bytes(k);
    p.search_bytes = lvs_flat_search_workspace_bytes(p.cap_open, k, d, 1, c_pack, x_pack);
    if (p.n3_bytes < 0 || p.acc_bytes < 0 || p.obj_bytes < 0 || p.search_bytes < 0) return false;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        const int64_t o = off;
        off += lvs_round_up(bytes > 0 ? bytes : 1, 256);
        return o;
    };
    p.off_keys2 = take(nn * 8);
    p.off_sec = take(nn * 4);
    p.off_third = take(nn * 4);
    p.off_pair = take(nn * 8);
    p.off_open = take(nn * 8);
    p.off_counts = take(16);
    p.off_sums = take((int64_t)k * d * 4 + (int64_t)k * 4);  // sums [k][d] | counts [k]: ONE float32 block for the all-reduce
    p.off_cnt = p.off_sums + (int64_t)k * d * 4;
    p.off_n3 = take(p.n3_bytes);
    p.off_acc = take(p.acc_bytes);
    p.off_obj = take(p.obj_bytes);
    p.off_sub = take(p.cap_open * p.ld * 2);
    p.off_subn = take(p.cap_open * 4);
    p.off_subk = take(p.cap_open * 8);
    p.off_search = take(p.search_bytes);
    p.total = off + 256;
    return true;
}

// the one-pass search's error bound per unit of |q| (lotus_amd/backend.py _nearest_coef, squared L2): see lvs_margin_select_stats
void nearest_coef(int32_t x_pack, int32_t d, int32_t exp_sum, float coef[5]) {
    const double dpad = (double)lvs_round_up(d, 64);
    const double qsplit = x_pack == LVS_PACK_SPLIT ? 1.0 : 0.0, c = 4.0;
    coef[0] = (float)c;
    coef[1] = (float)(c * (ldexp(1.0, -11) * qsplit + 8e-6) + ldexp(1.0, -16) * 2.0);
    coef[2] = (float)(1e-6 * ldexp(1.0, exp_sum));
    coef[3] = (float)(qsplit * c * sqrt(dpad) * ldexp(1.0, -25));
    coef[4] = (float)(1e-6 + ldexp(1.0, -16));
}

}  // namespace

extern "C" int64_t lvs_kmeans_iteration_workspace_bytes(int64_t n, int32_t d, int32_t k, int32_t x_pack, int32_t c_pack) {
    IterPlan p;
    if (!iter_plan(n, d, k, x_pack, c_pack, p)) return LVS_EINVAL;
    return p.total;
}

extern "C" int32_t lvs_kmeans_iteration(lvs_all_reduce_fn all_reduce, void* all_reduce_ctx, const void* x, int32_t x_pack,
                                        int64_t n, int32_t d, const float* x_norms_sq, const double* x_norms_sq_sum, int32_t exp_sum,
                                        int32_t k, int64_t n_train_total, float* centroids, int32_t c_pack, void* c_packed,
                                        float* c_norms, float* c_stats, uint64_t* out_keys, double* out_obj, int32_t* out_nsplit,
                                        int64_t* out_host_counts, void* workspace, int64_t workspace_bytes, void* stream) {
    IterPlan p;
    LVS_REQUIRE(iter_plan(n, d, k, x_pack, c_pack, p), "bad shape n=%lld d=%d k=%d pack=%d/%d (centroids must be hi|lo rows, k <= %d)",
                (long long)n, d, k, x_pack, c_pack, LVS_NEAREST3_MAX_ROWS);
    LVS_REQUIRE(centroids && c_packed && c_norms && c_stats && out_obj, "NULL centroid state / objective");
    LVS_REQUIRE(n == 0 || (x && x_norms_sq && out_keys), "NULL rows");
    if (!workspace || workspace_bytes < p.total) {
        lvs_set_error("workspace too small: need %lld bytes, got %lld", (long long)p.total, (long long)workspace_bytes);
        return LVS_ENOMEM;
    }
    LVS_DEVICE_GUARD(stream);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    uint64_t* keys2 = (uint64_t*)(ws + p.off_keys2);
    float* sec = (float*)(ws + p.off_sec);
    float* third = (float*)(ws + p.off_third);
    int64_t* pair_idx = (int64_t*)(ws + p.off_pair);
    int64_t* open_idx = (int64_t*)(ws + p.off_open);
    uint64_t* counts = (uint64_t*)(ws + p.off_counts);
    float* sums = (float*)(ws + p.off_sums);
    float* cnt = (float*)(ws + p.off_cnt);
    float coef[5];
    nearest_coef(x_pack, d, exp_sum, coef);
    const char* xr = (const char*)x;
    int64_t tot_pair = 0, tot_open = 0;

    // ---- assignment: the certified one-pass search, in chunks of rows that keep its scratch inside a fixed budget (results are
    // per row, so chunking changes nothing)
    for (int64_t r0 = 0; r0 < n; r0 += p.step) {
        const int64_t m = n - r0 < p.step ? n - r0 : p.step;
        const void* xq = xr + r0 * p.ld * 2;
        const float* qn = x_norms_sq + r0;
        uint64_t* keys = out_keys + r0;
        int32_t rc = lvs_nearest3(c_packed, c_pack, k, xq, x_pack, m, d, LVS_METRIC_L2, c_norms, qn, 0, keys, keys2 + r0, sec + r0,
                                  third + r0, ws + p.off_n3, p.n3_bytes, stream);
        if (rc != LVS_OK) return rc;
        LVS_HIP_CHECK(hipMemsetAsync(counts, 0, 16, st));
        rc = lvs_nearest3_select(keys, keys2 + r0, sec + r0, third + r0, qn, m, c_stats, coef, pair_idx, open_idx, counts, stream);
        if (rc != LVS_OK) return rc;
        uint64_t hc[2] = {0, 0};  // the call's host round trip (per chunk of rows): the exact search below is sized by it
        LVS_HIP_CHECK(hipMemcpyAsync(hc, counts, 16, hipMemcpyDeviceToHost, st));
        LVS_HIP_CHECK(hipStreamSynchronize(st));
        const int64_t n_pair = (int64_t)hc[0], n_open = (int64_t)hc[1];
        tot_pair += n_pair;
        tot_open += n_open;
        if (n_pair) {
            rc = lvs_resolve_pairs(c_packed, c_pack, xq, x_pack, d, LVS_METRIC_L2, c_norms, qn, 0, pair_idx, counts, n_pair, keys,
                                   keys2 + r0, stream);
            if (rc != LVS_OK) return rc;
        }
        for (int64_t o0 = 0; o0 < n_open; o0 += p.cap_open) {  // three or more centroids inside the bound: the exact search
            const int64_t mo = n_open - o0 < p.cap_open ? n_open - o0 : p.cap_open;
            void* sub = ws + p.off_sub;
            float* subn = (float*)(ws + p.off_subn);
            uint64_t* subk = (uint64_t*)(ws + p.off_subk);
            rc = lvs_gather_rows(xq, (int32_t)p.ld, open_idx + o0, mo, sub, stream);
            if (rc != LVS_OK) return rc;
            rc = lvs_gather_f32(qn, open_idx + o0, mo, subn, stream);
            if (rc != LVS_OK) return rc;
            rc = lvs_flat_search_keys(c_packed, c_pack, k, sub, x_pack, mo, d, LVS_METRIC_L2, 1, c_norms, subn, 0, nullptr, subk,
                                      ws + p.off_search, p.search_bytes, stream);
            if (rc != LVS_OK) return rc;
            hipLaunchKernelGGL(scatter_keys_kernel, dim3((unsigned)lvs_ceil_div(mo, 256)), dim3(256), 0, st, (u64*)keys,
                               (const int64_t*)(open_idx + o0), (const u64*)subk, (long long)mo);
            LVS_HIP_CHECK(hipGetLastError());
        }
    }
    if (out_host_counts) {
        out_host_counts[0] = tot_pair;
        out_host_counts[1] = tot_open;
    }
    // ---- sums in row order, objective, reduction over the ranks, update
    LVS_HIP_CHECK(hipMemsetAsync(sums, 0, ((size_t)k * d + (size_t)k) * 4, st));
    if (n > 0) {
        const int32_t rc = lvs_kmeans_accumulate_keys(x, n, d, x_pack, out_keys, 0, k, sums, cnt, ws + p.off_acc, p.acc_bytes, stream);
        if (rc != LVS_OK) return rc;
    }
    int32_t rc = lvs_kmeans_objective(centroids, sums, cnt, k, d, x_norms_sq_sum, out_obj, ws + p.off_obj, p.obj_bytes, stream);
    if (rc != LVS_OK) return rc;
    if (all_reduce) {
        rc = all_reduce(all_reduce_ctx, sums, (int64_t)k * d + k, /* float32 */ 0, stream);
        if (rc != LVS_OK) return rc;
        rc = all_reduce(all_reduce_ctx, out_obj, 1, /* float64 */ 1, stream);
        if (rc != LVS_OK) return rc;
    }
    return lvs_kmeans_update_centroids(sums, cnt, k, d, n_train_total, centroids, out_nsplit, c_pack, c_packed, c_norms, c_stats, stream);
}

// ---- RCCL as the transport (resolved at run time, like lvs_search_sharded_rccl) ----
namespace {
typedef int (*nccl_all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
nccl_all_reduce_t g_bound_all_reduce = nullptr;
nccl_all_reduce_t rccl_all_reduce_fn() {
    if (g_bound_all_reduce) return g_bound_all_reduce;
    static nccl_all_reduce_t f = [] {
        nccl_all_reduce_t r = (nccl_all_reduce_t)dlsym(RTLD_DEFAULT, "ncclAllReduce");
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            if (r) break;
            void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) r = (nccl_all_reduce_t)dlsym(h, "ncclAllReduce");
        }
        return r;
    }();
    return f;
}
int32_t rccl_all_reduce(void* comm, void* buf, int64_t count, int32_t dtype, void* stream) {
    nccl_all_reduce_t f = rccl_all_reduce_fn();
    if (!f) return LVS_EDEVICE;
    const int rc = f(buf, buf, (size_t)count, dtype == 0 ? /* ncclFloat32 */ 7 : /* ncclFloat64 */ 8, /* ncclSum */ 0, comm, (hipStream_t)stream);
    if (rc != 0) {
        lvs_set_error("ncclAllReduce failed with status %d", rc);
        return LVS_EDEVICE;
    }
    return LVS_OK;
}
}  // namespace

extern "C" int32_t lvs_rccl_bind_all_reduce(void* nccl_all_reduce) {
    g_bound_all_reduce = (nccl_all_reduce_t)nccl_all_reduce;
    return LVS_OK;
}

extern "C" int32_t lvs_kmeans_iteration_rccl(void* nccl_comm, const void* x, int32_t x_pack, int64_t n, int32_t d,
                                             const float* x_norms_sq, const double* x_norms_sq_sum, int32_t exp_sum, int32_t k,
                                             int64_t n_train_total, float* centroids, int32_t c_pack, void* c_packed, float* c_norms,
                                             float* c_stats, uint64_t* out_keys, double* out_obj, int32_t* out_nsplit,
                                             int64_t* out_host_counts, void* workspace, int64_t workspace_bytes, void* stream) {
    LVS_REQUIRE(nccl_comm, "NULL communicator");
    if (!rccl_all_reduce_fn()) {
        const char* why = dlerror();
        lvs_set_error("librccl.so.1 could not be loaded (or lacks ncclAllReduce): %s", why ? why : "");
        return LVS_EDEVICE;
    }
    return lvs_kmeans_iteration(rccl_all_reduce, nccl_comm, x, x_pack, n, d, x_norms_sq, x_norms_sq_sum, exp_sum, k, n_train_total,
                                centroids, c_pack, c_packed, c_norms, c_stats, out_keys, out_obj, out_nsplit, out_host_counts,
                                workspace, workspace_bytes, stream);
}

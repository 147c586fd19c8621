/* lotus_hip.h - C ABI of the MI355X-native LOTUS embedding-retrieval hot path (liblotus_hip.so).
 *
 * The reference (lotus-data/lotus, pure Python) has no FFI of its own: its arithmetic is the third-party
 * faiss-cpu wheel reached through SWIG.  Every entry point below names the reference call it replaces
 * (paths relative to the reference checkout).  Host code (Python/ctypes today - see INTEGRATION.md) binds
 * exactly these symbols.
 *
 * Conventions
 *   - every function returns int32 status: LVS_OK (0) or a negative LVS_E*; lvs_last_error() gives the
 *     thread-local message of the last failure.
 *   - all data pointers are DEVICE pointers (HBM) unless the name ends in _host; the caller owns every buffer;
 *     the library allocates nothing, keeps no pointer after a call returns and never synchronises: work is only
 *     enqueued on `stream` (a hipStream_t passed as void*, NULL = default stream).  The stream's device is made
 *     current for the duration of a call, so one process may drive several GPUs.
 *   - the library reads no environment variable.  (A separate tuning build, `make -C lotus_amd/csrc tuning`, compiles
 *     the LVS_* tuning/ablation knobs in; lvs_build_flags() tells the two apart.)
 *   - "rows" matrices are the library's device layout produced by lvs_pack_rows(): fp16, row-major, leading
 *     dimension lvs_packed_ld(d, mode) halfs.
 *   - result keys: one uint64 per (query, rank): ord32(score where larger = better) << 32 | (0xFFFFFFFF - id);
 *     descending key order == (score best-first, id ascending); key 0 == empty slot (fewer than k rows).
 */
#ifndef LOTUS_HIP_H
#define LOTUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LVS_ABI_VERSION 7 /* 2: pack modes in lvs_flat_search_workspace_bytes, lvs_build_flags, k-means device update;
                             3: k-means iteration entirely on the device (objective, split, repack, accumulate from keys),
                                lvs_pack_rows_checked (validation + power-of-two scale), lvs_absmax, lvs_margin_select_stats;
                                scale exponents in lvs_unpack_rows / lvs_keys_to_result / lvs_scores / lvs_range_join;
                             4: pooled sample thresholds of a sharded join (lvs_flat_search_seed_tiles / _seed_scores /
                                lvs_flat_search_keys_seeded); query-streaming nearest-row search with a two-candidate
                                certificate (lvs_nearest3 / _select / lvs_resolve_pairs);
                             5: banded candidate lists for the certified one-pass search (lvs_flat_search_keys_hi_banded,
                                lvs_certify_topk_banded);
                             6: the row-sharded search with its exchange steps inside the library (lvs_search_sharded with a
                                caller-supplied all-gather, lvs_search_sharded_rccl on an ncclComm_t, lvs_rccl_available / _bind) */

#define LVS_OK 0
#define LVS_EINVAL (-1)   /* bad argument */
#define LVS_ENOMEM (-2)   /* workspace too small */
#define LVS_EDEVICE (-3)  /* HIP runtime error */
#define LVS_EUNSUPPORTED (-4)

/* element types of caller-side embeddings */
#define LVS_DTYPE_F32 0
#define LVS_DTYPE_F16 1

/* metrics; values follow faiss.METRIC_INNER_PRODUCT / METRIC_L2 (lotus/vector_store/faiss_vs.py:14) */
#define LVS_METRIC_IP 0
#define LVS_METRIC_L2 1

/* packed-row modes */
#define LVS_PACK_F16 0   /* fp16 values, one MFMA pass (embeddings stored as fp16) */
#define LVS_PACK_SPLIT 1 /* fp32 values carried as fp16 hi|lo pair, three MFMA passes, ~2^-21 relative error */

/* largest k lvs_flat_search_keys accepts: one pass for k <= 56 (k <= 15 on the faster 256-query geometry); beyond that
   two passes (per-slab lists -> threshold key -> collect -> sort), falling back to ceil(k/56) selection passes when a
   candidate bucket overflows */
#define LVS_MAX_K 2048

int32_t lvs_abi_version(void);
const char* lvs_last_error(void);
/* bit set of LVS_BUILD_*: 0 for the shipped library */
#define LVS_BUILD_TUNING 1       /* environment-variable tuning / ablation knobs compiled in (results may be wrong) */
#define LVS_BUILD_COUNT_EVENTS 2 /* slow-path event counters compiled into the tile kernel */
int32_t lvs_build_flags(void);

/* Device discovery (no reference counterpart; the reference is CPU-only). */
int32_t lvs_device_count(int32_t* out_count);
int32_t lvs_device_info(int32_t device, char* name, int32_t name_cap, int32_t* out_cus, int64_t* out_hbm_bytes);

/* ---- packing: replaces faiss `index.add(embeddings)` (faiss_vs.py:24,64) and the python-wrapper cast of
 * `query_vectors` (faiss_vs.py:67,75): C-contiguous float32 -> device fp16 rows. ---- */
int32_t lvs_packed_ld(int32_t d, int32_t pack_mode);        /* leading dimension in halfs, or <0 */
/* src: [n][d] of src_dtype (device). dst: [n][ld] fp16 (device). If normalize != 0 rows are L2-normalised in
 * fp32 before rounding (cosine = inner product of normalised rows; sentence_transformers_rm.py:30,71).
 * out_norms_sq (nullable): [n] float32 |x_i|^2 of the stored (rounded) values, used by the L2 metric. */
int32_t lvs_pack_rows(const void* src, int32_t src_dtype, int64_t n, int32_t d, int32_t pack_mode,
                      int32_t normalize, void* dst, float* out_norms_sq, void* stream);
/* The same with input validation and an exact scale (faiss takes any float32, faiss_vs.py:24; fp16-based rows do not):
 * every value is multiplied by 2^scale_exp before it is rounded (a power of two: exact), so a caller can move its
 * embeddings into the middle of fp16's range whatever their magnitude - the hi|lo pair then keeps ~22 significant bits
 * for row norms of 1e-2 as for 1e+2 (unscaled, the lo half of small values falls into fp16's subnormals).  Norms and
 * scores downstream are those of the scaled values: 2^(e_b + e_q) x the true inner product, 2^(2e) x the true squared
 * distance when both sides share e (required for L2); lvs_keys_to_result / lvs_scores / lvs_range_join / lvs_unpack_rows
 * take the exponent to undo it.  *out_flags (device word, nullable, OR-ed into - zero it first) receives
 * LVS_PACK_FLAG_NONFINITE when an input value is inf / NaN and LVS_PACK_FLAG_RANGE when a finite value lies outside fp16's
 * range after scaling (|x 2^scale_exp| > 65504). */
#define LVS_PACK_FLAG_NONFINITE 1
#define LVS_PACK_FLAG_RANGE 2
int32_t lvs_pack_rows_checked(const void* src, int32_t src_dtype, int64_t n, int32_t d, int32_t pack_mode, int32_t normalize,
                              int32_t scale_exp, void* dst, float* out_norms_sq, uint32_t* out_flags, void* stream);
/* *inout_bits (device word) = max(*inout_bits, bit pattern of the largest |x| of src [n][d]): non-negative floats order like
 * their bit patterns, so chunks accumulate with atomicMax; inf / NaN show up as patterns >= 0x7F800000.  How a caller
 * picks scale_exp. */
int32_t lvs_absmax(const void* src, int32_t src_dtype, int64_t n, int32_t d, uint32_t* inout_bits, void* stream);
/* dst[i] = src[ids[i]] for packed rows (the `ids` branch gather, faiss_vs.py:59-64). */
int32_t lvs_gather_rows(const void* src, int32_t ld, const int64_t* ids, int64_t n_ids, void* dst, void* stream);
/* dst[i][0..d) = float32 value (hi + lo) of packed row ids[i] (row i when ids is NULL): the inverse of lvs_pack_rows
 * up to its rounding (values x 2^-scale_exp: the exponent the rows were packed with) - serves `get_vectors_from_index` (faiss_vs.py:38-41) for device-resident indexes and the initial
 * k-means centroids (lotus/utils.py:62) without a host copy of the matrix. */
int32_t lvs_unpack_rows(const void* src, int32_t d, int32_t pack_mode, const int64_t* ids, int64_t n, int32_t scale_exp,
                        float* dst, void* stream);
/* dst[i] = src[ids[i]] for a float32 vector (row norms of a gathered subset). */
int32_t lvs_gather_f32(const float* src, const int64_t* ids, int64_t n_ids, float* dst, void* stream);

/* ---- exact top-k search: replaces faiss `IndexFlat::search` behind `index.search(query_vectors, K)`
 * (faiss_vs.py:67,75) - tiled MFMA distance + fused per-query top-k. ---- */
/* bytes of scratch lvs_flat_search_keys needs for this problem (same shape and pack modes as the search call) */
int64_t lvs_flat_search_workspace_bytes(int64_t nq, int64_t nb, int32_t d, int32_t k, int32_t xb_pack,
                                        int32_t xq_pack);
/* xb: [nb][ld(xb_pack)] packed corpus shard, xq: [nq][ld(xq_pack)] packed queries; the two sides may use
 * different pack modes (e.g. fp16 points against fp32-accurate centroids).
 * xb_norms_sq / xq_norms_sq: |.|^2 per row, required for LVS_METRIC_L2, ignored for IP.
 * id_offset: global id of shard row 0 (keys carry global ids < 2^32).
 * row_ids (nullable): [nb] uint32 - id to report for each shard row instead of id_offset + row (subset search).
 * out_keys: [nq][k] uint64, best first.
 * k > 56 runs two corpus passes (per-slab lists -> threshold key -> collect -> sort); should a candidate bucket
 * overflow (mass ties), ceil(k/56) selection passes follow, predicated on a device-side flag - no host round trip. */
int32_t lvs_flat_search_keys(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack, int64_t nq,
                             int32_t d, int32_t metric, int32_t k, const float* xb_norms_sq, const float* xq_norms_sq,
                             int64_t id_offset, const uint32_t* row_ids, uint64_t* out_keys, void* workspace,
                             int64_t workspace_bytes, void* stream);
/* ---- a join whose corpus is row-sharded over several GPUs (the split of sem_sim_join.py:132-134): every shard's search
 * starts from thresholds that only know its own rows, and a 125 k-row shard of an 8-GPU join pays ~3 x the list insertions
 * per query of the unsharded 1 M-row stream.  So every shard scores a sample of its rows (the first `tiles` tiles of 256
 * rows, best score per tile and query: lvs_flat_search_seed_scores), the [tiles][nq] blocks are all-gathered (RCCL), and
 * every shard searches with the k-th largest of ALL shards' sample maxima as its starting threshold
 * (lvs_flat_search_keys_seeded).  Exact: every value is the score of a real row of the searched set exactly as the search
 * computes it, and the k-th largest of a subset never exceeds the k-th largest of all rows; a shard may return fewer than k
 * keys (empty slots = key 0) when fewer of its rows reach the threshold - lvs_merge_keys of all shards' lists is complete.
 * ---- */
/* sample tiles a shard of nb rows contributes for nq queries (0: this shape is not seeded; k <= 56 only).  Beyond 64 query
 * tiles: the call's own sample size or a twenty-fourth of the shard, whichever is larger, at most 24 tiles. */
int32_t lvs_flat_search_seed_tiles(int64_t nq, int64_t nb, int32_t k);
/* out_scores [tiles][nq] float32: the best score (larger = better domain, the values lvs_flat_search_keys ranks, operands'
 * pack scales included) of every query over tile t = rows [256 t, 256 t + 256) of xb; tiles past the shard's last whole
 * tile are filled with -inf ("no row seen"); a shard shorter than one tile (nb < 256, NULL buffers allowed) fills its whole
 * block that way. */
int32_t lvs_flat_search_seed_scores(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack, int64_t nq,
                                    int32_t d, int32_t metric, const float* xb_norms_sq, const float* xq_norms_sq,
                                    int32_t tiles, float* out_scores, void* stream);
/* lvs_flat_search_keys with the starting thresholds taken from seed_scores [seed_rows][nq] (scores of rows of the searched
 * set, e.g. the all-gathered lvs_flat_search_seed_scores blocks of every shard) instead of the call's own sample pass.
 * Ignored (plain search) when seed_rows < k or k > 56. */
int32_t lvs_flat_search_keys_seeded(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack, int64_t nq,
                                    int32_t d, int32_t metric, int32_t k, const float* xb_norms_sq, const float* xq_norms_sq,
                                    int64_t id_offset, const uint32_t* row_ids, const float* seed_scores, int32_t seed_rows,
                                    uint64_t* out_keys, void* workspace, int64_t workspace_bytes, void* stream);
/* The same search over the fp16 "hi" parts of both operands only: ONE MFMA pass whatever the pack modes (hi|lo rows are
 * read at their own leading dimension).  Scores inside the keys are approximations: |s_hi - s| <= |q| |lo_row| + |lo_q| |row|
 * <= 2^-10 |q| |row|.  Used with k1 > k list slots + lvs_rescore_keys + lvs_sort_keys_desc + lvs_certify_topk this gives the
 * exact top k of fp32-accurate operands at the cost of fp16 ones (same workspace size as lvs_flat_search_keys). */
int32_t lvs_flat_search_keys_hi(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack, int64_t nq,
                                int32_t d, int32_t metric, int32_t k, const float* xb_norms_sq, const float* xq_norms_sq,
                                int64_t id_offset, const uint32_t* row_ids, uint64_t* out_keys, void* workspace,
                                int64_t workspace_bytes, void* stream);
/* lvs_flat_search_keys_hi with BANDED lists: k list slots per query, but a row is admitted only while its one-pass score is
 * >= max(last slot, kc-th slot - (band_scale * sqrt(xq_norms_sq[q]) + band_slack)) - both rise monotonically during the search.
 * A row further than the certificate's band below the kc-th best can never matter to lvs_certify_topk_banded, so the lists
 * cost the insertions of kc-deep ones and slots beyond the band stay EMPTY (key 0, sorted last) unless the band is crowded.
 * kc <= k; needs xq_norms_sq under either metric.  Launches that do not go through the tiled list kernel (small batches on
 * the streaming kernel) return plain k-deep lists - which lvs_certify_topk_banded accepts just the same. */
int32_t lvs_flat_search_keys_hi_banded(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack, int64_t nq,
                                       int32_t d, int32_t metric, int32_t k, int32_t kc, float band_scale, float band_slack,
                                       const float* xb_norms_sq, const float* xq_norms_sq, int64_t id_offset,
                                       const uint32_t* row_ids, uint64_t* out_keys, void* workspace, int64_t workspace_bytes,
                                       void* stream);
/* keys [nq][k] sorted best-first in place (k <= 64). */
int32_t lvs_sort_keys_desc(uint64_t* keys, int64_t nq, int32_t k, void* stream);
/* approx_keys [nq][k1] from lvs_flat_search_keys_hi (best first), exact_keys [nq][k1] the same candidates after
 * lvs_rescore_keys + lvs_sort_keys_desc.  A query is certified when its k-th exact score is strictly above
 * (last one-pass score of its list) + scale * sqrt(q_norms_sq) + slack: then no row outside the list can reach the top k.
 * Uncertified queries are appended to out_idx (order unspecified), *out_count (device uint64, zeroed by the caller) += n. */
int32_t lvs_certify_topk(const uint64_t* approx_keys, const uint64_t* exact_keys, const float* q_norms_sq, int64_t nq,
                         int32_t k1, int32_t k, float scale, float slack, int64_t* out_idx, uint64_t* out_count, void* stream);
/* The same for lists from lvs_flat_search_keys_hi_banded: with bound = scale * sqrt(q_norms_sq) + slack, a row outside the list
 * scored at most max(last slot, k-th one-pass score - band * bound), `band` no larger than the search's (band_scale / scale); the
 * query is certified when its k-th exact score is strictly above that + bound - always, once band > 2 and the band was not
 * crowded beyond the list's slots.  band = 0: lvs_certify_topk. */
int32_t lvs_certify_topk_banded(const uint64_t* approx_keys, const uint64_t* exact_keys, const float* q_norms_sq, int64_t nq,
                                int32_t k1, int32_t k, float scale, float slack, float band, int64_t* out_idx,
                                uint64_t* out_count, void* stream);
/* Merge `nparts` candidate lists (e.g. the all-gathered per-shard lists): parts [nparts][nq][k] -> out [nq][k].
 * Any nparts >= 1 and k <= LVS_MAX_K (long lists are folded in rounds of at most 4096 keys per query). */
int32_t lvs_merge_keys(const uint64_t* parts, int32_t nparts, int64_t nq, int32_t k, uint64_t* out_keys,
                       void* stream);
/* ---- the row-sharded search with its exchange steps inside the library (one rank = one GPU = one caller; the split of
 * sem_sim_join.py:132-134 over the GPUs of a node).  What it runs, on `stream`, never synchronising:
 *   [seed_tiles > 0, fp16 rows, k <= 56, nranks > 1]  lvs_flat_search_seed_scores -> all-gather of the [tiles][nq] blocks
 *   lvs_flat_search_keys_seeded(id_offset = global id of this shard's row 0)       -> all-gather of the [nq][k] key lists
 *   lvs_merge_keys                                                                 -> out_keys [nq][k], identical on every rank
 * Every rank must pass the same nq, d, k, metric, pack modes and seed_tiles (= lvs_flat_search_seed_tiles(nq, NOMINAL shard
 * rows, k), or 0 for no pooled thresholds); nb_local may differ per rank and may be 0 (an empty shard contributes empty lists).
 * Equal to the single-launch search of the concatenated shards key for key (keys are a total order).
 * Errors and the collective: arguments, shapes and the workspace size are checked BEFORE the first exchange, so a rank
 * with bad arguments returns without having entered a collective its peers would wait in - but the check is per rank: if
 * the ranks disagree (one passes a short workspace) the others still enter the all-gather.  Any non-OK return after that
 * point (a failed launch, an error from the all-gather) leaves the communicator in an undefined collective state: the
 * caller must abort it, as after any failed NCCL collective. ---- */
/* the all-gather the library calls: `send` [bytes_per_rank] of this rank -> `recv` [nranks][bytes_per_rank] in rank order,
 * device pointers, enqueued on `stream` (or finished before returning); 0 on success */
typedef int32_t (*lvs_all_gather_fn)(void* ctx, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
int64_t lvs_search_sharded_workspace_bytes(int32_t nranks, int64_t nq, int64_t nb_local, int32_t d, int32_t k, int32_t xb_pack,
                                           int32_t xq_pack, int32_t seed_tiles);
int32_t lvs_search_sharded(lvs_all_gather_fn all_gather, void* all_gather_ctx, int32_t nranks, const void* xb, int32_t xb_pack,
                           int64_t nb_local, const void* xq, int32_t xq_pack, int64_t nq, int32_t d, int32_t metric, int32_t k,
                           const float* xb_norms_sq, const float* xq_norms_sq, int64_t id_offset, int32_t seed_tiles,
                           uint64_t* out_keys, void* workspace, int64_t workspace_bytes, void* stream);
/* The same with RCCL as the transport: nccl_comm is the caller's ncclComm_t over the ranks that hold the shards (its size is
 * the number of shards, its rank order the shard order); the two exchanges are ncclAllGather calls on `stream`.  librccl.so.1
 * is resolved at run time, so this library loads and every other entry point works on a host without RCCL: first the
 * ncclAllGather the process already exposes (a host linked against librccl), then dlopen of librccl.so.1 / librccl.so.
 * lvs_rccl_available() = 1 when it resolved.  The communicator must come from THAT copy of RCCL; a process that holds
 * several (PyTorch wheels bundle their own) names the one its communicator belongs to with lvs_rccl_bind (the addresses of
 * its ncclAllGather, ncclCommCount and - optionally - ncclGetErrorString; NULL, NULL, NULL unbinds). */
int32_t lvs_rccl_available(void);
int32_t lvs_rccl_bind(void* nccl_all_gather, void* nccl_comm_count, void* nccl_get_error_string);
int32_t lvs_search_sharded_rccl(void* nccl_comm, const void* xb, int32_t xb_pack, int64_t nb_local, const void* xq,
                                int32_t xq_pack, int64_t nq, int32_t d, int32_t metric, int32_t k, const float* xb_norms_sq,
                                const float* xq_norms_sq, int64_t id_offset, int32_t seed_tiles, uint64_t* out_keys,
                                void* workspace, int64_t workspace_bytes, void* stream);
/* keys -> faiss-shaped result (faiss_vs.py:67,75 return values): D float32 [nq][k], I int64 [nq][k];
 * empty slots become I = -1, D = -FLT_MAX (IP) / +FLT_MAX (L2).  id_map (nullable): I = id_map[id]
 * (the sub-index -> global id remap of faiss_vs.py:71-72).  score_exp: D is multiplied by 2^-score_exp (the sum of the two
 * operands' pack exponents; 0 for unscaled rows). */
int32_t lvs_keys_to_result(const uint64_t* keys, int64_t nq, int32_t k, int32_t metric, const int64_t* id_map,
                           int32_t score_exp, float* out_D, int64_t* out_I, void* stream);

/* ---- full score rows (callers that ask for K = N: sem_dedup.py:45, sem_filter.py:491-497, sem_join.py:367):
 * out [nq][ld_out] float32 "better" scores (IP: the product; L2: minus the squared distance). ---- */
int32_t lvs_scores(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack, int64_t nq, int32_t d,
                   int32_t metric, const float* xb_norms_sq, const float* xq_norms_sq, int32_t score_exp, float* out,
                   int64_t ld_out, void* stream);
/* Rank every score row best-first: scores [nq][ld] (from lvs_scores) -> out_keys [nq][nb], ids = id_offset + column.
 * nq * nb must stay below 2^32.  Serves K = N callers beyond LVS_MAX_K. */
int64_t lvs_sort_rows_workspace_bytes(int64_t nq, int64_t nb);
int32_t lvs_sort_rows_desc(const float* scores, int64_t nq, int64_t nb, int64_t ld, int64_t id_offset,
                           uint64_t* out_keys, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- threshold join: replaces `sem_sim_join(self, K = len(df))` + `_scores > threshold` (sem_dedup.py:45-46), which
 * materialises N^2 results in the reference.  Emits (query, corpus row id, score) for every score STRICTLY greater
 * than `threshold` (for L2 the score is minus the squared distance).  q_row0 >= 0 selects the self-join: query r is
 * corpus row q_row0 + r and only pairs with id > q_row0 + r are kept (each unordered pair once), tiles below the
 * diagonal are skipped.  qt_stride / qt_phase deal 256-query tiles round-robin to ranks (multi-GPU, corpus replicated).
 * out_count (device uint64, zeroed by the caller) receives the number of qualifying pairs; pairs beyond `capacity`
 * are counted but not stored, so a caller can size the buffers and run again.  Pair order is unspecified.  threshold and
 * out_s are in the caller's units: score_exp (sum of the operands' pack exponents) is applied on the way in and out. ---- */
int32_t lvs_range_join(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack, int64_t nq,
                       int32_t d, int32_t metric, const float* xb_norms_sq, const float* xq_norms_sq, float threshold,
                       int32_t score_exp, int64_t q_row0, int64_t id_offset, int32_t qt_stride, int32_t qt_phase,
                       int64_t capacity, int64_t* out_q, int64_t* out_j, float* out_s, uint64_t* out_count, void* stream);

/* ---- k-means pieces: replace faiss `Kmeans(d, k, niter).train(x)` (lotus/utils.py:61-62); the assignment step and
 * the final `kmeans.index.search(x, 1)` (utils.py:65) are lvs_flat_search_keys with k = 1 and LVS_METRIC_L2 - or, in ONE
 * MFMA pass whatever the pack modes, the certified pair below. ---- */
/* Nearest corpus row of every query from the fp16 "hi" parts of both operands only (hi|lo rows are read at their own
 * leading dimension, the lo halves are skipped), with the exact norms of the stored values: out_keys [nq] = winner key,
 * out_second [nq] = runner-up score in the "larger = better" domain (-inf when there is none).  The true score of a
 * (query, row) pair differs from this one by at most 2 |q| |lo_row| + 2 |lo_q| |row| (Cauchy-Schwarz), so a winner whose
 * margin exceeds twice that bound is the exact winner; lvs_margin_select lists the queries that are NOT certified so
 * that only those go through the exact (2-3 pass) search again.  The kernel carries a score's position in the low six
 * mantissa bits of u = q.y (inner product) or u = 2 q.y - |y|^2 (L2) while it runs, so the reported winner and runner-up
 * scores are each perturbed by less than 2^-17 |u|: add 2^-16 max|u| to the margin bound. */
int64_t lvs_nearest_hi_workspace_bytes(int64_t nq, int64_t nb, int32_t d);
/* The same one-pass search for a SMALL corpus (at most LVS_NEAREST3_MAX_ROWS rows: the centroids of a k-means,
 * lotus/utils.py:62,65) with the loop turned around - a workgroup keeps one 256-row corpus tile and streams the query
 * tiles past it, the workgroups holding the other corpus tiles walk the same queries at the same time and share them
 * through their XCD's L2 - and with one more output: out_keys [nq] winner key, out_keys2 [nq] runner-up key (0: none),
 * out_second [nq] runner-up score, out_third [nq] third-best score (-inf: none), all in the "larger = better" domain and
 * perturbed like lvs_nearest_hi's.  lvs_nearest3_select splits the queries three ways with the bound of
 * lvs_margin_select_stats: certified (best - second > bound), PAIR (best - third > bound: only the best two rows can win -
 * appended to out_pair_idx) and open (appended to out_open_idx; exact search over every row); out_counts [2] (device uint64,
 * zeroed by the caller) += pairs / open.  lvs_resolve_pairs settles the pairs with two exact (hi + lo, float32) scores per
 * query - the arithmetic of lvs_rescore_keys - and writes the better key (score, then lower id) into keys[q]; it reads
 * the pair count from the device (at most `capacity` entries), so nothing waits for the host. */
#define LVS_NEAREST3_MAX_ROWS 16384
int64_t lvs_nearest3_workspace_bytes(int64_t nq, int64_t nb, int32_t d);
int32_t lvs_nearest3(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack, int64_t nq, int32_t d,
                     int32_t metric, const float* xb_norms_sq, const float* xq_norms_sq, int64_t id_offset,
                     uint64_t* out_keys, uint64_t* out_keys2, float* out_second, float* out_third, void* workspace,
                     int64_t workspace_bytes, void* stream);
int32_t lvs_nearest3_select(const uint64_t* keys, const uint64_t* keys2, const float* second, const float* third,
                            const float* q_norms_sq, int64_t nq, const float* corpus_stats, const float* coef5,
                            int64_t* out_pair_idx, int64_t* out_open_idx, uint64_t* out_counts, void* stream);
int32_t lvs_resolve_pairs(const void* xb, int32_t xb_pack, const void* xq, int32_t xq_pack, int32_t d, int32_t metric,
                          const float* xb_norms_sq, const float* xq_norms_sq, int64_t id_offset, const int64_t* pair_idx,
                          const uint64_t* pair_count, int64_t capacity, uint64_t* keys, const uint64_t* keys2, void* stream);
int32_t lvs_nearest_hi(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack, int64_t nq, int32_t d,
                       int32_t metric, const float* xb_norms_sq, const float* xq_norms_sq, int64_t id_offset,
                       uint64_t* out_keys, float* out_second, void* workspace, int64_t workspace_bytes, void* stream);
/* keys [nq][k] in/out: the score inside every key is replaced by the exact fp32 score (hi + lo parts of both operands) of
 * the pair (query q, the row the key names; row = id - id_offset); empty slots (key 0) stay empty; the order inside a
 * query's list is NOT restored (lvs_sort_keys_desc). */
int32_t lvs_rescore_keys(const void* xb, int32_t xb_pack, const void* xq, int32_t xq_pack, int64_t nq, int32_t d,
                         int32_t metric, const float* xb_norms_sq, const float* xq_norms_sq, int64_t id_offset, int32_t k,
                         uint64_t* keys, void* stream);
/* out_idx [<= nq] (order unspecified) = queries with score(key) - second <= scale * sqrt(q_norms_sq) + slack;
 * *out_count (device uint64, zeroed by the caller) += their number.  q_norms_sq NULL means |q| = 1. */
int32_t lvs_margin_select(const uint64_t* keys, const float* second, const float* q_norms_sq, int64_t nq, float scale,
                          float slack, int64_t* out_idx, uint64_t* out_count, void* stream);
/* The same test with the corpus statistics read from the device: corpus_stats[0] = R^2 (largest squared row norm),
 * corpus_stats[1] = E^2 (largest squared lo-part norm) as written by lvs_kmeans_pack_centroids / lvs_kmeans_update_centroids;
 * bound = (coef5[0] E + coef5[1] R) sqrt(q_norms_sq) + coef5[2] + coef5[3] R + coef5[4] R^2.  Lets the k-means loop run
 * without a host round trip per iteration. */
int32_t lvs_margin_select_stats(const uint64_t* keys, const float* second, const float* q_norms_sq, int64_t nq,
                                const float* corpus_stats, const float* coef5, int64_t* out_idx, uint64_t* out_count,
                                void* stream);
int64_t lvs_kmeans_accumulate_workspace_bytes(int64_t n, int32_t k);
/* sums [k][d] float32: the packed rows of x, grouped by assign[i] (int64, values outside [0,k) are skipped), are added to
 * what is there, one after the other in row order per centroid (faiss compute_centroids order) - the chain of float32
 * additions CONTINUES from the caller's values, so handing the rows over in consecutive ranges (one call each, sums carried
 * along, zeros before the first) gives bit for bit the sums of one call over all rows: a caller may run the sums of one range
 * on a side stream under the assignment search of the next (lotus_amd/cluster.py).  counts [k] float32 += group sizes.  Both
 * must be initialised by the caller.  (Rows are bucketed by a stable counting sort on the centroid ids: per-chunk histograms,
 * a scan, an in-order scatter; one wave per (centroid, 128 columns) walks its bucket.) */
int32_t lvs_kmeans_accumulate(const void* x, int64_t n, int32_t d, int32_t pack_mode, const int64_t* assign,
                              int32_t k, float* sums, float* counts, void* workspace, int64_t workspace_bytes,
                              void* stream);
/* The same straight from the assignment search's result keys [n] (row i is assigned to centroid id(keys[i]) - id_offset;
 * empty keys are skipped): no decode pass in between. */
int32_t lvs_kmeans_accumulate_keys(const void* x, int64_t n, int32_t d, int32_t pack_mode, const uint64_t* keys,
                                   int64_t id_offset, int32_t k, float* sums, float* counts, void* workspace,
                                   int64_t workspace_bytes, void* stream);
/* faiss's objective of an iteration (sum of the squared assignment distances, Clustering.cpp) without a pass over the
 * points, from what the update needs anyway: *out_obj (device float64) = *x_norms_sq_sum - 2 sum_j <c_j, S_j> + sum_j n_j |c_j|^2
 * with the centroids BEFORE the update (float64 arithmetic, fixed summation order).  x_norms_sq_sum (device float64,
 * nullable = 0): sum of |x_i|^2 over the assigned rows.  Linear in (sums, counts): ranks add their partial values. */
int64_t lvs_kmeans_objective_workspace_bytes(int32_t k);
int32_t lvs_kmeans_objective(const float* centroids, const float* sums, const float* counts, int32_t k, int32_t d,
                             const double* x_norms_sq_sum, double* out_obj, void* workspace, int64_t workspace_bytes,
                             void* stream);
/* packed_out / norms_out = lvs_pack_rows(centroids) and stats_out (device float[2], nullable) = {largest |row|^2, largest
 * |lo part of a row|^2} - the centroid-side terms of the one-pass assignment's certificate (lvs_margin_select_stats). */
int32_t lvs_kmeans_pack_centroids(const float* centroids, int32_t k, int32_t d, int32_t pack_mode, void* packed_out,
                                  float* norms_out, float* stats_out, void* stream);
/* One call = the rest of a faiss Clustering iteration after the sums: centroids [k][d] float32 (in/out): c = sums * (1 / count)
 * where count > 0, unchanged where the cluster is empty (compute_centroids); then, when n_train > 0, faiss's split_clusters
 * on the device (std::mt19937(1234) replayed by one thread: same draws and decisions as lvs_kmeans_split_clusters_host;
 * counts is modified as faiss modifies hassign; *out_nsplit (device int32, nullable) = clusters re-seeded); then, when
 * packed_out is given, the updated centroids are repacked (lvs_kmeans_pack_centroids). */
int32_t lvs_kmeans_update_centroids(const float* sums, float* counts, int32_t k, int32_t d, int64_t n_train, float* centroids,
                                    int32_t* out_nsplit, int32_t pack_mode, void* packed_out, float* norms_out,
                                    float* stats_out, void* stream);
/* ---- (ABI 7) ONE Lloyd iteration of faiss `Clustering::train` as one call (lotus/utils.py:61-62) - SURVEY.md 8(b)'s
 * `lvs_kmeans`, cut at the iteration so that the caller owns the loop, the subsample and the initial centroids
 * (lvs_rand_perm_host).  x: this rank's n training rows (packed, |x|^2 beside them; x_norms_sq_sum = device float64, their sum);
 * centroids [k][d] float32 (in: the iteration's centroids; out: the next ones), c_packed / c_norms / c_stats = their hi|lo
 * image as lvs_kmeans_pack_centroids / this call leave it (c_pack must be LVS_PACK_SPLIT: fp32-accurate centroids; k <=
 * LVS_NEAREST3_MAX_ROWS); exp_sum = pack exponent of x + pack exponent of the centroids (0 for unscaled rows).
 * Out: out_keys [n] = the iteration's assignment (row i -> centroid id(key)), *out_obj (device float64) = faiss's objective
 * of the iteration (in the rows' stored domain), *out_nsplit (device int32, nullable) = clusters split_clusters re-seeded,
 * out_host_counts (HOST int64[2], nullable) = rows settled by two exact dot products / by the exact search.
 * What it runs: lvs_nearest3 -> lvs_nearest3_select -> lvs_resolve_pairs -> exact lvs_flat_search_keys of the open rows ->
 * lvs_kmeans_accumulate_keys -> lvs_kmeans_objective -> [all_reduce of sums | counts (float32) and the objective (float64)]
 * -> lvs_kmeans_update_centroids with n_train_total (all ranks' training rows; the split replays identically on every rank).
 * Bit-identical to issuing those calls one by one (lotus_amd/cluster.py).  THE ONE ENTRY POINT THAT SYNCHRONISES `stream`:
 * once per 2 GiB of search scratch (once per call up to ~100 M rows x 1 024 centroids), to size the exact search of the open rows.
 * all_reduce (nullable = single rank): in-place sum of `count` values of dtype 0 = float32 / 1 = float64 across the ranks,
 * enqueued on `stream`; 0 on success. */
typedef int32_t (*lvs_all_reduce_fn)(void* ctx, void* buf, int64_t count, int32_t dtype, void* stream);
int64_t lvs_kmeans_iteration_workspace_bytes(int64_t n, int32_t d, int32_t k, int32_t x_pack, int32_t c_pack);
int32_t lvs_kmeans_iteration(lvs_all_reduce_fn all_reduce, void* all_reduce_ctx, const void* x, int32_t x_pack, int64_t n,
                             int32_t d, const float* x_norms_sq, const double* x_norms_sq_sum, int32_t exp_sum, int32_t k,
                             int64_t n_train_total, float* centroids, int32_t c_pack, void* c_packed, float* c_norms,
                             float* c_stats, uint64_t* out_keys, double* out_obj, int32_t* out_nsplit,
                             int64_t* out_host_counts, void* workspace, int64_t workspace_bytes, void* stream);
/* The same with RCCL as the transport: two ncclAllReduce calls (sum) on the caller's communicator and stream (SURVEY.md 8(e)
 * row 3).  librccl is resolved at run time as for lvs_search_sharded_rccl; a process holding several copies names the one
 * its communicator belongs to with lvs_rccl_bind_all_reduce(address of its ncclAllReduce; NULL unbinds). */
int32_t lvs_rccl_bind_all_reduce(void* nccl_all_reduce);
int32_t lvs_kmeans_iteration_rccl(void* nccl_comm, const void* x, int32_t x_pack, int64_t n, int32_t d, const float* x_norms_sq,
                                  const double* x_norms_sq_sum, int32_t exp_sum, int32_t k, int64_t n_train_total,
                                  float* centroids, int32_t c_pack, void* c_packed, float* c_norms, float* c_stats,
                                  uint64_t* out_keys, double* out_obj, int32_t* out_nsplit, int64_t* out_host_counts,
                                  void* workspace, int64_t workspace_bytes, void* stream);
/* ---- exact distance bounds across k-means iterations (Hamerly): rows whose nearest centroid provably has not changed are
 * not searched again - same assignments as faiss's exhaustive iteration (lotus/utils.py:62), less work once the centroids
 * settle.  Per training row: assign (int32, -1 = unknown), ub >= its distance to the assigned centroid, lb <= its distance
 * to every other centroid (Euclidean, in the rows' stored domain). ---- */
/* out_delta [k] = |c_new_j - c_old_j| (rounded up), out_top2 [3] = {largest delta, its centroid (int bits), second largest} */
int32_t lvs_kmeans_centroid_shift(const float* c_old, const float* c_new, int32_t k, int32_t d, float* out_delta,
                                  float* out_top2, void* stream);
/* Bounds of rows positions[i] (NULL: rows 0..m) from an L2 search result: keys [m][key_stride] (winner first; with
 * second == NULL the runner-up is keys[i][1], an exact k = 2 search), second [m] the runner-up score of lvs_nearest_hi.
 * The search's own error bound (coef5 / corpus_stats as in lvs_margin_select_stats) widens both bounds. */
int32_t lvs_kmeans_bounds_set(const uint64_t* keys, int32_t key_stride, const float* second, const float* q_norms_sq,
                              const int64_t* positions, int64_t m, const float* corpus_stats, const float* coef5,
                              int64_t id_offset, int32_t* assign, float* ub, float* lb, void* stream);
/* Rows whose one-pass winner was not certified and that the exact k = 1 search decided: approx_keys [m] = their one-pass keys
 * (whose bounds lvs_kmeans_bounds_set wrote), exact_keys [m] = the exact search's.  ub is renewed from the exact distance
 * (+ exact_coef2[0] R |x| + exact_coef2[1] R^2 of float32 rounding); lb is kept when the winner is the same centroid and
 * otherwise lowered below the distance to the one-pass winner. */
int32_t lvs_kmeans_bounds_fix(const uint64_t* approx_keys, const uint64_t* exact_keys, const float* q_norms_sq,
                              const int64_t* positions, int64_t m, const float* corpus_stats, const float* coef5,
                              const float* exact_coef2, int64_t id_offset, int32_t* assign, float* ub, float* lb, void* stream);
/* After a centroid update: ub += delta[assign], lb -= largest delta among the other centroids; rows with ub (1 + 1e-5) >= lb
 * (or no assignment yet) are appended to out_idx (order unspecified), *out_count (device uint64, zeroed by the caller) += n. */
int32_t lvs_kmeans_bounds_step(const int32_t* assign, float* ub, float* lb, const float* delta, const float* top2, int64_t n,
                               int64_t* out_idx, uint64_t* out_count, void* stream);
/* HOST helpers (plain host pointers), bit-exact with faiss: rand_perm(n, seed) = Fisher-Yates on std::mt19937
 * (training subsample and initial centroids), and split_clusters (empty-cluster re-seeding, RNG seed 1234). */
int32_t lvs_rand_perm_host(int64_t n, int64_t seed, int64_t* out_perm);
/* out_prefix[0..m) = the first m entries of lvs_rand_perm_host(n, seed) in O(m) time and memory (m <= n). */
int32_t lvs_rand_perm_prefix_host(int64_t n, int64_t seed, int64_t m, int64_t* out_prefix);
int32_t lvs_kmeans_split_clusters_host(int32_t d, int32_t k, int64_t n, float* hassign, float* centroids,
                                       int32_t* out_nsplit);

/* ---- measurement hook: average duration in ms of the dominant search kernel's launches since the last reset,
 * measured with HIP events on the launch stream (enabled with lvs_timing_enable(1)). ---- */
int32_t lvs_timing_enable(int32_t on);
int32_t lvs_timing_read(double* out_total_ms, int64_t* out_launches);
/* The same plus: out_calls = searches (entry-point calls) that timed at least one launch - a search of more than 4 096 queries
 * that runs through the register-resident-queries kernels times one launch per chunk of its queries, so total / calls is the
 * dominant kernel's time per search -, out_kernel = LVS_KERNEL_* of the last timed launch (which kernel family served it). */
#define LVS_KERNEL_TILE 0   /* lvs_tile_kernel: 256 x 256 score tiles, candidate lists under locks */
#define LVS_KERNEL_STREAM 1 /* lvs_stream_kernel: up to 96 queries resident in LDS */
#define LVS_KERNEL_RQ 2     /* lvs_rq_kernel: 32 queries per wave resident in registers */
#define LVS_KERNEL_RJ 3     /* lvs_rj_kernel: 64 queries per wave resident in registers, one wave per SIMD */
int32_t lvs_timing_read_calls(double* out_total_ms, int64_t* out_launches, int64_t* out_calls, int32_t* out_kernel);

#ifdef __cplusplus
}
#endif
#endif /* LOTUS_HIP_H */

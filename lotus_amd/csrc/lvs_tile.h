// Launch interface of the tiled distance / top-k kernel (lvs_tile.hip) and the small-batch kernel (lvs_stream.hip).
#pragma once
#include "lvs_common.h"

#define LVS_BC 256          // corpus rows per score tile (MFMA M)
#define LVS_BK 64           // halfs per K-step

#define LVS2_KCAP 15        // 256 x 256 geometry: list slots per query = largest k per pass
#define LVS2_BQ 256         //                     queries per score tile (MFMA N)
#define LVS3_KCAP 56        // 256 x 128 geometry (k > LVS2_KCAP): list slots per query = largest k per pass
#define LVS3_BQ 128
#define LVS_KPASS LVS3_KCAP // largest k any single pass selects; k up to LVS_MAX_K runs ceil(k / LVS_KPASS) passes

#define LVS_MODE_TOPK 0
#define LVS_MODE_SCORES 1
#define LVS_MODE_TOP1 2
#define LVS_MODE_RANGE 3
#define LVS_MODE_COLLECT 4
#define LVS_MODE_TOP2 5     // TOP1 + the runner-up score per query (certified nearest-row search, lvs_nearest_hi)
#define LVS_MODE_SEED 6     // best TOPK-domain score of every (slab, query): seeds the thresholds of a launch with few query tiles

struct LvsTileArgs {
    const void* xb;           // [nb][ld] fp16 packed corpus shard
    const void* xq;           // [nq][ld] fp16 packed queries
    const float* bn;          // [nb] |y|^2 (L2 only)
    const float* qn;          // [nq] |q|^2 (L2 only)
    const uint32_t* row_ids;  // nullable [nb]: id reported for a shard row
    const u64* ub;            // nullable: per-query exclusive upper bound key (multi-pass k > LVS_KPASS)
    long long ub_stride;      // stride of ub in u64 elements
    uint32_t* gtau;           // [nq] shared running thresholds (ord32 of the k-th best score), zero-initialised
    u64* out;                 // [nslab][nq][k] per-slab candidate keys
    float* out_second;        // LVS_MODE_TOP2: [nslab][nq] runner-up score ("better" domain) of every slab
    float* seed_out;          // LVS_MODE_SEED: [nslab][nq] best score of every slab, bit-identical to the score LVS_MODE_TOPK ranks
    float* scores;            // LVS_MODE_SCORES: [nq][ld_scores]
    // LVS_MODE_RANGE: emit (query, corpus row, score) for score > threshold
    long long* pair_q;
    long long* pair_j;
    float* pair_s;
    unsigned long long* pair_count;
    long long pair_capacity;
    long long q_row0;         // >= 0: self-join, query r is corpus row q_row0 + r and only pairs j > i are kept
    long long q_base;         // RANGE: added to the query numbers written to pair_q (a launch over a slice of the call's queries)
    float threshold;
    float out_scale;          // SCORES / RANGE: factor applied to a score on its way out (2^-e of operands packed with a scale)
    int qt_stride, qt_phase;  // only query tiles with qt % qt_stride == qt_phase are processed (multi-GPU deal)
    long long ld_scores;
    long long nb, nq;
    long long ldb, ldq;       // leading dimensions in halfs (corpus / queries)
    long long id_offset;
    int nkd;                  // padded d / 64
    int nk;                   // K-steps per tile = nseg * nkd
    int nseg;                 // K segments: product = sum over segments of q[seg_q..] . y[seg_c..]
    int seg_q[3], seg_c[3];   // column offsets (halfs) of each segment in the query / corpus rows
    int metric;
    int k;                    // <= the geometry's list capacity
    int bq;                   // queries per tile the plan was made for: LVS2_BQ (256 x 256 geometry) or LVS3_BQ
    int ntiles, tiles_per_slab, nslab, nqt;
    int debug_hot;            // -DLVS_TUNING builds only (env LVS_DEBUG_HOT), timing ablations - results are wrong
                              // unless 0: 2 skip the top-k slow path, 3 scan hits but skip insertions, 4 no wait for
                              // the staging loads.  The shipped kernel does not read this field.
    const uint32_t* pred;     // nullable: the launch is a no-op unless *pred != 0 (device-side predicate)
    int gq;                   // query tiles per XCD group (1,2,4,8,16,32); slabs per group = 32 / gq
    int lead_slabs;           // slabs walked first in 32-wide groups (see item_of_block); 0 or 1
    int no_share;             // 1: slabs do not exchange thresholds - every slab's list is its own exact top-k
    // LVS_MODE_TOPK, banded lists (kc > 0, kc < k; needs qn): a row is admitted only while its score is >= max(last slot,
    // kc-th slot - (bscale * |q| + bslack)); slots beyond the band stay empty (lvs_flat_search_keys_hi_banded)
    int kc;
    float bscale, bslack;
    // LVS_MODE_COLLECT: keys >= thr_key[q] go to bucket[q][*] (unordered, at most bucket_capacity are kept; bucket_count
    // keeps counting so that the caller sees an overflow)
    const u64* thr_key;       // [nq]
    uint32_t* bucket_count;   // [nq] zero-initialised
    u64* bucket;              // [nq][bucket_capacity]
    int bucket_capacity;
    unsigned long long* dbg;  // tuning aid (build with -DLVS_COUNT_EVENTS, run with LVS_COUNT=1): [0] block visits, [1] insertions, [2] wave-tiles, [3..5] cycles in filter / visit loop / insertions
};

// ---- the (query tile, slab) items of a launch in XCD groups of 32 slots (see item_of_block in lvs_tile.hip) ----
// Three regions, in launch order:
//   lead       the first lead_slabs slab(s) of every query tile in 32-wide groups (32 query tiles x 1 slab);
//   full       the nqt / gq whole query groups: gq query tiles x 32 / gq slabs per group, walking the query groups first,
//              then the slabs;
//   remainder  the last nqt % gq query tiles.  Dealt like the others they would fill (nqt % gq) / gq of every group's slots
//              (49 tiles with gq = 8: one tile x 4 slabs in a 32-slot group, 28 CUs of an XCD idle for every 7th group), so
//              they get groups of their own shape: gqr = the power of two >= the remainder query tiles x 32 / gqr slabs.
struct LvsTileGroups {
    int nqg32, n0;        // lead region: nqg32 groups per slab, n0 in all
    int nqf, nsg, n1;     // full region: nqf query groups x nsg slab groups = n1 groups
    int rem, gqr, nsgr;   // remainder region: rem query tiles, gqr x (32 / gqr) slots per group, nsgr groups
    int total, full;      // all groups; the largest multiple of 8 below (whole generations: one group per XCD)
};
__host__ __device__ inline LvsTileGroups lvs_tile_groups(int nqt, int nslab, int gq, int lead_slabs) {
    LvsTileGroups g;
    const int gs = 32 / gq;
    const int rest = nslab - lead_slabs;
    g.nqg32 = (nqt + 31) / 32;
    g.n0 = g.nqg32 * lead_slabs;
    g.nqf = nqt / gq;
    g.nsg = (rest + gs - 1) / gs;
    g.n1 = g.nqf * g.nsg;
    g.rem = nqt - g.nqf * gq;
    g.gqr = 1;
    while (g.gqr < g.rem) g.gqr <<= 1;
    g.nsgr = g.rem ? (rest + 32 / g.gqr - 1) / (32 / g.gqr) : 0;
    g.total = g.n0 + g.n1 + g.nsgr;
    g.full = g.total & ~7;
    return g;
}
// slot r (0..31) of group g -> (query tile, slab); false when the slot is empty
__host__ __device__ inline bool lvs_tile_group_slot(int nqt, int nslab, int gq, int lead_slabs, const LvsTileGroups& gr,
                                                    int g, int r, int& qt, int& slab) {
    if (g < gr.n0) {
        slab = g / gr.nqg32;
        qt = (g % gr.nqg32) * 32 + r;
    } else if (g < gr.n0 + gr.n1) {
        const int h = g - gr.n0;
        qt = (h % gr.nqf) * gq + (r % gq);
        slab = lead_slabs + (h / gr.nqf) * (32 / gq) + (r / gq);
    } else {
        const int h = g - gr.n0 - gr.n1;
        if ((r % gr.gqr) >= gr.rem) return false;
        qt = gr.nqf * gq + (r % gr.gqr);
        slab = lead_slabs + h * (32 / gr.gqr) + (r / gr.gqr);
    }
    return qt < nqt && slab < nslab;
}
// valid slots of group g
__host__ __device__ inline int lvs_tile_group_items(int nqt, int nslab, int gq, int lead_slabs, const LvsTileGroups& gr, int g) {
    int vq, vs;
    if (g < gr.n0) {
        vq = nqt - (g % gr.nqg32) * 32;
        vq = vq > 32 ? 32 : vq;
        vs = 1;
    } else if (g < gr.n0 + gr.n1) {
        vq = gq;
        vs = nslab - lead_slabs - ((g - gr.n0) / gr.nqf) * (32 / gq);
        vs = vs > 32 / gq ? 32 / gq : vs;
    } else {
        vq = gr.rem;
        vs = nslab - lead_slabs - (g - gr.n0 - gr.n1) * (32 / gr.gqr);
        vs = vs > 32 / gr.gqr ? 32 / gr.gqr : vs;
    }
    return (vq < 0 ? 0 : vq) * (vs < 0 ? 0 : vs);
}
// block b of the launch -> (group, slot).  Whole generations: XCD b % 8 runs slot (b / 8) % 32 of group (b / 256) * 8 + b % 8;
// the last total % 8 groups are dealt slot by slot across all XCDs (slot r on XCD r % 8).  false: b is past the last group.
__host__ __device__ inline bool lvs_tile_block_slot(const LvsTileGroups& gr, int b, int& g, int& r) {
    if (b < gr.full * 32) {
        const int x = b & 7, j = b >> 3;
        g = (j >> 5) * 8 + x;
        r = j & 31;
        return true;
    }
    const int f = b - gr.full * 32;
    g = gr.full + (f >> 5);
    r = f & 31;
    return g < gr.total;
}
inline int lvs_tile_grid_blocks(int nqt, int nslab, int gq, int lead_slabs) {
    return lvs_tile_groups(nqt, nslab, gq, lead_slabs).total * 32;
}

// Items (valid (query tile, slab) pairs) each XCD receives under item_of_block's deal, in units of rounds: an XCD runs 32
// items at a time (one workgroup per CU), so it needs ceil(items / 32) item-times.  Returns the largest over the 8 XCDs.
inline int lvs_tile_xcd_rounds(int nqt, int nslab, int gq, int lead_slabs) {
    const LvsTileGroups gr = lvs_tile_groups(nqt, nslab, gq, lead_slabs);
    long long items[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int g = 0; g < gr.full; ++g) items[g & 7] += lvs_tile_group_items(nqt, nslab, gq, lead_slabs, gr, g);
    for (int g = gr.full; g < gr.total; ++g)
        for (int r = 0; r < 32; ++r) {
            int qt, slab;
            if (lvs_tile_group_slot(nqt, nslab, gq, lead_slabs, gr, g, r, qt, slab)) ++items[r & 7];
        }
    long long worst = 0;
    for (int x = 0; x < 8; ++x) worst = items[x] > worst ? items[x] : worst;
    return (int)((worst + 31) / 32);
}
hipError_t lvs_tile_launch(int mode, const LvsTileArgs& a, hipStream_t stream);

// HIP events around a dominant-kernel launch while lvs_timing_enable(1) is on (lvs_capi.hip); a no-op otherwise
struct LvsKernelTimer {
    explicit LvsKernelTimer(hipStream_t st);
    ~LvsKernelTimer();
    void* impl;
};

// ---- small-batch streaming kernel (lvs_stream.hip): nq <= 256, k <= LVS_KPASS.  One or two blocks of 32 queries per
// workgroup; beyond 64 queries 2 or 4 sibling workgroups share a corpus range through their XCD's L2. ----
#define LVS_STREAM_MAXQ 256
#define LVS_STREAM_MAXWG 768   // most corpus ranges (= partial candidate lists per query) a launch may use
#define LVS_STREAM_SEED_ROWS 32768  // sample rows (the first of the shard) whose scores seed the thresholds of a multi-query call
struct LvsStreamArgs {
    const void* xb;
    const void* xq;
    const float* bn;
    const float* qn;
    const uint32_t* row_ids;
    uint32_t* gtau;  // [nq] zero-initialised, or seeded with a valid lower bound of every query's k-th best score
    u64* out;        // [nparts][nq][k]
    float* seed_out; // non-NULL: SEED mode - [nparts][nq] best score of every (corpus range, query) instead of lists
    long long nb, ldb, ldq, id_offset;
    int nq, k, metric;
    int nj;                  // MFMAs per 32-row block = nseg * dpad / 16
    int jper;                // dpad / 16
    int nseg;
    int seg_q[3], seg_c[3];  // column offsets (halfs) of each K segment
    int seg_b[3];            // first B-fragment index of each K segment (segments sharing query columns share fragments)
    int nbfrag;              // B fragments held in LDS per query block
    int blocks_per_wg;       // 32-row blocks per corpus range (contiguous)
    int kcap;                // list slots per query in LDS (k <= kcap <= 64), from lvs_stream_plan
    int nqb;                 // 32-query blocks per workgroup (1 .. 3), from lvs_stream_plan
    int groups;              // sibling workgroups per corpus range (1 .. 4), from lvs_stream_plan
    int nparts;              // out: corpus ranges of the launch = candidate lists per query
    int debug;               // -DLVS_TUNING builds only (env LVS_STREAM_DEBUG): 1 no MFMA / B reads, 2 no block epilogue
};

// ---- lvs_rq.hip: 97 .. 256 queries with the queries resident in registers ----
#define LVS_RQ_GROUPQ 256   // queries of one workgroup (eight waves x 32)
#define LVS_RQ_MAXQ 4096    // most queries of a call lvs_rq_fits takes as ONE launch of up to 16 groups (also the seed pass' chunk)
#define LVS_RQ_CHUNK_MAX 65536  // most queries per launch of a chunked call: 256 groups x 1 range
#define LVS_RQ_CHUNK_DEFAULT 32768  // 128 groups x 2 ranges: fastest of 4 096 .. 65 536 at 100 k x 1 M (profiles/r10j_join_variants.log)
#define LVS_RQ_KMAX 16      // most list slots per query
#define LVS_RQ_JOIN_DEFAULT 1  // calls beyond LVS_RQ_MAXQ queries in chunks through lvs_rq_kernel / lvs_rj_kernel (1) or through the list kernel (0)
#define LVS_RQ_SEED_ROWS 65536  // sample rows (the first of the shard) whose scores seed the thresholds: a workgroup sees ~4 000
                                // rows, its lists never fill, so the seed IS its threshold - 64 k rows beat 32 k by 4 % per call, 128 k tie
struct LvsRqArgs {
    const void* xb;
    const void* xq;
    const float* bn;
    const float* qn;
    const uint32_t* row_ids;
    uint32_t* gtau;   // [nq] zero-initialised, or seeded with a valid lower bound of every query's k-th best score
    u64* out;         // [nparts][nq][k]
    float* seed_out;  // non-NULL: SEED mode - [nparts][nq] best score of every (corpus range, query) instead of lists
    long long nb, ldb, ldq, id_offset;
    int nq, k, metric;
    int blocks_per_wg;  // 32-row blocks per corpus range (set by lvs_rq_launch)
    int nparts;         // out: corpus ranges of the launch
    int groups;         // out: query groups of LVS_RQ_GROUPQ (sibling workgroups per corpus range)
    int debug;          // -DLVS_TUNING builds only (env LVS_RQ_DEBUG), timing ablations with WRONG results: bit 0 no staging loads in
                        // the loop, bit 1 no block epilogue, bit 2 no MFMAs, bit 3 no fragment reads, bit 4 no unit barrier
    int drain_every;    // lvs_rj_kernel: every wave empties its candidate buffer every that many blocks (a power of two)
    // lvs_rj_kernel in RANGE mode (threshold join, lvs_range_join): every (query, row) with score > threshold (strict, as
    // sem_dedup.py:46) is appended to the pair list; q_row0 >= 0: self-join, pairs with row id > query row only
    long long* pair_q;
    long long* pair_j;
    float* pair_s;
    unsigned long long* pair_count;
    long long pair_capacity, q_row0, q_base;  // q_base: added to the query numbers written to pair_q (the launch's slice of the call)
    float threshold, out_scale;
};
// blockIdx -> (corpus range, query group) of a grouped launch (lvs_rq_kernel, lvs_rj_kernel).  Workgroup b lands on XCD b % 8
// (observed; used for speed only): the siblings of a range take slots of ONE XCD, so the range comes from HBM once per XCD and
// reaches the siblings - which run in step - through that XCD's L2.  Up to 32 groups: 32 / groups ranges per XCD at a time
// (ranges 8 i + x on XCD x).  More than 32 groups (a multiple of 32; r6): the launch is 256 items in range-major order, XCD x runs
// items 32 x .. 32 x + 31 - its 32 workgroups share one range, 256 / groups ranges in all: every query sees FEWER, LONGER
// ranges, i.e. fewer list cold starts and candidates per query (16 groups x 16 ranges: ~24 candidates per query and range; 64 x 4: ~37).
__host__ __device__ inline int lvs_rq_ranges_for(int groups) {
    return groups <= 1 ? 256 : (groups <= 32 ? 8 * (32 / groups) : 256 / groups);
}
__host__ __device__ inline int lvs_rq_grid(int groups, int ranges) {
    return groups <= 1 ? ranges : (groups <= 32 ? 8 * groups * ((ranges + 7) / 8) : 256);
}
__host__ __device__ inline bool lvs_rq_item(int b, int groups, int nparts, int& range, int& group) {
    if (groups <= 1) {
        range = b;
        group = 0;
        return true;
    }
    if (groups <= 32) {
        const int s = b >> 3;
        range = (s / groups) * 8 + (b & 7);
        group = s % groups;
        return range < nparts;  // (the grid is rounded up to whole XCD rows)
    }
    const int item = (b & 7) * 32 + (b >> 3);
    range = item / groups;
    group = item % groups;
    return range < nparts;
}
bool lvs_rq_fits(int64_t nq, int64_t nb, int dpad, int k);
bool lvs_rq_shape_ok(int dpad, int k);
#define LVS_RQ_JOIN_MINROWS 65536  // shortest corpus (shard) a chunked call takes: 2 ranges of 1 024 blocks at 128 groups
hipError_t lvs_rq_launch(LvsRqArgs& a, int dpad, hipStream_t stream);
// ---- lvs_rj.hip: the same launches with ONE wave per SIMD and 64 queries per wave (B fragments in named accumulation registers)
#define LVS_RJ_DEFAULT 1    // launches lvs_rj_fits accepts go through lvs_rj_kernel (1) or lvs_rq_kernel (0)
bool lvs_rj_fits(int64_t nq, int64_t nb, int dpad, int k, bool has_row_ids);
hipError_t lvs_rj_launch(LvsRqArgs& a, int dpad, hipStream_t stream);  // whole 32-row blocks only: the caller adds the tail
hipError_t lvs_rj_range_launch(LvsRqArgs& a, int dpad, hipStream_t stream);  // the same geometry, RANGE epilogue (inner product)
#undef LVS_RQ_JOIN_MINROWS
#define LVS_RQ_JOIN_MINROWS 32768  // shortest corpus (shard) a chunked call takes (r6: 50 k x 50 k 5.3 -> 4.4 ms; at 20 000 rows the list kernel wins)

int lvs_stream_ranges(int64_t nb, int groups);
size_t lvs_stream_lds_bytes(int nbfrag, int nqb, int kcap);
int lvs_stream_plan(int64_t nq, int k, int nbfrag, int* out_kcap, int* out_nqb, int* out_groups);
hipError_t lvs_stream_launch(LvsStreamArgs& a, hipStream_t stream);

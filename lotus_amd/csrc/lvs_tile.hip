// The dominant kernel: 256 corpus rows x 256 queries per workgroup on the matrix cores with the selection fused in.
//
// Replaces faiss IndexFlat::search as reached from lotus/vector_store/faiss_vs.py:67,75 (top-k, k <= LVS2_KCAP per
// pass), the k = 1 L2 search of lotus/utils.py:62,65 (k-means assignment), the K = N score rows of the cascade callers
// (SCORES) and the all-pairs search of lotus/sem_ops/sem_dedup.py:45-46 (RANGE: threshold self-join).  Geometry:
//   score tile   256 corpus rows (MFMA M) x 256 queries (MFMA N): 128 flop per byte staged from L2
//   wave layout  2 (corpus) x 4 (queries); each wave 128 x 64 = 4 x 2 accumulators of v_mfma_f32_32x32x16_f16
//                (128 accumulator VGPRs), 32 MFMAs per wave between barriers; operands swapped (corpus = A, queries = B)
//                so that a lane owns a query column and its running threshold is a register
//   LDS          2 x 512 rows x 128 B staging (global_load_lds, source-side XOR swizzle) = 128 KB, which leaves 31 KB
//                for candidates: one sorted k-list per query (256 x KCAP x 8 B) plus a lock word per query.  A score
//                that beats its query's current k-th best is inserted at once by 16 cooperating lanes under the
//                query's LDS lock (hits are rare: ~k (1 + ln(N/k)) per query), so thresholds tighten immediately
//                and the epilogue needs no workgroup barrier.
//   K-step       inline-asm fragment reads with counted lgkmcnt waits, see below.
// Measurements and the tuning history: profiles/r01_tuning.md, DESIGN.md section 3.1.
#include <type_traits>
#include <utility>

#include "lvs_common.h"
#include "lvs_kstep.h"
#include "lvs_tile.h"

namespace {

using lvs_kstep::BC;
using lvs_kstep::BK;
using lvs_kstep::ROWB;
using lvs_kstep::glds16;
using lvs_kstep::static_for;

// Two geometries of the same kernel, selected by MI = 32-row accumulator blocks per wave in the corpus direction:
//   MI = 4: waves 2 (corpus) x 4 (queries), wave tile 128 x 64, 256 queries per workgroup, 15 list slots per query
//           (k <= 15 per pass) - the fast one, everything above describes it;
//   MI = 2: waves 4 x 2, wave tile 64 x 64, 128 queries per workgroup; the 32 KB of staging this frees hold 56 list
//           slots per query (16 <= k <= 56 per pass).  Top-k mode only.
template <int MI>
struct Geo {
    static_assert(MI == 4 || MI == 2, "two geometries");
    static constexpr int WN = MI;                    // wave columns (queries); wave rows WM = 8 / WN
    static constexpr int WM = 8 / WN;
    static constexpr int BQ = WN * 64;               // queries per workgroup
    static constexpr int QG = BQ / 64;               // staging loads (8 rows each) per wave for the query rows
    static constexpr int NF = 4 * MI;                // steps (2 MFMAs each) per K-step
    static constexpr int STAGE_BYTES = (BC + BQ) * ROWB;
    static constexpr int KCAP = MI == 4 ? LVS2_KCAP : LVS3_KCAP;
    static constexpr int OFF_LIST = 2 * STAGE_BYTES;             // u64 [BQ][KCAP] sorted descending, first k used
    static constexpr int OFF_LOCK = OFF_LIST + BQ * KCAP * 8;    // u32 [BQ] list locks
    static constexpr int OFF_EFF = OFF_LOCK + BQ * 4;            // u32 [BQ] banded lists (a.kc > 0): admission threshold per query
    static constexpr int LDS_TOTAL = OFF_EFF + BQ * 4;
    static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
    static_assert(KCAP <= 64, "one lane per list slot");
};

// Scores are finite or -inf, never NaN, so the maxima need none of fmaxf's canonicalisation (hipcc emits one extra
// `v_max_f32 x, x` per MFMA output to quiet signalling NaNs): v_max3_f32 directly - 8 instructions for 16 values.
__device__ inline float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ inline float max4(float a, float b, float c, float d) { return max3(max3(a, b, c), d, d); }
__device__ inline float max16(const f32x16& v) {
    const float a = max3(v[0], v[1], v[2]), b = max3(v[3], v[4], v[5]), c = max3(v[6], v[7], v[8]);
    const float d = max3(v[9], v[10], v[11]), e = max3(v[12], v[13], v[14]);
    return max3(max3(a, b, c), max3(d, e, v[15]), v[15]);
}
// wave-uniform "any lane" without materialising a per-lane bool
__device__ inline bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

__device__ inline float tau_float(uint32_t ord) { return ord == 0 ? -INFINITY : lvs_unord32(ord); }

// Every lane's first candidate (lowest accumulator register r with v[r] >= tf) of one 32 x 32 block, in straight-line
// code: per register one compare, two selects and two scalar mask operations - no per-row branch.  any_m = lanes with at
// least one candidate, multi_m = lanes with more than one (they need the row-by-row pass for the rest).
__device__ __forceinline__ void first_hit(const f32x16& v, float tf, bool qv, float& cs, int& cr, u64& any_m, u64& multi_m) {
    const float tfe = qv ? tf : INFINITY;  // lanes without a query never match
#pragma unroll
    for (int r = 15; r >= 0; --r) {
        const bool hit = v[r] >= tfe;
        const u64 m = __builtin_amdgcn_ballot_w64(hit);
        multi_m |= m & any_m;
        any_m |= m;
        cs = hit ? v[r] : cs;
        cr = hit ? r : cr;
    }
}

// blockIdx -> (query tile, slab).  Blocks are dealt to the XCDs round-robin (b % 8, observed; used for speed only) and an
// XCD runs 32 of them at a time, so 32 consecutive blocks of one XCD form a "group" that shares operands through that
// XCD's L2: gq query tiles x 32/gq slabs.  Groups walk the query tiles first, then the slabs, so a query's later slabs
// start with the thresholds its earlier slabs published.  With a.lead_slabs = 1 the first slab of EVERY query tile is
// done first in 32-wide groups (one slab at a time per query: a single cold start), and only the remaining slabs use
// the narrow groups whose 32/gq slabs per query run concurrently.
// The last G % 8 groups would occupy G % 8 XCDs for a whole item while the others idle: their slots are dealt across
// all eight XCDs instead (slot r of such a group runs on XCD r % 8), so every XCD ends within one item of the others.
__device__ inline bool item_of_block(const LvsTileArgs& a, int b, int& qt, int& slab) {
    const LvsTileGroups gr = lvs_tile_groups(a.nqt, a.nslab, a.gq, a.lead_slabs);
    int g, r;
    if (!lvs_tile_block_slot(gr, b, g, r)) return false;
    return lvs_tile_group_slot(a.nqt, a.nslab, a.gq, a.lead_slabs, gr, g, r, qt, slab);
}

}  // namespace

// MODE  TOPK: sorted lists (k <= KCAP); TOP1: k == 1, per-lane best; RANGE: threshold join; SCORES: matrix out;
//       COLLECT: every key >= a per-query threshold key into that query's bucket (second phase of large-k search)
template <int MODE, int MI>
__global__ __launch_bounds__(512, 2) void lvs_tile_kernel(const LvsTileArgs a) {
    using G = Geo<MI>;
    constexpr int BQ = G::BQ, QG = G::QG, KCAP = G::KCAP, STAGE_BYTES = G::STAGE_BYTES;
    constexpr int OFF_LIST = G::OFF_LIST, OFF_LOCK = G::OFF_LOCK;
    static_assert(MODE == LVS_MODE_TOPK || MI == 4, "the other epilogues exist for the 256 x 256 geometry only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / G::WN, wn = wave % G::WN;

    int qt, slab;
    if (!item_of_block(a, blockIdx.x, qt, slab)) return;
    // predicated launch (fallback passes of the large-k search): nothing to do unless the device-side flag is set
    if (a.pred && __builtin_nontemporal_load(a.pred) == 0u) return;
    const long long q0 = (long long)qt * BQ;
    int tile0_ = slab * a.tiles_per_slab;
    const int tile1 = min(a.ntiles, tile0_ + a.tiles_per_slab);
    if (MODE == LVS_MODE_RANGE) {
        if (a.qt_stride > 1 && (qt % a.qt_stride) != a.qt_phase) return;
        if (a.q_row0 >= 0) {  // self-join: tiles whose rows are all <= every query row of this tile hold no j > i
            const long long first = (a.q_row0 + q0 - a.id_offset) / BC;
            if (first > tile0_) tile0_ = (int)(first < tile1 ? first : tile1);
        }
    }
    const int tile0 = tile0_;
    if (tile0 >= tile1) return;

    u64* lists = (u64*)(smem + OFF_LIST);
    uint32_t* locks = (uint32_t*)(smem + OFF_LOCK);
    uint32_t* eff = (uint32_t*)(smem + Geo<MI>::OFF_EFF);
    float* bnl = (float*)(smem + OFF_LIST);           // TOP1: |y|^2 of the current corpus tile [BC]
    u64* part = (u64*)(smem + OFF_LIST + BC * 4);     // TOP1: [BQ][4] per-lane partial best keys
    float* psec = (float*)(smem + OFF_LIST + BC * 4 + BQ * 4 * 8);  // TOP2: [BQ][4] per-lane second-best scores

    const _Float16* __restrict__ xb = (const _Float16*)a.xb;
    const _Float16* __restrict__ xq = (const _Float16*)a.xq;
    const long long ldb = a.ldb, ldq = a.ldq;
    const int nk = a.nk, nkd = a.nkd;
    const int k = a.k;

    // ---- staging addresses: wave stages corpus rows [wave*32, +32) and query rows [wave*8*QG, +8*QG) ----------
    const int srow = lane >> 3, sp = lane & 7;
    int s_row[4], s_col[4], qs_row[QG], qs_col[QG];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = wave * 32 + i * 8 + srow;
        s_row[i] = row;
        s_col[i] = (sp ^ ((row >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < QG; ++i) {
        int row = wave * (8 * QG) + i * 8 + srow;
        qs_row[i] = row;
        qs_col[i] = (sp ^ ((row >> 1) & 7)) * 8;
    }
    const _Float16* q_src[QG];
#pragma unroll
    for (int i = 0; i < QG; ++i) {
        long long grow = q0 + qs_row[i];
        if (grow > a.nq - 1) grow = a.nq - 1;
        q_src[i] = xq + grow * ldq + qs_col[i];
    }

    auto stage = [&](int t, int buf) {
        int ti = t / nk, ks = t - ti * nk;
        int seg = ks / nkd, r = ks - seg * nkd;
        int qcol = a.seg_q[seg] + r * BK;
        int ccol = a.seg_c[seg] + r * BK;
        long long trow0 = (long long)(tile0 + ti) * BC;
        char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long grow = trow0 + s_row[i];
            if (grow > a.nb - 1) grow = a.nb - 1;
            glds16(xb + grow * ldb + ccol + s_col[i], base + (wave * 32 + i * 8) * ROWB);
        }
#pragma unroll
        for (int i = 0; i < QG; ++i) glds16(q_src[i] + qcol, base + BC * ROWB + (wave * (8 * QG) + i * 8) * ROWB);
    };

    int foff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        foff[kk] = (lane & 31) * ROWB + ((((kk * 2) + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    const int a_base = (wm * MI * 32) * ROWB;
    const int b_base = BC * ROWB + (wn * 64) * ROWB;

    int qloc[2];
    bool qvalid[2];
    float tauf[2];
    uint32_t gord[2];
    u64 ubk[2];
    float qnv[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        qloc[ni] = wn * 64 + ni * 32 + (lane & 31);
        qvalid[ni] = (q0 + qloc[ni]) < a.nq;
        tauf[ni] = -INFINITY;
        gord[ni] = 0;
        ubk[ni] = ~0ull;
        qnv[ni] = 0.f;
        if (qvalid[ni]) {
            if (MI != 4 && a.ub) ubk[ni] = a.ub[(q0 + qloc[ni]) * a.ub_stride];  // multi-pass (k > 56) runs on the 128-query geometry only
            if (a.metric == LVS_METRIC_L2) qnv[ni] = a.qn[q0 + qloc[ni]];
            if (MODE == LVS_MODE_COLLECT) {  // ubk = LOWER bound key here, tauf = its score
                ubk[ni] = a.thr_key[q0 + qloc[ni]];
                tauf[ni] = tau_float((uint32_t)(ubk[ni] >> 32));
            }
        }
    }

    if (MODE == LVS_MODE_TOPK) {
        for (int i = tid; i < BQ * KCAP; i += 512) lists[i] = 0;
        for (int i = tid; i < BQ; i += 512) locks[i] = 0;
        for (int i = tid; i < BQ; i += 512) eff[i] = 0;
    }
    float bestv[2] = {-INFINITY, -INFINITY};
    float secv[2] = {-INFINITY, -INFINITY};  // TOP2: second-best score seen by this lane
    uint32_t besti[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    // TOP2 under L2: running best / runner-up of u = 2 q.y - |y|^2 (the query norm and the clamp are applied once, at the
    // end), the position of a score inside the tile (6 bits) carried in the low mantissa bits; first row of the winner's tile
    float bestu[2] = {-INFINITY, -INFINITY}, secu[2] = {-INFINITY, -INFINITY};
    uint32_t bestrow0[2] = {0u, 0u};

    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // per-lane byte offsets of the staging loads relative to the tile's / query tile's first row
    unsigned c_loff[4], q_loff[QG];
#pragma unroll
    for (int i = 0; i < QG; ++i) {
        long long qrow = q0 + qs_row[i];
        if (qrow > a.nq - 1) qrow = a.nq - 1;
        q_loff[i] = (unsigned)(((qrow - q0) * ldq + qs_col[i]) * 2);
    }
    auto set_tile_offsets = [&](int tile_rel) {  // rows past the end of the shard re-read the last valid row
        const long long trow0 = (long long)(tile0 + tile_rel) * BC;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long grow = trow0 + s_row[i];
            if (grow > a.nb - 1) grow = a.nb - 1;
            c_loff[i] = (unsigned)(((grow - trow0) * ldb + s_col[i]) * 2);
        }
    };
    set_tile_offsets(0);
    int n_tile = 0, n_seg = 0, n_r = 0;  // (tile, segment, k-block) of the K-step being prefetched

    const int T = (tile1 - tile0) * nk;
    stage(0, 0);
    int ks_in_tile = 0, ti = 0;
    // the second-dispatched half of the waves loses VALU/MFMA arbitration to the older half on every segment: one
    // static priority raise evens it out (+2 % on the bare loop, tools/probe_gemm.hip)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    uint32_t gpre[2] = {0u, 0u};  // cross-workgroup thresholds, prefetched one K-step before the tile epilogue
    uint32_t lpre[2] = {0u, 0u};  // own lists' thresholds, read in the same K-step (nobody inserts between epilogues)
#ifdef LVS_COUNT_EVENTS
    unsigned n_visit = 0, n_ins = 0, n_wt = 0;  // tuning aid, see a.dbg
    unsigned long long c_filter = 0, c_visit = 0, c_ins = 0;  // cycles (s_memtime) in the filter / visit loop / insertions
#endif
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
#ifdef LVS_TUNING
        if (a.debug_hot == 4) {  // tuning aid: no wait for the staging loads (results are garbage, timing only)
            __builtin_amdgcn_s_barrier();
        } else
#endif
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (MODE == LVS_MODE_TOPK && ks_in_tile == nk - 1) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if (qvalid[ni] && !a.no_share) gpre[ni] = a.gtau[q0 + qloc[ni]];
                // every wave passed this K-step's barrier, so every insertion of the previous epilogue is in the list
                // score half of the k-th key; banded lists: the admission threshold the last insertion left (>= that score)
                lpre[ni] = *(a.kc ? (const uint32_t*)(eff + qloc[ni]) : (const uint32_t*)(lists + qloc[ni] * KCAP + k - 1) + 1);
            }
        }
        // ---- one K-step, software-pipelined by hand: 4*MI steps f = kk*MI + mi of {A-fragment read two steps ahead,
        // one staging load of the NEXT K-step (first 4 + QG steps), 2 MFMAs}; B fragments double-buffered per kk ----
        const char* sb = smem + buf * STAGE_BYTES;
        // next K-step's source: uniform 64-bit bases (SGPRs) + constant per-lane 32-bit byte offsets, advanced
        // incrementally (no divisions, no per-load 64-bit VALU).  The step after the last one re-loads the last
        // K-step into the idle buffer (never read), so there is no branch around the loads.
        char* n_base = smem + (buf ^ 1) * STAGE_BYTES;
        if (t + 1 < T) {
            if (++n_r == nkd) {
                n_r = 0;
                if (++n_seg == a.nseg) {
                    n_seg = 0;
                    ++n_tile;
                    set_tile_offsets(n_tile);
                }
            }
        }
        const char* c_sbase = (const char*)xb + ((long long)(tile0 + n_tile) * BC * ldb + a.seg_c[n_seg] + n_r * BK) * 2;
        const char* q_sbase = (const char*)xq + (q0 * ldq + a.seg_q[n_seg] + n_r * BK) * 2;
        lvs_kstep::run<MI, QG>(sb, n_base, c_sbase, q_sbase, c_loff, q_loff, wave, a_base, b_base, foff, acc);

        if (++ks_in_tile < nk) continue;
        ks_in_tile = 0;
        // =============================== tile epilogue ===================================================
        const long long trow0 = (long long)(tile0 + ti) * BC;
        ++ti;
        const int lrow_base = wm * (MI * 32) + 4 * (lane >> 5);  // + mi*32 + (r&3) + 8*(r>>2)

        if constexpr (MODE == LVS_MODE_RANGE || MODE == LVS_MODE_SCORES || MODE == LVS_MODE_COLLECT) {
            if (a.metric == LVS_METRIC_L2) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long long row = trow0 + lrow_base + mi * 32 + (r & 3) + 8 * (r >> 2);
                        const float bnv = row < a.nb ? a.bn[row] : 0.f;
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            acc[mi][ni][r] = -fmaxf((qnv[ni] + bnv) - 2.0f * acc[mi][ni][r], 0.f);
                    }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const long long qg = q0 + qloc[ni];
                    if constexpr (MODE == LVS_MODE_COLLECT) {
                        // tauf / ubk were loaded from a.thr_key: score part (fast filter) and full key (exact test)
                        const bool th = qvalid[ni] && (max16(acc[mi][ni]) >= tauf[ni]);
                        if (!wave_any(th)) continue;
                        if (!th) continue;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float s = acc[mi][ni][r];
                            if (!(s >= tauf[ni])) continue;
                            const long long row = trow0 + lrow_base + mi * 32 + (r & 3) + 8 * (r >> 2);
                            if (row >= a.nb) continue;
                            const uint32_t id = a.row_ids ? a.row_ids[row] : (uint32_t)(row + a.id_offset);
                            const u64 key = lvs_pack_key(s, id);
                            if (key < ubk[ni]) continue;  // below the threshold KEY (same score, larger id)
                            const uint32_t pos = atomicAdd(&a.bucket_count[qg], 1u);
                            if (pos < (uint32_t)a.bucket_capacity) a.bucket[qg * a.bucket_capacity + pos] = key;
                        }
                    } else if (MODE == LVS_MODE_SCORES) {
                        if (!qvalid[ni]) continue;
                        float* orow = a.scores + qg * a.ld_scores;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const long long row = trow0 + lrow_base + mi * 32 + (r & 3) + 8 * (r >> 2);
                            if (row < a.nb) orow[row] = acc[mi][ni][r] * a.out_scale;
                        }
                    } else {
                        const bool th = qvalid[ni] && (max16(acc[mi][ni]) > a.threshold);
                        if (!__any(th)) continue;
                        if (!th) continue;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float s = acc[mi][ni][r];
                            if (!(s > a.threshold)) continue;  // strict, as sem_dedup.py:46
                            const long long row = trow0 + lrow_base + mi * 32 + (r & 3) + 8 * (r >> 2);
                            if (row >= a.nb) continue;
                            const long long jg = row + a.id_offset;
                            if (a.q_row0 >= 0 && jg <= a.q_row0 + qg) continue;
                            const unsigned long long pos = atomicAdd(a.pair_count, 1ull);
                            if ((long long)pos < a.pair_capacity) {
                                a.pair_q[pos] = qg + a.q_base;
                                a.pair_j[pos] = jg;
                                a.pair_s[pos] = s * a.out_scale;
                            }
                        }
                    }
                }
        } else if constexpr (MODE == LVS_MODE_TOP1 || MODE == LVS_MODE_TOP2) {
            // ---- k == 1: per-lane running best, no lists.  Rows are visited in increasing order, so a strict
            // "greater" keeps the lowest row among equal scores (the oracle's tie rule).  TOP2 also keeps the
            // second-best SCORE (an equal score counts: margin 0), from which the caller certifies the winner. ----
            {
                __syncthreads();
                if (tid < BC) {
                    const long long row = trow0 + tid;
                    // |y|^2 (0 under inner product); rows past the end never win (TOP2 tags scores in their mantissa: a
                    // finite sentinel, not inf)
                    const float nv = a.metric == LVS_METRIC_L2 ? a.bn[row < a.nb ? row : a.nb - 1] : 0.f;
                    bnl[tid] = row < a.nb ? nv : (MODE == LVS_MODE_TOP2 ? 3.0e38f : INFINITY);
                }
                __syncthreads();
            }
            if constexpr (MODE == LVS_MODE_TOP2) {
                // Certified nearest row (lvs_nearest_hi): the caller only needs a winner, its margin over the runner-up
                // and an approximate score, so each score costs four VALU operations instead of eight: u = c s - |y|^2
                // (one fma; c = 2 under L2, where |q|^2 and the clamp at 0 do not change the order and are applied at the
                // end; c = 1 and |y|^2 := 0 under inner product), the score's position in the tile (mi * 16 + r, an inline
                // constant) replaces its low six mantissa bits (a relative perturbation < 2^-17 that the caller's margin
                // bound includes), best = max, runner-up = med3.
                const float before0 = bestu[0], before1 = bestu[1];
                const float cs = a.metric == LVS_METRIC_L2 ? 2.0f : 1.0f;
                static_for<MI>([&](auto mic) {
                    constexpr int mi = decltype(mic)::value;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 bn4 = *(const f32x4*)(bnl + lrow_base + mi * 32 + 8 * r4);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni) {
                                const float u = __builtin_fmaf(cs, acc[mi][ni][r4 * 4 + e], -bn4[e]);
                                const float up = __uint_as_float((__float_as_uint(u) & 0xFFFFFFC0u) | (uint32_t)(mi * 16 + r4 * 4 + e));
                                secu[ni] = __builtin_amdgcn_fmed3f(up, bestu[ni], secu[ni]);
                                bestu[ni] = fmaxf(bestu[ni], up);
                            }
                        __builtin_amdgcn_sched_barrier(0);  // keep the norm reads next to their uses (register footprint)
                    }
                });
                if (bestu[0] != before0) bestrow0[0] = (uint32_t)trow0;
                if (bestu[1] != before1) bestrow0[1] = (uint32_t)trow0;
            } else {
                // TOP1 (exact): the same one-fma order value u = c s - |y|^2; a strict "greater" keeps the lowest row among
                // equal values, the position inside the tile is selected as an inline constant, the winner's tile is
                // noted once per tile - four VALU operations per score instead of eight.
                const float before0 = bestu[0], before1 = bestu[1];
                const float cs = a.metric == LVS_METRIC_L2 ? 2.0f : 1.0f;
                static_for<MI>([&](auto mic) {
                    constexpr int mi = decltype(mic)::value;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 bn4 = *(const f32x4*)(bnl + lrow_base + mi * 32 + 8 * r4);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni) {
                                const float u = __builtin_fmaf(cs, acc[mi][ni][r4 * 4 + e], -bn4[e]);
                                besti[ni] = u > bestu[ni] ? (uint32_t)(mi * 16 + r4 * 4 + e) : besti[ni];
                                bestu[ni] = fmaxf(bestu[ni], u);
                            }
                        __builtin_amdgcn_sched_barrier(0);  // keep the norm reads next to their uses (register footprint)
                    }
                });
                if (bestu[0] != before0) bestrow0[0] = (uint32_t)trow0;
                if (bestu[1] != before1) bestrow0[1] = (uint32_t)trow0;
            }
        } else {
        if (a.metric == LVS_METRIC_L2) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        long long row = trow0 + lrow_base + mi * 32 + (r & 3) + 8 * (r >> 2);
                        float bnv = row < a.nb ? a.bn[row] : 0.f;
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            float dis = (qnv[ni] + bnv) - 2.0f * acc[mi][ni][r];
                            acc[mi][ni][r] = -fmaxf(dis, 0.f);
                        }
                    }
            }
        if constexpr (MODE == LVS_MODE_SEED) {
            // the scores are the ones the list mode ranks (same K order, same expression): keep the best per lane.  The
            // caller hands over whole tiles only (nb a multiple of BC), so every score belongs to a real row.
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) bestv[ni] = fmaxf(bestv[ni], max16(acc[mi][ni]));
        } else {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            if (qvalid[ni]) {
                const uint32_t g = gpre[ni];
                gord[ni] = g > gord[ni] ? g : gord[ni];
                tauf[ni] = fmaxf(tauf[ni], tau_float(gord[ni]));
            }
        // own lists' thresholds (they may have risen since this lane last looked: the wave sharing these queries inserts too)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) tauf[ni] = fmaxf(tauf[ni], tau_float(lpre[ni]));
        // fast filter: which of the wave's 2*MI 32x32 blocks hold a score that may enter some query's list?
        uint32_t hitmask = 0;  // wave-uniform, bit tsel = mi * 2 + ni
#ifdef LVS_COUNT_EVENTS
        ++n_wt;
        const unsigned long long tm0 = __builtin_amdgcn_s_memtime();
#endif
        // lanes without a query compare against +inf: no exec-masked branch around each block's max16
        const float tfe[2] = {qvalid[0] ? tauf[0] : INFINITY, qvalid[1] ? tauf[1] : INFINITY};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                if (wave_any(max16(acc[mi][ni]) >= tfe[ni])) hitmask |= 1u << (mi * 2 + ni);
#ifdef LVS_TUNING
        if (a.debug_hot == 2) hitmask = 0;  // tuning aid: skip the slow path (results are wrong, timing only)
#endif
#ifdef LVS_COUNT_EVENTS
        const unsigned long long tm1 = __builtin_amdgcn_s_memtime();
        c_filter += tm1 - tm0;
#endif
        {
            // Wave-cooperative insertion of the wave's pending hits (at most one per lane), one at a time: the hit is
            // broadcast, lane j < 16 owns slot j of that query's sorted list and computes its new content in ONE step
            // (new[j] = L[j] if L[j] > key, else key if L[j-1] > key, else L[j-1]) - cost independent of k.
            // The list's lock is taken by lane 0 only (waves wm = 0, 1 share queries).
            auto insert_pending = [&](bool pending, u64 key, int q, float& tf) __attribute__((always_inline)) {
                unsigned long long pm = __ballot(pending);
#ifdef LVS_TUNING
                if (a.debug_hot == 3) pm = 0;  // tuning aid: scan for hits but skip the insertions
#endif
                while (pm) {
#ifdef LVS_COUNT_EVENTS
                    ++n_ins;
                    const unsigned long long ti0 = __builtin_amdgcn_s_memtime();
#endif
                    const int src = __ffsll((long long)pm) - 1;
                    pm &= pm - 1;
                    const uint32_t klo = __builtin_amdgcn_readlane((uint32_t)key, src);
                    const uint32_t khi = __builtin_amdgcn_readlane((uint32_t)(key >> 32), src);
                    const u64 ukey = ((u64)khi << 32) | klo;
                    const int uq = __builtin_amdgcn_readlane(q, src);
                    // Lock + slot reads under ONE wait: lane 0 issues the compare-and-swap, lanes < k issue their
                    // slot reads right behind it.  LDS executes a wave's DS instructions in order, so when the
                    // swap succeeded the reads saw the list under the lock; otherwise everything is retried.
                    u64* UL = lists + uq * KCAP;
                    u64 mine = 0;
                    for (;;) {
                        uint32_t seen = 0;
                        if (lane == 0) {
                            __hip_atomic_compare_exchange_strong(&locks[uq], &seen, 1u, __ATOMIC_RELAXED,
                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        asm volatile("" ::: "memory");  // keep the reads behind the swap in program order
                        if (lane < k) mine = UL[lane];
                        if (__builtin_amdgcn_readfirstlane(seen) == 0) break;  // lock word was 0: we own it
                    }
                    // slot j - 1 is the neighbouring lane's `mine` (DPP wave_shr:1; lane 0 keeps the "old" operand = all
                    // ones: nothing is above slot 0) - one LDS read per insertion less
                    const uint32_t plo = __builtin_amdgcn_update_dpp(0xFFFFFFFFu, (uint32_t)mine, 0x138, 0xF, 0xF, false);
                    const uint32_t phi = __builtin_amdgcn_update_dpp(0xFFFFFFFFu, (uint32_t)(mine >> 32), 0x138, 0xF, 0xF, false);
                    const u64 prev = ((u64)phi << 32) | plo;
                    u64 newv = 0;
                    if (lane < k) newv = mine > ukey ? mine : (prev > ukey ? ukey : prev);
                    __builtin_amdgcn_wave_barrier();
                    if (lane < k) UL[lane] = newv;
                    uint32_t ntau = __builtin_amdgcn_readlane((uint32_t)(newv >> 32), k - 1);
                    if (a.kc) {
                        // Banded list (the certified one-pass search, lvs_flat_search_keys_hi_banded): a row further than the
                        // query's band below the CURRENT kc-th best can never be needed by the certificate, so the admission
                        // threshold is max(last slot, kc-th slot - band) - a list of k slots then costs the insertions of a
                        // list of kc, and slots stay empty unless the band is crowded.  The threshold only ever rises.
                        const uint32_t nkc = __builtin_amdgcn_readlane((uint32_t)(newv >> 32), a.kc - 1);
                        if (nkc) {
                            const float band = a.bscale * __builtin_sqrtf(a.qn[q0 + uq]) + a.bslack;  // uniform: a scalar load
                            const uint32_t ob = lvs_ord32(lvs_unord32(nkc) - band);
                            ntau = ob > ntau ? ob : ntau;
                        }
                        if (lane == 0) eff[uq] = ntau;  // under the lock, before the unlock store below
                    }
                    // unlock: the LDS executes one wave's DS instructions in issue order, so a plain (relaxed)
                    // store issued after the slot writes is observed after them - no wait for the writes needed
                    asm volatile("" ::: "memory");
                    if (lane == 0)
                        __hip_atomic_store(&locks[uq], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (q == uq) tf = fmaxf(tf, tau_float(ntau));
#ifdef LVS_COUNT_EVENTS
                    c_ins += __builtin_amdgcn_s_memtime() - ti0;
#endif
                }
            };
            // key of a candidate (score s at accumulator register r of the 32 x 32 block whose first row is rbase) and
            // whether it takes part: inside the shard, below this pass's upper bound, not below the shared threshold
            auto make_key = [&](float s, int r, long long rbase, u64 ubq, uint32_t go, u64& key) __attribute__((always_inline)) {
                const long long row = rbase + (r & 3) + 8 * (r >> 2);
                if (row >= a.nb) return false;
                const uint32_t id = a.row_ids ? a.row_ids[row] : (uint32_t)(row + a.id_offset);
                key = lvs_pack_key(s, id);
                return key < ubq && (uint32_t)(key >> 32) >= go;  // re-checked under the lock
            };

            while (hitmask) {  // rare path: only blocks with candidates are visited
                const int tsel = __builtin_ctz(hitmask);
                hitmask &= hitmask - 1;
                const int mi = tsel >> 1, ni = tsel & 1;
                float tf = ni ? tauf[1] : tauf[0];
                const bool qv = ni ? qvalid[1] : qvalid[0];
                // ---- fast extraction: every lane's FIRST candidate of the block in straight-line code (no copy of the
                // block, no per-row wave-uniform branch); `multi` = lanes holding more than one
                float cs = 0.f;
                int cr = -1;
                u64 anym = 0, multi = 0;
                switch (tsel) {  // wave-uniform branch table; cases >= 2 * MI are never taken
                    case 0: first_hit(acc[0][0], tf, qv, cs, cr, anym, multi); break;
                    case 1: first_hit(acc[0][1], tf, qv, cs, cr, anym, multi); break;
                    case 2: first_hit(acc[1][0], tf, qv, cs, cr, anym, multi); break;
                    case 3: first_hit(acc[1][1], tf, qv, cs, cr, anym, multi); break;
                    case 4: first_hit(acc[2 % MI][0], tf, qv, cs, cr, anym, multi); break;
                    case 5: first_hit(acc[2 % MI][1], tf, qv, cs, cr, anym, multi); break;
                    case 6: first_hit(acc[3 % MI][0], tf, qv, cs, cr, anym, multi); break;
                    default: first_hit(acc[3 % MI][1], tf, qv, cs, cr, anym, multi); break;
                }
                if (anym == 0) continue;  // the thresholds rose since the filter looked
#ifdef LVS_COUNT_EVENTS
                ++n_visit;
#endif
                const int q = ni ? qloc[1] : qloc[0];
                const u64 ubq = MI == 4 ? ~0ull : (ni ? ubk[1] : ubk[0]);  // no upper bound on the 256-query geometry (one pass)
                const uint32_t go = ni ? gord[1] : gord[0];
                const long long rbase = trow0 + lrow_base + mi * 32;
                // the visiting wave is on the workgroup's critical path (the next barrier waits for it): let its
                // instructions win the issue arbitration against the other wave's MFMAs
                __builtin_amdgcn_s_setprio(3);
                {
                    u64 key = 0;
                    const bool pending = cr >= 0 && make_key(cs, cr, rbase, ubq, go, key);
                    insert_pending(pending, key, q, tf);
                }
                if (multi) {
                    // some lane holds two or more candidates in this block (cold lists, duplicate-heavy data): the
                    // row-by-row pass takes the rest against the thresholds the first candidates just raised
                    f32x16 tv;
                    switch (tsel) {  // a branch table + 16 moves (a select chain would cost 7 x 16 VALU)
                        case 0: tv = acc[0][0]; break;
                        case 1: tv = acc[0][1]; break;
                        case 2: tv = acc[1][0]; break;
                        case 3: tv = acc[1][1]; break;
                        case 4: tv = acc[2 % MI][0]; break;
                        case 5: tv = acc[2 % MI][1]; break;
                        case 6: tv = acc[3 % MI][0]; break;
                        default: tv = acc[3 % MI][1]; break;
                    }
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {  // rows 8g .. 8g+3: skip the group when none of its four scores can enter
                        const float m4 = max4(tv[4 * g4], tv[4 * g4 + 1], tv[4 * g4 + 2], tv[4 * g4 + 3]);
                        if (!wave_any(qv && m4 >= tf)) continue;
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            const int r = 4 * g4 + e4;
                            const float s = tv[r];
                            const bool cand = qv && s >= tf && r != cr;  // (lane, cr) went through the fast path
                            if (!wave_any(cand)) continue;               // wave-uniform skip before any per-lane work
                            u64 key = 0;
                            const bool pending = cand && make_key(s, r, rbase, ubq, go, key);
                            insert_pending(pending, key, q, tf);
                        }
                    }
                }
                if (wave >= 4) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                if (ni) tauf[1] = fmaxf(tauf[1], tf); else tauf[0] = fmaxf(tauf[0], tf);
            }
        }
#ifdef LVS_COUNT_EVENTS
        c_visit += __builtin_amdgcn_s_memtime() - tm1;
#endif
        // publish thresholds for the other slabs of these queries (several slabs of one query tile may run at the same time)
        if (!a.no_share && (ti <= 8 || ((ti & 7) == 0) || t + 1 == T)) {  // every tile while the lists fill, then every 8
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const uint32_t lo = a.kc ? eff[qloc[ni]] : (uint32_t)(lists[qloc[ni] * KCAP + k - 1] >> 32);
                if (qvalid[ni] && lane < 32 && wm == 0 && lo > gord[ni]) atomicMax(&a.gtau[q0 + qloc[ni]], lo);
            }
        }
        }  // TOPK
        }  // MODE
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    }

#ifdef LVS_COUNT_EVENTS
    if (MODE == LVS_MODE_TOPK && a.dbg && lane == 0) {
        atomicAdd(&a.dbg[0], (unsigned long long)n_visit);
        atomicAdd(&a.dbg[1], (unsigned long long)n_ins);
        atomicAdd(&a.dbg[2], (unsigned long long)n_wt);
        atomicAdd(&a.dbg[3], c_filter);
        atomicAdd(&a.dbg[4], c_visit);
        atomicAdd(&a.dbg[5], c_ins);
    }
#endif
    if constexpr (MODE == LVS_MODE_RANGE || MODE == LVS_MODE_SCORES || MODE == LVS_MODE_COLLECT) return;
    if constexpr (MODE == LVS_MODE_SEED) {
        // four lanes (l, l + 32 of waves wm = 0, 1) hold partial maxima of each query
        float* pmax = (float*)part;
        __syncthreads();
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) pmax[qloc[ni] * 4 + wm * 2 + (lane >> 5)] = bestv[ni];
        __syncthreads();
        if (tid < BQ && q0 + tid < a.nq)
            a.seed_out[(long long)slab * a.nq + q0 + tid] =
                fmaxf(fmaxf(pmax[tid * 4], pmax[tid * 4 + 1]), fmaxf(pmax[tid * 4 + 2], pmax[tid * 4 + 3]));
        return;
    }
    if constexpr (MODE == LVS_MODE_TOP1 || MODE == LVS_MODE_TOP2) {
        {
            // row from (tile, position: TOP2 carries it in the value's low mantissa bits, TOP1 beside it); score =
            // -max(|q|^2 - u, 0) under L2 ("better" domain)
            const bool l2 = a.metric == LVS_METRIC_L2;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if (bestu[ni] > -1.0e38f) {
                    const uint32_t tag = MODE == LVS_MODE_TOP2 ? (__float_as_uint(bestu[ni]) & 63u) : besti[ni];
                    const uint32_t r = tag & 15u;
                    besti[ni] = bestrow0[ni] + (uint32_t)(wm * (MI * 32) + 4 * (lane >> 5)) + (tag >> 4) * 32u + (r & 3u) + 8u * (r >> 2);
                    bestv[ni] = l2 ? -fmaxf(qnv[ni] - bestu[ni], 0.f) : bestu[ni];
                    secv[ni] = secu[ni] > -1.0e38f ? (l2 ? -fmaxf(qnv[ni] - secu[ni], 0.f) : secu[ni]) : -INFINITY;
                }
            }
        }
        // four lanes (l, l+32 of waves wm = 0, 1) hold partial winners of each query: combine by key
        __syncthreads();
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            u64 key = 0;
            if (besti[ni] != 0xFFFFFFFFu) key = lvs_pack_key(bestv[ni], (uint32_t)(besti[ni] + a.id_offset));
            part[qloc[ni] * 4 + wm * 2 + (lane >> 5)] = key;
            if constexpr (MODE == LVS_MODE_TOP2) psec[qloc[ni] * 4 + wm * 2 + (lane >> 5)] = secv[ni];
        }
        __syncthreads();
        if (tid < BQ && q0 + tid < a.nq) {
            u64 b = part[tid * 4];
#pragma unroll
            for (int j = 1; j < 4; ++j) b = part[tid * 4 + j] > b ? part[tid * 4 + j] : b;
            a.out[(long long)slab * a.nq + q0 + tid] = b;
            if constexpr (MODE == LVS_MODE_TOP2) {
                // runner-up score of the slab: the winner lane's own second best, every other lane's best
                float sec = -INFINITY;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u64 pj = part[tid * 4 + j];
                    const float cand = pj == b ? psec[tid * 4 + j] : (pj ? lvs_unord32((uint32_t)(pj >> 32)) : -INFINITY);
                    sec = fmaxf(sec, cand);
                }
                a.out_second[(long long)slab * a.nq + q0 + tid] = sec;
            }
        }
        return;
    }
    // lists are sorted and complete: write the slab's candidates
    __syncthreads();
    for (int i = tid; i < BQ * k; i += 512) {
        int q = i / k, j = i - q * k;
        if (q0 + q < a.nq) a.out[((long long)slab * a.nq + q0 + q) * k + j] = lists[q * KCAP + j];
    }
}

template <int MODE, int MI>
static hipError_t launch_one(const LvsTileArgs& a, hipStream_t stream) {
    static LvsPerDeviceOnce attr;  // one per instantiation; the attribute is a per-device property
    constexpr int lds = Geo<MI>::LDS_TOTAL;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!attr.done(dev, (size_t)lds)) {
        e = hipFuncSetAttribute((const void*)lvs_tile_kernel<MODE, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr.set(dev, (size_t)lds);
    }
    dim3 grid(lvs_tile_grid_blocks(a.nqt, a.nslab, a.gq, a.lead_slabs)), block(512);
    hipLaunchKernelGGL((lvs_tile_kernel<MODE, MI>), grid, block, lds, stream, a);
    return hipGetLastError();
}

// a.bq names the geometry a.nqt was computed for: LVS2_BQ (k <= LVS2_KCAP and the TOP1 / RANGE / SCORES modes) or
// LVS3_BQ (top-k with k <= LVS3_KCAP)
hipError_t lvs_tile_launch(int mode, const LvsTileArgs& a, hipStream_t stream) {
    if (a.bq == LVS3_BQ) {
        if (mode != LVS_MODE_TOPK || a.k > LVS3_KCAP) return hipErrorInvalidValue;
        return launch_one<LVS_MODE_TOPK, 2>(a, stream);
    }
    if (a.bq != LVS2_BQ || a.k > LVS2_KCAP || a.ub) return hipErrorInvalidValue;
    if (mode == LVS_MODE_TOP1) return launch_one<LVS_MODE_TOP1, 4>(a, stream);
    if (mode == LVS_MODE_TOP2) return a.out_second ? launch_one<LVS_MODE_TOP2, 4>(a, stream) : hipErrorInvalidValue;
    if (mode == LVS_MODE_RANGE) return launch_one<LVS_MODE_RANGE, 4>(a, stream);
    if (mode == LVS_MODE_SCORES) return launch_one<LVS_MODE_SCORES, 4>(a, stream);
    if (mode == LVS_MODE_COLLECT) return launch_one<LVS_MODE_COLLECT, 4>(a, stream);
    if (mode == LVS_MODE_SEED) return (a.seed_out && a.nb % LVS_BC == 0) ? launch_one<LVS_MODE_SEED, 4>(a, stream) : hipErrorInvalidValue;
    return launch_one<LVS_MODE_TOPK, 4>(a, stream);
}

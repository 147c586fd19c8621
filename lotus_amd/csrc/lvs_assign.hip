// Nearest-row search against a SMALL corpus with the queries streaming: the k-means assignment (lotus/utils.py:62,65 ->
// faiss Kmeans.train / index.search(x, 1): 10 M points x 1 024 centroids at BASELINE configs[4]).
//
// lvs_tile_kernel<TOP2> walks corpus tiles for a fixed query tile: with a four-tile corpus every 256-point tile is staged
// four times (12.7 GB of fabric reads for 3.07 GB of points, L2 hit rate 49 %, profiles/r04b_kernels.json) and an item is
// only 48 K-steps long.  Here the loop is turned around: a workgroup keeps ONE corpus tile (256 centroids) and streams a slab
// of query tiles (points) past it; the workgroups that hold the other corpus tiles walk the same slab at the same time on the
// same XCD (XCD group = gq corpus tiles x 32 / gq query slabs), so a point tile is fetched from the fabric once and served
// from that XCD's L2 to the others, while the 1.5 MB of centroids stay L2-resident.  MFMA roles are unchanged (corpus rows =
// A, queries = B: a lane owns a query column), the K-step is the shared one (lvs_kstep.h).
//
// Epilogue per (corpus tile, query tile): every lane folds its 64 scores per query into a running (best, second, third) of
// u = c s - |y|^2 with the row's position in the low six mantissa bits (three VALU per score: max, med3, med3), the four
// lanes holding a query's partial triples meet in LDS, and (best key, second key, third score) go out per corpus tile.
// With the THIRD score the caller can certify "only the best two can win" for a query whose best-second margin is inside the
// one-pass error bound (lvs_nearest3_select) and settle it with two exact dot products (lvs_resolve_pairs) instead of a
// search over every row.
#include <string.h>

#include "lvs_common.h"
#include "lvs_kstep.h"
#include "lvs_tile.h"

namespace {

using lvs_kstep::BC;
using lvs_kstep::BK;
using lvs_kstep::ROWB;
using lvs_kstep::glds16;

constexpr int AQ = 256;                        // queries per tile
constexpr int A_STAGE = (BC + AQ) * ROWB;      // 64 KB
constexpr int A_OFF_BN = 2 * A_STAGE;          // float [BC]: |y|^2 of the workgroup's corpus tile (sentinel past the end)
constexpr int A_OFF_PART = A_OFF_BN + BC * 4;  // float [AQ][4][3]: (best, second, third) of the four partial holders of a query
constexpr int A_LDS = A_OFF_PART + AQ * 4 * 3 * 4;
static_assert(A_LDS <= 160 * 1024, "LDS budget");

struct AssignArgs {
    const void* xb;   // [nb][ldb] corpus rows (fp16 hi parts at columns [0, dpad))
    const void* xq;   // [nq][ldq] query rows
    const float* bn;  // [nb] |y|^2 (L2)
    const float* qn;  // [nq] |q|^2 (L2)
    u64* out1;        // [nct][nq] best key of every (corpus tile, query)
    u64* out2;        // [nct][nq] second-best key (0: none)
    float* out3;      // [nct][nq] third-best score, "larger = better" domain (-inf: none)
    long long nb, nq, ldb, ldq, id_offset;
    int nkd, metric;
    int nct, nqt, tiles_per_slab, nslab, gq;
};

__device__ inline float med3f(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
// scores are finite or -inf, never NaN: the plain instruction, without the `v_max_f32 x, x` hipcc puts in front of fmaxf to
// quiet a signalling NaN (the tagged values come out of integer operations, so it cannot prove there is none)
__device__ inline float max_nc(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// insert v into the running top three (b >= s >= t): three VALU operations
__device__ inline void top3_insert(float v, float& b, float& s, float& t) {
    t = med3f(v, s, t);
    s = med3f(v, b, s);
    b = max_nc(b, v);
}

template <bool SKEW>
__global__ __launch_bounds__(512, 2) void lvs_assign_kernel(const AssignArgs a) {
    constexpr int MI = 4, QG = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / 4, wn = wave % 4;

    // block -> (corpus tile, query slab): the deal of the list kernel with the corpus tiles in the role of its query tiles
    int ct, slab;
    {
        const LvsTileGroups gr = lvs_tile_groups(a.nct, a.nslab, a.gq, 0);
        int g, r;
        if (!lvs_tile_block_slot(gr, blockIdx.x, g, r)) return;
        if (!lvs_tile_group_slot(a.nct, a.nslab, a.gq, 0, gr, g, r, ct, slab)) return;
    }
    const int tile0 = slab * a.tiles_per_slab;
    const int tile1 = min(a.nqt, tile0 + a.tiles_per_slab);
    if (tile0 >= tile1) return;
    const long long c0 = (long long)ct * BC;

    float* bnl = (float*)(smem + A_OFF_BN);
    float* part = (float*)(smem + A_OFF_PART);
    const _Float16* __restrict__ xb = (const _Float16*)a.xb;
    const _Float16* __restrict__ xq = (const _Float16*)a.xq;
    const long long ldb = a.ldb, ldq = a.ldq;
    const int nkd = a.nkd;
    const bool l2 = a.metric == LVS_METRIC_L2;

    // The accumulators START at -|y|^2 / 2 (0 under inner product), so the matrix cores deliver v = q.y - |y|^2 / 2 - the order
    // value u / 2 of squared L2 - and the epilogue needs no multiply-add per score (four VALU operations per score instead of
    // five; the epilogue's VALU work, not MFMA, is what separates this kernel from the list kernel's rate).  Rows past the end
    // start at a finite sentinel: they never win, and the position tags stay meaningful.
    if (tid < BC) {
        const long long row = c0 + tid;
        bnl[tid] = row < a.nb ? (l2 ? -0.5f * a.bn[row] : 0.f) : -1.5e38f;  // the accumulators' starting values, as stored
    }
    __syncthreads();

    // staging: wave stages corpus rows [wave*32, +32) and query rows [wave*32, +32), 8 rows per load
    const int srow = lane >> 3, sp = lane & 7;
    unsigned c_loff[4], q_loff[QG];
    int s_col[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + srow;
        s_col[i] = (sp ^ ((row >> 1) & 7)) * 8;
        long long grow = c0 + row;
        if (grow > a.nb - 1) grow = a.nb - 1;
        c_loff[i] = (unsigned)(((grow - c0) * ldb + s_col[i]) * 2);
    }
    auto set_q_offsets = [&](int tile) {  // rows past the last query re-read the last valid row
        const long long q0 = (long long)tile * AQ;
#pragma unroll
        for (int i = 0; i < QG; ++i) {
            long long grow = q0 + wave * 32 + i * 8 + srow;
            if (grow > a.nq - 1) grow = a.nq - 1;
            q_loff[i] = (unsigned)(((grow - q0) * ldq + s_col[i]) * 2);
        }
    };
    set_q_offsets(tile0);
    const char* c_tile = (const char*)xb + c0 * ldb * 2;
    {  // prologue: K-step 0 of the first tile into buffer 0
        const char* qb = (const char*)xq + (long long)tile0 * AQ * ldq * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(c_tile + c_loff[i], smem + (wave * 32 + i * 8) * ROWB);
#pragma unroll
        for (int i = 0; i < QG; ++i) glds16(qb + q_loff[i], smem + BC * ROWB + (wave * 32 + i * 8) * ROWB);
    }

    int foff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        foff[kk] = (lane & 31) * ROWB + ((((kk * 2) + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    const int a_base = (wm * MI * 32) * ROWB;
    const int b_base = BC * ROWB + (wn * 64) * ROWB;

    const int lrow_base = wm * (MI * 32) + 4 * (lane >> 5);
    f32x16 acc[MI][2];
    // both query blocks start from the same per-row values: two LDS reads straight into the accumulator registers (asm: the
    // compiler would read once and copy with 128 VALU moves - VALU issue slots are what this kernel is short of)
    const unsigned bnl_addr = (unsigned)(unsigned long long)(bnl + lrow_base);
    auto init_acc = [&]() {
        lvs_kstep::static_for<MI * 4>([&](auto ic) {
            constexpr int mi = decltype(ic)::value / 4, r4 = decltype(ic)::value % 4;
            f32x4 h0, h1;
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%3"
                         : "=&v"(h0), "=&v"(h1)
                         : "v"(bnl_addr), "n"((mi * 32 + 8 * r4) * 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[mi][0][r4 * 4 + e] = h0[e];
                acc[mi][1][r4 * 4 + e] = h1[e];
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    init_acc();

    const int T = (tile1 - tile0) * nkd;
    int n_tile = 0, n_r = 0;  // (query tile, k-block) of the K-step being prefetched
    int ks = 0, ti = 0;
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    const int holder = wm * 2 + (lane >> 5);

    // tile epilogue, first half (every wave): fold the 128 scores a lane holds into (best, second, third) per query block -
    // four VALU operations per score -, start the accumulators afresh, leave the partial triples in LDS
    auto fold_and_publish = [&]() {
        float bu[2] = {-INFINITY, -INFINITY}, su[2] = {-INFINITY, -INFINITY}, tu[2] = {-INFINITY, -INFINITY};
        lvs_kstep::static_for<MI>([&](auto mic) {
            constexpr int mi = decltype(mic)::value;
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const float up = __uint_as_float((__float_as_uint(acc[mi][ni][r]) & 0xFFFFFFC0u) | (uint32_t)(mi * 16 + r));
                    top3_insert(up, bu[ni], su[ni], tu[ni]);
                }
        });
        init_acc();
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            float* p = part + ((wn * 64 + ni * 32 + (lane & 31)) * 4 + holder) * 3;
            p[0] = bu[ni];
            p[1] = su[ni];
            p[2] = tu[ni];
        }
    };
    // second half (256 threads, after a barrier): combine the four partial holders of a query, decode ids, write out
    auto decode = [&](long long q0, float qn_v) {
        if (tid < AQ && q0 + tid < a.nq) {
            const float* p = part + tid * 12;
            float B = -INFINITY, S = -INFINITY, Tt = -INFINITY;
            u64 k1 = 0, k2 = 0;
            // v = q.y - |y|^2 / 2 (tagged): squared L2 = |q|^2 - 2 v; the doubling is exact and leaves the tag bits alone
            auto score_of = [&](float v) { return v > -1.0e38f ? (l2 ? -fmaxf(qn_v - 2.0f * v, 0.f) : v) : -INFINITY; };
#pragma unroll
            for (int h = 0; h < 4; ++h) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float u = p[h * 3 + j];
                    top3_insert(u, B, S, Tt);
                    if (j < 2 && u > -1.0e38f) {  // a candidate with an id: its row from (holder, position tag)
                        const uint32_t tag = __float_as_uint(u) & 63u, r = tag & 15u;
                        const uint32_t row = (uint32_t)((h >> 1) * (MI * 32) + 4 * (h & 1)) + (tag >> 4) * 32u + (r & 3u) + 8u * (r >> 2);
                        const u64 key = lvs_pack_key(score_of(u), (uint32_t)(c0 + row + a.id_offset));
                        if (key > k1) {
                            k2 = k1;
                            k1 = key;
                        } else if (key > k2) {
                            k2 = key;
                        }
                    }
                }
            }
            const long long o = (long long)ct * a.nq + q0 + tid;
            a.out1[o] = k1;
            a.out2[o] = k2;
            a.out3[o] = score_of(Tt);
        }
    };
    // SKEW (tiles of three or more K-steps): the two waves of a SIMD fold at DIFFERENT moments, so that one wave's VALU work
    // runs beside the other's MFMAs instead of both queueing at the one VALU while the matrix pipe idles.  Waves 4-7 (the
    // half with issue priority) fold right after the tile's last K-step, as before; waves 0-3 pass the next K-step's barrier
    // first and fold at the START of that K-step, while waves 4-7 already issue its MFMAs.  The partial triples of a tile are
    // therefore complete one barrier later, and the decode runs after the barrier that follows (two K-step barriers after
    // the tile's end; the next tile's first triple is written no earlier than the barrier after that).  Same scores, same
    // folds, same decode: bit-identical output.
    const bool late = SKEW && wave < 4;
    bool fold_due = false;
    int dec_wait = 0;
    long long dec_q0 = 0;
    float dec_qn = 0.f;
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (SKEW && dec_wait && --dec_wait == 0) decode(dec_q0, dec_qn);
        const char* sb = smem + buf * A_STAGE;
        char* n_base = smem + (buf ^ 1) * A_STAGE;
        if (t + 1 < T) {
            if (++n_r == nkd) {
                n_r = 0;
                ++n_tile;
                set_q_offsets(tile0 + n_tile);
            }
        }
        const char* c_sbase = c_tile + (long long)n_r * BK * 2;
        const char* q_sbase = (const char*)xq + ((long long)(tile0 + n_tile) * AQ * ldq + (long long)n_r * BK) * 2;
        if (late && fold_due) {
            fold_and_publish();
            fold_due = false;
        }
        lvs_kstep::run<MI, QG>(sb, n_base, c_sbase, q_sbase, c_loff, q_loff, wave, a_base, b_base, foff, acc);
        if (++ks < nkd) continue;
        ks = 0;
        // ======================== tile epilogue: (corpus tile ct) x (query tile tile0 + ti) ========================
        const long long q0 = (long long)(tile0 + ti) * AQ;
        ++ti;
        float qn_v = 0.f;
        if (tid < AQ && l2) qn_v = a.qn[q0 + tid < a.nq ? q0 + tid : a.nq - 1];  // used after the barrier(s) below
        if (!SKEW) {
            fold_and_publish();
            __syncthreads();
            // the next write of `part` is at least one K-step barrier away: no second barrier needed
            decode(q0, qn_v);
        } else {
            if (late) fold_due = true;
            else fold_and_publish();
            dec_wait = 2;
            dec_q0 = q0;
            dec_qn = qn_v;
        }
    }
    if (SKEW) {  // the last tile of the slab
        if (late && fold_due) fold_and_publish();
        __syncthreads();
        if (dec_wait) decode(dec_q0, dec_qn);
    }
}

// per query: combine the (best key, second key, third score) triples of the nparts corpus tiles
__global__ __launch_bounds__(256) void assign_merge_kernel(const u64* __restrict__ p1, const u64* __restrict__ p2,
                                                           const float* __restrict__ p3, int nparts, long long nq,
                                                           u64* __restrict__ keys1, u64* __restrict__ keys2,
                                                           float* __restrict__ second, float* __restrict__ third) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    u64 k1 = 0, k2 = 0;
    float B = -INFINITY, S = -INFINITY, T = -INFINITY;
    for (int p = 0; p < nparts; ++p) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const u64 key = (j ? p2 : p1)[(long long)p * nq + q];
            if (!key) continue;
            top3_insert(lvs_unord32((uint32_t)(key >> 32)), B, S, T);
            const u64 lo = key > k1 ? k1 : key;
            k1 = key > k1 ? key : k1;
            k2 = lo > k2 ? lo : k2;
        }
        const float t3 = p3[(long long)p * nq + q];
        if (t3 > -INFINITY) top3_insert(t3, B, S, T);
    }
    keys1[q] = k1;
    keys2[q] = k2;
    second[q] = k2 ? lvs_unord32((uint32_t)(k2 >> 32)) : -INFINITY;
    third[q] = T;
}

struct Coef5 {
    float c[5];
};
// Three-way certificate of a one-pass nearest-row search.  bound(q) = (c0 E + c1 R) |q| + c2 + c3 R + c4 R^2 (R, E: largest
// row norm / lo-part norm of the corpus, from `stats`), as lvs_margin_select_stats:
//   best - second > bound            certified: the one-pass winner is the exact winner;
//   else best - third > bound        only the best two can win: query index appended to pair_idx;
//   else                             open: appended to open_idx (exact search over every row).
// counts[0] / counts[1] (device uint64, zeroed by the caller) += pairs / open queries.
__global__ __launch_bounds__(256) void nearest3_select_kernel(const u64* __restrict__ keys1, const u64* __restrict__ keys2,
                                                              const float* __restrict__ second, const float* __restrict__ third,
                                                              const float* __restrict__ qn, long long nq,
                                                              const float* __restrict__ stats, Coef5 coef, long long per_block,
                                                              long long* __restrict__ pair_idx, long long* __restrict__ open_idx,
                                                              unsigned long long* __restrict__ counts) {
    const float R = sqrtf(stats[0]), E = sqrtf(stats[1]);
    const float scale = coef.c[0] * E + coef.c[1] * R;
    const float slack = coef.c[2] + coef.c[3] * R + coef.c[4] * R * R;
    __shared__ unsigned long long s_base[2];
    __shared__ unsigned s_count[2];
    const long long q_begin = (long long)blockIdx.x * per_block;
    const long long q_end = q_begin + per_block < nq ? q_begin + per_block : nq;
    auto classify = [&](long long q) -> int {  // 0 certified, 1 pair, 2 open
        const u64 kq = keys1[q];
        if (!kq) return 0;
        const float best = lvs_unord32((uint32_t)(kq >> 32));
        const float bound = scale * sqrtf(qn ? qn[q] : 1.0f) + slack;
        if ((best - second[q]) > bound) return 0;  // false for NaN / inf - inf: never certify what cannot be compared
        if (keys2[q] && (best - third[q]) > bound) return 1;
        return 2;
    };
    if (threadIdx.x < 2) s_count[threadIdx.x] = 0;
    __syncthreads();
    unsigned mine[2] = {0, 0};
    for (long long q = q_begin + threadIdx.x; q < q_end; q += blockDim.x) {
        const int c = classify(q);
        if (c) ++mine[c - 1];
    }
    if (mine[0]) atomicAdd(&s_count[0], mine[0]);
    if (mine[1]) atomicAdd(&s_count[1], mine[1]);
    __syncthreads();
    if (s_count[0] == 0 && s_count[1] == 0) return;
    if (threadIdx.x < 2) {
        s_base[threadIdx.x] = s_count[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], (unsigned long long)s_count[threadIdx.x]) : 0ull;
        s_count[threadIdx.x] = 0;
    }
    __syncthreads();
    for (long long q = q_begin + threadIdx.x; q < q_end; q += blockDim.x) {
        const int c = classify(q);
        if (c == 1) pair_idx[s_base[0] + atomicAdd(&s_count[0], 1u)] = q;
        if (c == 2) open_idx[s_base[1] + atomicAdd(&s_count[1], 1u)] = q;
    }
}

typedef _Float16 as_half8 __attribute__((ext_vector_type(8)));
__device__ inline float as_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// one wave per listed query: exact (hi + lo of both operands, float32) scores of the query against the rows named by
// keys1[q] and keys2[q]; the better key (score, then lower id) replaces keys1[q].  Same arithmetic as rescore_keys_kernel.
template <int QSPLIT, int BSPLIT>
__global__ __launch_bounds__(256) void resolve_pairs_kernel(const _Float16* __restrict__ xb, long long ldb,
                                                            const _Float16* __restrict__ xq, long long ldq, int dpad, int metric,
                                                            const float* __restrict__ bn, const float* __restrict__ qn,
                                                            long long id_offset, const long long* __restrict__ idx,
                                                            const unsigned long long* __restrict__ count, long long cap,
                                                            u64* __restrict__ keys1, const u64* __restrict__ keys2) {
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n = (long long)*count < cap ? (long long)*count : cap;
    if (i >= n) return;
    const long long q = idx[i];
    const u64 ka = keys1[q], kb = keys2[q];
    if (!ka || !kb) return;
    const uint32_t ida = 0xFFFFFFFFu - (uint32_t)(ka & 0xFFFFFFFFull), idb = 0xFFFFFFFFu - (uint32_t)(kb & 0xFFFFFFFFull);
    const long long ra = (long long)ida - id_offset, rb = (long long)idb - id_offset;
    const _Float16* qr = xq + q * ldq;
    const _Float16* ar = xb + ra * ldb;
    const _Float16* br = xb + rb * ldb;
    float sa = 0.f, sb = 0.f;
    for (int j = lane * 8; j < dpad; j += 512) {
        const as_half8 qh = *(const as_half8*)(qr + j), ah = *(const as_half8*)(ar + j), bh = *(const as_half8*)(br + j);
        as_half8 ql, al, bl;
        if (QSPLIT) ql = *(const as_half8*)(qr + dpad + j);
        if (BSPLIT) {
            al = *(const as_half8*)(ar + dpad + j);
            bl = *(const as_half8*)(br + dpad + j);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float x = (float)qh[t] + (QSPLIT ? (float)ql[t] : 0.f);
            sa += x * ((float)ah[t] + (BSPLIT ? (float)al[t] : 0.f));
            sb += x * ((float)bh[t] + (BSPLIT ? (float)bl[t] : 0.f));
        }
    }
    sa = as_wave_sum(sa);
    sb = as_wave_sum(sb);
    if (lane == 0) {
        float fa = sa, fb = sb;
        if (metric == LVS_METRIC_L2) {
            fa = -fmaxf((qn[q] + bn[ra]) - 2.0f * sa, 0.f);
            fb = -fmaxf((qn[q] + bn[rb]) - 2.0f * sb, 0.f);
        }
        const u64 na = lvs_pack_key(fa, ida), nb_ = lvs_pack_key(fb, idb);
        keys1[q] = na > nb_ ? na : nb_;
    }
}

struct AssignPlan {
    int dpad, nkd, nct, nqt, tps, nslab, gq;
    long long ldb, ldq;
    int64_t off1, off2, off3, total;
};
bool assign_plan(int64_t nq, int64_t nb, int32_t d, int32_t xb_pack, int32_t xq_pack, AssignPlan& p) {
    if (nq < 0 || nb <= 0 || d <= 0) return false;
    if ((xb_pack != LVS_PACK_F16 && xb_pack != LVS_PACK_SPLIT) || (xq_pack != LVS_PACK_F16 && xq_pack != LVS_PACK_SPLIT)) return false;
    if (nb > LVS_NEAREST3_MAX_ROWS) return false;
    p.dpad = (int)lvs_round_up(d, LVS_BK);
    p.nkd = p.dpad / LVS_BK;
    p.ldb = xb_pack == LVS_PACK_SPLIT ? 2 * p.dpad : p.dpad;
    p.ldq = xq_pack == LVS_PACK_SPLIT ? 2 * p.dpad : p.dpad;
    p.nct = (int)lvs_ceil_div(nb, LVS_BC);
    p.nqt = (int)lvs_ceil_div(nq > 0 ? nq : 1, AQ);
    // ~4096 items (16 rounds of 256 workgroups), an item at most 16 query tiles long
    long long tps = (long long)p.nqt * p.nct / lvs_tune("LVS_ASSIGN_ITEMS", 4096);
    const long long tps_max = lvs_tune("LVS_ASSIGN_TPS", 16);
    tps = tps < 1 ? 1 : (tps > tps_max ? tps_max : tps);
    p.tps = (int)tps;
    p.nslab = (int)lvs_ceil_div(p.nqt, p.tps);
    p.gq = 1;
    while (p.gq * 2 <= p.nct && p.gq * 2 <= 32) p.gq *= 2;
    if (lvs_tune_set("LVS_ASSIGN_GQ")) {
        const int v = (int)lvs_tune("LVS_ASSIGN_GQ", 0);
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32) p.gq = v;
    }
    int64_t off = 0;
    p.off1 = off;
    off += lvs_round_up((int64_t)p.nct * nq * 8, 256);
    p.off2 = off;
    off += lvs_round_up((int64_t)p.nct * nq * 8, 256);
    p.off3 = off;
    off += lvs_round_up((int64_t)p.nct * nq * 4, 256);
    p.total = off + 256;
    return true;
}

}  // namespace

extern "C" int64_t lvs_nearest3_workspace_bytes(int64_t nq, int64_t nb, int32_t d) {
    AssignPlan p;
    if (!assign_plan(nq, nb, d, LVS_PACK_F16, LVS_PACK_F16, p)) return LVS_EINVAL;
    return p.total;
}

extern "C" int32_t lvs_nearest3(const void* xb, int32_t xb_pack, int64_t nb, const void* xq, int32_t xq_pack, int64_t nq,
                                int32_t d, int32_t metric, const float* xb_norms_sq, const float* xq_norms_sq,
                                int64_t id_offset, uint64_t* out_keys, uint64_t* out_keys2, float* out_second,
                                float* out_third, void* workspace, int64_t workspace_bytes, void* stream) {
    AssignPlan p;
    LVS_REQUIRE(assign_plan(nq, nb, d, xb_pack, xq_pack, p), "bad shape nq=%lld nb=%lld d=%d pack=%d/%d (at most %d corpus rows)",
                (long long)nq, (long long)nb, d, xb_pack, xq_pack, LVS_NEAREST3_MAX_ROWS);
    LVS_REQUIRE(metric == LVS_METRIC_IP || metric == LVS_METRIC_L2, "bad metric %d", metric);
    LVS_REQUIRE(id_offset >= 0 && id_offset + nb < 0xFFFFFFFFll, "ids must stay below 2^32-1");
    if (nq == 0) return LVS_OK;
    LVS_REQUIRE(out_keys && out_keys2 && out_second && out_third, "NULL output");
    LVS_REQUIRE(xb && xq, "NULL rows");
    LVS_REQUIRE(metric != LVS_METRIC_L2 || (xb_norms_sq && xq_norms_sq), "L2 needs both norm vectors");
    if (!workspace || workspace_bytes < p.total) {
        lvs_set_error("workspace too small: need %lld bytes, got %lld", (long long)p.total, (long long)workspace_bytes);
        return LVS_ENOMEM;
    }
    LVS_DEVICE_GUARD(stream);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    AssignArgs a;
    memset(&a, 0, sizeof(a));
    a.xb = xb;
    a.xq = xq;
    a.bn = xb_norms_sq;
    a.qn = xq_norms_sq;
    a.out1 = (u64*)(ws + p.off1);
    a.out2 = (u64*)(ws + p.off2);
    a.out3 = (float*)(ws + p.off3);
    a.nb = nb;
    a.nq = nq;
    a.ldb = p.ldb;  // SPLIT rows keep their leading dimension; only the fp16 "hi" half (columns [0, dpad)) is read
    a.ldq = p.ldq;
    a.id_offset = id_offset;
    a.nkd = p.nkd;
    a.metric = metric;
    a.nct = p.nct;
    a.nqt = p.nqt;
    a.tiles_per_slab = p.tps;
    a.nslab = p.nslab;
    a.gq = p.gq;
    static LvsPerDeviceOnce attr;
    int dev = 0;
    LVS_HIP_CHECK(hipGetDevice(&dev));
    if (!attr.done(dev, (size_t)A_LDS)) {
        LVS_HIP_CHECK(hipFuncSetAttribute((const void*)lvs_assign_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS));
        LVS_HIP_CHECK(hipFuncSetAttribute((const void*)lvs_assign_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS));
        attr.set(dev, (size_t)A_LDS);
    }
    {
        LvsKernelTimer timer(st);
        const dim3 grid((unsigned)lvs_tile_grid_blocks(p.nct, p.nslab, p.gq, 0));
        // skewed folds need two K-step barriers between a tile's end and the next tile's first fold: tiles of >= 3 K-steps
        if (p.nkd >= 3 && lvs_tune("LVS_ASSIGN_SKEW", 1) != 0) hipLaunchKernelGGL(lvs_assign_kernel<true>, grid, dim3(512), A_LDS, st, a);
        else hipLaunchKernelGGL(lvs_assign_kernel<false>, grid, dim3(512), A_LDS, st, a);
        LVS_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(assign_merge_kernel, dim3((unsigned)lvs_ceil_div(nq, 256)), dim3(256), 0, st, (const u64*)a.out1,
                       (const u64*)a.out2, (const float*)a.out3, p.nct, (long long)nq, (u64*)out_keys, (u64*)out_keys2, out_second,
                       out_third);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_nearest3_select(const uint64_t* keys, const uint64_t* keys2, const float* second, const float* third,
                                       const float* q_norms_sq, int64_t nq, const float* corpus_stats, const float* coef5,
                                       int64_t* out_pair_idx, int64_t* out_open_idx, uint64_t* out_counts, void* stream) {
    LVS_REQUIRE(nq >= 0 && corpus_stats && coef5, "bad arguments");
    if (nq == 0) return LVS_OK;
    LVS_REQUIRE(keys && keys2 && second && third && out_pair_idx && out_open_idx && out_counts, "NULL buffer");
    Coef5 c;
    for (int i = 0; i < 5; ++i) {
        LVS_REQUIRE(coef5[i] >= 0.f, "negative coefficient");
        c.c[i] = coef5[i];
    }
    LVS_DEVICE_GUARD(stream);
    long long per_block = lvs_ceil_div(nq, 2048);
    per_block = lvs_round_up(per_block < 1024 ? 1024 : per_block, 256);
    hipLaunchKernelGGL(nearest3_select_kernel, dim3((unsigned)lvs_ceil_div(nq, per_block)), dim3(256), 0, (hipStream_t)stream,
                       (const u64*)keys, (const u64*)keys2, second, third, q_norms_sq, (long long)nq, corpus_stats, c, per_block,
                       (long long*)out_pair_idx, (long long*)out_open_idx, (unsigned long long*)out_counts);
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

extern "C" int32_t lvs_resolve_pairs(const void* xb, int32_t xb_pack, const void* xq, int32_t xq_pack, int32_t d, int32_t metric,
                                     const float* xb_norms_sq, const float* xq_norms_sq, int64_t id_offset,
                                     const int64_t* pair_idx, const uint64_t* pair_count, int64_t capacity, uint64_t* keys,
                                     const uint64_t* keys2, void* stream) {
    LVS_REQUIRE(d > 0 && capacity >= 0, "bad shape");
    LVS_REQUIRE(metric == LVS_METRIC_IP || metric == LVS_METRIC_L2, "bad metric %d", metric);
    LVS_REQUIRE((xb_pack == LVS_PACK_F16 || xb_pack == LVS_PACK_SPLIT) && (xq_pack == LVS_PACK_F16 || xq_pack == LVS_PACK_SPLIT),
                "bad pack mode %d/%d", xb_pack, xq_pack);
    if (capacity == 0) return LVS_OK;
    LVS_REQUIRE(xb && xq && pair_idx && pair_count && keys && keys2, "NULL buffer");
    LVS_REQUIRE(metric != LVS_METRIC_L2 || (xb_norms_sq && xq_norms_sq), "L2 needs both norm vectors");
    LVS_DEVICE_GUARD(stream);
    const int dpad = (int)lvs_round_up(d, LVS_BK);
    const long long ldb = xb_pack == LVS_PACK_SPLIT ? 2 * dpad : dpad, ldq = xq_pack == LVS_PACK_SPLIT ? 2 * dpad : dpad;
    const dim3 grid((unsigned)lvs_ceil_div(capacity, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define LVS_RESOLVE(QS, BS)                                                                                                  \
    hipLaunchKernelGGL((resolve_pairs_kernel<QS, BS>), grid, block, 0, st, (const _Float16*)xb, ldb, (const _Float16*)xq, ldq,  \
                       dpad, metric, xb_norms_sq, xq_norms_sq, (long long)id_offset, (const long long*)pair_idx,               \
                       (const unsigned long long*)pair_count, (long long)capacity, (u64*)keys, (const u64*)keys2)
    if (xq_pack == LVS_PACK_SPLIT && xb_pack == LVS_PACK_SPLIT) LVS_RESOLVE(1, 1);
    else if (xq_pack == LVS_PACK_SPLIT) LVS_RESOLVE(1, 0);
    else if (xb_pack == LVS_PACK_SPLIT) LVS_RESOLVE(0, 1);
    else LVS_RESOLVE(0, 0);
#undef LVS_RESOLVE
    LVS_HIP_CHECK(hipGetLastError());
    return LVS_OK;
}

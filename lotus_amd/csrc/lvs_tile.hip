// Tiled brute-force distance kernel with fused per-query top-k for gfx950 (MI355X, wave64, MFMA).
//
// Replaces faiss IndexFlat::search as reached from lotus/vector_store/faiss_vs.py:67,75 (and the k = 1 L2
// search of lotus/utils.py:62,65).  Not a port: faiss computes sgemm blocks and heap-inserts every score on the
// CPU; here one workgroup owns a 128-query tile, walks a slab of 256-row corpus tiles, accumulates the
// 256 x 128 score tile over the whole embedding dimension on the matrix cores, and filters it against
// per-query running thresholds so that almost no score ever leaves the registers.
//
// Geometry (one workgroup = 8 waves = 512 threads, 1 workgroup per CU):
//   score tile   256 corpus rows (MFMA M) x 128 queries (MFMA N); K-step 64 halfs
//   wave layout  4 (corpus) x 2 (queries); each wave 64 x 64 = 2 x 2 v_mfma_f32_32x32x16_f16 accumulators
//   operands     corpus = A, queries = B  ->  D[i = corpus row][j = query]; lane l holds query j = l & 31 of a
//                32-wide block, so a query's running threshold is a per-lane register ("swapped" product)
//   LDS          2 x (256 + 128) rows x 128 B staging (global_load_lds, 16 B per lane, XOR-swizzled chunks)
//                + 128 queries x 32 candidate slots x 8 B + thresholds/counters   = 132.6 KB
//
// fp32 embeddings (LVS_PACK_SPLIT): every value is carried as an fp16 pair hi + lo and the product is
// hi*hi + hi*lo + lo*hi: the same kernel runs three K segments (3 x the MFMA work, ~2^-21 relative error).
#include "lvs_common.h"
#include "lvs_tile.h"

namespace {

constexpr int ROWB = LVS_BK * 2;                          // bytes per staged row (128)
constexpr int STAGE_BYTES = (LVS_BC + LVS_BQ) * ROWB;     // one staging buffer (49152)
constexpr int OFF_LIST = 2 * STAGE_BYTES;                 // candidate lists u64 [BQ][LCAP]
constexpr int OFF_TAU = OFF_LIST + LVS_BQ * LVS_LCAP * 8; // u64 [BQ] k-th best key at last compaction
constexpr int OFF_CNT = OFF_TAU + LVS_BQ * 8;             // u32 [BQ] list fill
constexpr int OFF_FLAG = OFF_CNT + LVS_BQ * 4;            // u32 [2] overflow flags (ping-pong) + pad
constexpr int LDS_TOTAL = OFF_FLAG + 16;
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
static_assert(LDS_TOTAL == LVS_TILE_LDS_BYTES, "keep lvs_tile.h in sync");

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ inline void glds16(const void* gsrc, void* ldst) {
    // 16 B per lane, LDS destination = wave-uniform base + lane * 16
    __builtin_amdgcn_global_load_lds((gbl_void_t*)gsrc, (lds_void_t*)ldst, 16, 0, 0);
}

__device__ inline float max16(const f32x16& v) {
    float a = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    float b = fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7]));
    float c = fmaxf(fmaxf(v[8], v[9]), fmaxf(v[10], v[11]));
    float d = fmaxf(fmaxf(v[12], v[13]), fmaxf(v[14], v[15]));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

__device__ inline float tau_float(uint32_t ord) { return ord == 0 ? -INFINITY : lvs_unord32(ord); }

// blockIdx -> (query tile, slab).  Blocks are dealt round-robin to the 8 XCDs (block b runs on XCD b % 8,
// observed; used for speed only).  Each XCD walks "groups" of 8 query tiles x 4 slabs = 32 blocks, i.e. what is
// resident on its 32 CUs at a time shares 8 query tiles (1.5 MB, stays in the 4 MB L2) and 4 corpus streams.
__device__ inline bool item_of_block(const LvsTileArgs& a, int b, int& qt, int& slab) {
    int x = b & 7, j = b >> 3;
    int gseq = j >> 5, r = j & 31;
    int g = gseq * 8 + x;
    int gq = a.gq, gs = 32 / a.gq;  // group = gq query tiles x gs slabs = 32 blocks
    int nqg = (a.nqt + gq - 1) / gq;
    int qgroup = g % nqg, sgroup = g / nqg;
    qt = qgroup * gq + (r % gq);
    slab = sgroup * gs + (r / gq);
    return qt < a.nqt && slab < a.nslab;
}

}  // namespace

template <int MODE>
__global__ __launch_bounds__(LVS_TILE_THREADS, 2) void lvs_tile_kernel(const LvsTileArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int qt, slab;
    if (!item_of_block(a, blockIdx.x, qt, slab)) return;
    const long long q0 = (long long)qt * LVS_BQ;
    int tile0 = slab * a.tiles_per_slab;
    const int tile1 = min(a.ntiles, tile0 + a.tiles_per_slab);
    if (MODE == LVS_MODE_RANGE) {
        if (a.qt_stride > 1 && (qt % a.qt_stride) != a.qt_phase) return;
        if (a.q_row0 >= 0) {  // self-join: tiles whose rows are all <= every query row of this tile hold no j > i
            const long long first = (a.q_row0 + q0 - a.id_offset) / LVS_BC;
            if (first > tile0) tile0 = (int)(first < tile1 ? first : tile1);
        }
    }
    if (tile0 >= tile1) return;

    u64* lists = (u64*)(smem + OFF_LIST);
    u64* taus = (u64*)(smem + OFF_TAU);
    uint32_t* cnts = (uint32_t*)(smem + OFF_CNT);
    uint32_t* flags = (uint32_t*)(smem + OFF_FLAG);

    const _Float16* __restrict__ xb = (const _Float16*)a.xb;
    const _Float16* __restrict__ xq = (const _Float16*)a.xq;
    const long long ldb = a.ldb, ldq = a.ldq;
    const int nk = a.nk, nkd = a.nkd;

    // ---- per-lane staging addresses --------------------------------------------------------------------
    // one glds covers 8 rows x 128 B: lane -> (row = R0 + lane/8, physical chunk p = lane%8) holds logical
    // chunk c = p ^ ((row >> 1) & 7) of that row (source-side swizzle; LDS image stays lane-linear).
    const int srow = lane >> 3, sp = lane & 7;
    int c_row[4], c_col[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = wave * 32 + i * 8 + srow;
        c_row[i] = row;
        c_col[i] = (sp ^ ((row >> 1) & 7)) * 8;
    }
    const _Float16* q_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int row = wave * 16 + i * 8 + srow;
        long long grow = q0 + row;
        if (grow > a.nq - 1) grow = a.nq - 1;
        q_src[i] = xq + grow * ldq + (sp ^ ((row >> 1) & 7)) * 8;
    }

    auto stage = [&](int t, int buf) {
        int ti = t / nk, ks = t - ti * nk;
        int seg = ks / nkd, r = ks - seg * nkd;
        int qcol = a.seg_q[seg] + r * LVS_BK;
        int ccol = a.seg_c[seg] + r * LVS_BK;
        long long trow0 = (long long)(tile0 + ti) * LVS_BC;
        char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long grow = trow0 + c_row[i];
            if (grow > a.nb - 1) grow = a.nb - 1;
            glds16(xb + grow * ldb + ccol + c_col[i], base + (wave * 32 + i * 8) * ROWB);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(q_src[i] + qcol, base + LVS_BC * ROWB + (wave * 16 + i * 8) * ROWB);
    };

    // ---- per-lane fragment read offsets (16 B per lane, chunk XOR-swizzled: conflict-free ds_read_b128) ----
    int foff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        foff[kk] = (lane & 31) * ROWB + ((((kk * 2) + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    const int a_base = (wm * 64) * ROWB;
    const int b_base = LVS_BC * ROWB + (wn * 64) * ROWB;

    // ---- per-query state (lane owns queries qloc[0], qloc[1]; lanes l and l+32 share them) ----------------
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
            if (MODE == LVS_MODE_TOPK && a.ub) ubk[ni] = a.ub[(q0 + qloc[ni]) * a.ub_stride];
            if (a.metric == LVS_METRIC_L2) qnv[ni] = a.qn[q0 + qloc[ni]];
        }
    }

    if (MODE == LVS_MODE_TOPK) {
        for (int i = tid; i < LVS_BQ * LVS_LCAP; i += LVS_TILE_THREADS) lists[i] = 0;
        for (int i = tid; i < LVS_BQ; i += LVS_TILE_THREADS) {
            taus[i] = 0;
            cnts[i] = a.k;
        }
        if (tid < 4) flags[tid] = 0;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    int round = 0;  // overflow-round parity, persists across tiles (flag ping-pong)
    const int T = (tile1 - tile0) * nk;
    stage(0, 0);
    int ks_in_tile = 0, ti = 0;
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // buffer `buf` landed for every wave; everyone finished reading buffer buf^1
        if (t + 1 < T) stage(t + 1, buf ^ 1);

        const char* sb = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            half8 a0 = *(const half8*)(sb + a_base + foff[kk]);
            half8 a1 = *(const half8*)(sb + a_base + 32 * ROWB + foff[kk]);
            half8 b0 = *(const half8*)(sb + b_base + foff[kk]);
            half8 b1 = *(const half8*)(sb + b_base + 32 * ROWB + foff[kk]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
        }

        if (++ks_in_tile < nk) continue;
        ks_in_tile = 0;
        // =============================== tile epilogue ===================================================
        const long long trow0 = (long long)(tile0 + ti) * LVS_BC;
        ++ti;
        const int lrow_base = wm * 64 + 4 * (lane >> 5);  // + mi*32 + (r&3) + 8*(r>>2)

        if (a.metric == LVS_METRIC_L2) {
            // better = -max((|q|^2 + |y|^2) - 2<q,y>, 0): same fp32 expression as the oracle / faiss BLAS path
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
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

        if (MODE == LVS_MODE_SCORES) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if (!qvalid[ni]) continue;
                    float* orow = a.scores + (q0 + qloc[ni]) * a.ld_scores;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        long long row = trow0 + lrow_base + mi * 32 + 8 * r4;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (row + e < a.nb) orow[row + e] = acc[mi][ni][r4 * 4 + e];
                    }
                }
        } else if (MODE == LVS_MODE_RANGE) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const bool th = qvalid[ni] && (max16(acc[mi][ni]) > a.threshold);
                    if (!__any(th)) continue;
                    if (!th) continue;
                    const long long qg = q0 + qloc[ni];
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
                            a.pair_q[pos] = qg;
                            a.pair_j[pos] = jg;
                            a.pair_s[pos] = s;
                        }
                    }
                }
        } else if (MODE == LVS_MODE_TOPK) {
            // refresh the cross-workgroup threshold (any slab's k-th best is a valid lower bound for the final
            // k-th best; a stale value is only conservative)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                if (qvalid[ni]) {
                    uint32_t g = a.gtau[q0 + qloc[ni]];
                    gord[ni] = g > gord[ni] ? g : gord[ni];
                    tauf[ni] = fmaxf(tauf[ni], tau_float(gord[ni]));
                }
            u64 done = 0;
            for (;;) {
                bool anyhit = false;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) anyhit |= qvalid[ni] && (max16(acc[mi][ni]) >= tauf[ni]);
                if (__any(anyhit)) {
                    // rare path: rolled over the four 32x32 accumulators to keep register pressure low
#pragma unroll 1
                    for (int tsel = 0; tsel < 4; ++tsel) {
                        const int mi = tsel >> 1, ni = tsel & 1;
                        const f32x16 tv = tsel == 0 ? acc[0][0] : tsel == 1 ? acc[0][1] : tsel == 2 ? acc[1][0] : acc[1][1];
                        const float tf = ni ? tauf[1] : tauf[0];
                        const bool qv = ni ? qvalid[1] : qvalid[0];
                        const bool th = qv && (max16(tv) >= tf);
                        if (!__any(th)) continue;
                        if (th) {
                            const int q = ni ? qloc[1] : qloc[0];
                            const u64 ubq = ni ? ubk[1] : ubk[0];
                            const uint32_t go = ni ? gord[1] : gord[0];
                            const u64 tk = taus[q];
                            const long long rbase = trow0 + lrow_base + mi * 32;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float s = tv[r];
                                const u64 bit = 1ull << (tsel * 16 + r);
                                if (s >= tf && !(done & bit)) {
                                    const long long row = rbase + (r & 3) + 8 * (r >> 2);
                                    done |= bit;  // cleared again only when the append has to be retried
                                    if (row < a.nb) {
                                        const uint32_t id = a.row_ids ? a.row_ids[row] : (uint32_t)(row + a.id_offset);
                                        const u64 key = lvs_pack_key(s, id);
                                        if (key > tk && key < ubq && (uint32_t)(key >> 32) >= go) {
                                            const uint32_t pos = atomicAdd(&cnts[q], 1u);
                                            if (pos < LVS_LCAP) {
                                                lists[q * LVS_LCAP + pos] = key;
                                            } else {
                                                flags[round & 1] = 1u;  // retried after compaction
                                                done &= ~bit;
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                const uint32_t ovf = flags[round & 1];
                if (!ovf) break;
                if (tid == 0) flags[(round + 1) & 1] = 0;
                // compaction: each wave owns 16 of the 128 queries
                for (int i = 0; i < LVS_BQ / 8; ++i) {
                    const int q = wave * (LVS_BQ / 8) + i;
                    const uint32_t c = cnts[q];
                    if (c < LVS_LCAP) continue;
                    u64 v = lane < LVS_LCAP ? lists[q * LVS_LCAP + lane] : 0ull;
                    v = lvs_wave_sort_desc(v, lane);
                    if (lane < a.k) lists[q * LVS_LCAP + lane] = v;
                    const u64 tk = lvs_shfl_u64(v, a.k - 1);
                    if (lane == 0) {
                        taus[q] = tk;
                        cnts[q] = a.k;
                        if (tk != 0 && (q0 + q) < a.nq) atomicMax(&a.gtau[q0 + q], (uint32_t)(tk >> 32));
                    }
                }
                __syncthreads();
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    uint32_t lo = (uint32_t)(taus[qloc[ni]] >> 32);
                    tauf[ni] = fmaxf(tauf[ni], tau_float(lo));
                }
                ++round;
            }
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    }

    if (MODE == LVS_MODE_TOPK) {
        __syncthreads();
        for (int i = 0; i < LVS_BQ / 8; ++i) {
            const int q = wave * (LVS_BQ / 8) + i;
            if (q0 + q >= a.nq) continue;
            uint32_t c = cnts[q];
            c = c < LVS_LCAP ? c : LVS_LCAP;
            u64 v = lane < (int)c ? lists[q * LVS_LCAP + lane] : 0ull;
            v = lvs_wave_sort_desc(v, lane);
            if (lane < a.k) a.out[((long long)slab * a.nq + q0 + q) * a.k + lane] = v;
            const u64 tk = lvs_shfl_u64(v, a.k - 1);
            if (lane == 0 && tk != 0) atomicMax(&a.gtau[q0 + q], (uint32_t)(tk >> 32));
        }
    }
}

template __global__ void lvs_tile_kernel<LVS_MODE_TOPK>(const LvsTileArgs);
template __global__ void lvs_tile_kernel<LVS_MODE_SCORES>(const LvsTileArgs);
template __global__ void lvs_tile_kernel<LVS_MODE_RANGE>(const LvsTileArgs);

// ---- launch helpers (called from lvs_capi.hip) -------------------------------------------------------------
int lvs_tile_grid_blocks(int nqt, int nslab, int gq) {
    int gs = 32 / gq;
    long long nqg = (nqt + gq - 1) / gq, nsg = (nslab + gs - 1) / gs;
    long long groups = lvs_round_up(nqg * nsg, 8);
    return (int)(groups * 32);
}

hipError_t lvs_tile_launch(int mode, const LvsTileArgs& a, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)lvs_tile_kernel<LVS_MODE_TOPK>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)lvs_tile_kernel<LVS_MODE_SCORES>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)lvs_tile_kernel<LVS_MODE_RANGE>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    dim3 grid(lvs_tile_grid_blocks(a.nqt, a.nslab, a.gq)), block(LVS_TILE_THREADS);
    if (mode == LVS_MODE_TOPK)
        hipLaunchKernelGGL(lvs_tile_kernel<LVS_MODE_TOPK>, grid, block, LDS_TOTAL, stream, a);
    else if (mode == LVS_MODE_RANGE)
        hipLaunchKernelGGL(lvs_tile_kernel<LVS_MODE_RANGE>, grid, block, LDS_TOTAL, stream, a);
    else
        hipLaunchKernelGGL(lvs_tile_kernel<LVS_MODE_SCORES>, grid, block, LDS_TOTAL, stream, a);
    return hipGetLastError();
}

// Small-batch search kernel: up to LVS_STREAM_MAXQ (256) queries against the whole corpus shard - the HBM-bound regime of the
// path (the literal `sem_search` operator issues ONE query per call: lotus/sem_ops/sem_search.py:121-122 -> faiss_vs.py:75;
// its K-doubling loop, small sim-joins and batched searches send a few dozen to a few hundred).
//
// Roofline: HBM.  Algorithmic bytes per launch = nb * ld * 2 (every corpus byte exactly once) + O(nq * ld);
// at 1 M x 768 fp16 that is 1.536 GB -> 0.19 ms at 8 TB/s.  Nothing is staged through LDS except the queries:
//   * the queries live in LDS for the whole kernel, laid out as ready-made MFMA B fragments, one set per block of 32
//     queries ([NQB][K/16][64 lanes][16 B], lane-linear -> conflict-free ds_read_b128); NQB = 1, 2 or 3 blocks per
//     workgroup (d = 768 fp16: 48 KB per block);
//   * corpus rows stream from HBM straight into registers as A fragments (lane (r, h) reads 16 B of row r; the two
//     half-wave lanes of a row cover 32 contiguous bytes, four consecutive K-slices one 128-B line), UNROLL loads
//     (1 KB each per wave) in flight ahead of the MFMAs; every fragment feeds NQB MFMAs (one per query block);
//   * WAVES = 4 (one query block: 2-3 workgroups per CU) or 8 (two query blocks: 96 KB of fragments leave room for one
//     workgroup per CU, so the workgroup itself carries the 128 KB of loads in flight that HBM needs - round 2's 4 waves
//     x 16 KB were latency-bound);
//   * more than 96 queries: G = 2 .. 4 SIBLING workgroups (same XCD, consecutive dispatch slots) stream the SAME corpus
//     range, each with its own 64 queries in LDS.  The first sibling to touch a line pulls it from HBM, the others find it
//     in that XCD's L2 (or the memory-side cache): HBM still sees every corpus byte once, the L2 -> CU path carries it G
//     times (G x 31 GB/s per CU at 8 TB/s - within the 64 B / clock of a CU's vector memory path for G <= 4);
//   * each wave owns 32-row blocks: 32 x 32 x K product per query block on v_mfma_f32_32x32x16_f16, operands swapped as
//     in the tile kernels (corpus = A, queries = B) so that a lane owns one query column and its threshold is a register;
//   * hits go through the same wave-cooperative sorted insertion into per-query lists (LDS, a.kcap slots per query, one
//     lock per query because the waves of a workgroup share the queries).  With several queries the thresholds are SEEDED
//     (lvs_flat_search_keys): the kernel first runs in SEED mode over a sample of the rows - no lists, every workgroup keeps
//     the best score per query - and the k-th largest of those per-range maxima is the starting threshold, so that a
//     workgroup only inserts rows that beat it (~k nb / sample per query over the whole launch) instead of building its
//     lists from cold (~k (1 + ln(rows / k)) insertions per query AND workgroup - with dozens of queries that, not HBM, set
//     the time in round 2).  Exact: a threshold taken from real rows never excludes a top-k row.  Thresholds are also
//     shared across workgroups through the global per-query word; every workgroup writes its k candidates and the merge
//     kernel finishes.
#include "lvs_common.h"
#include "lvs_tile.h"

namespace {

constexpr int SQ = 32;          // queries per query block (MFMA N)
// UNROLL = A-fragment loads in flight per wave (1 KB each) = fragments per inner-loop iteration: a template parameter so
// that the branch-free fast path exists for every row length whose fragment count per K segment is a multiple of 8:
//   16 (d = 256, 512, 768, 1024, ...), 24 (d = 384 - BASELINE configs[0]'s dimension -, 1152), 8 (d = 128, 640, ...)

__device__ inline float tau_float(uint32_t ord) { return ord == 0 ? -INFINITY : lvs_unord32(ord); }

// scores are finite or -inf, never NaN: v_max3_f32 without fmaxf's canonicalisation (8 instructions for 16 values)
__device__ inline float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ inline float max16(const f32x16& v) {
    const float a = max3(v[0], v[1], v[2]), b = max3(v[3], v[4], v[5]), c = max3(v[6], v[7], v[8]);
    const float d = max3(v[9], v[10], v[11]), e = max3(v[12], v[13], v[14]);
    return max3(max3(a, b, c), max3(d, e, v[15]), v[15]);
}

}  // namespace

// SEED = true: the same scan over a SAMPLE of the rows with a trivial epilogue - every lane keeps the best score it saw, the
// workgroup writes one value per query (a.seed_out[range][q]) and no lists exist.  lvs_flat_search_keys takes the k-th
// largest of a query's values as its starting threshold (a valid lower bound of the k-th best score: each value is the
// score of a real row, and the k-th largest of a subset never exceeds the k-th largest of the whole).
template <int UNROLL, int NQB, int WAVES, bool SEED>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? 2 : 1) void lvs_stream_kernel(const LvsStreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NQ = NQB * SQ;
    const int KCAP = a.kcap;  // list slots per query (>= k, <= 64 = lanes of the cooperative insertion)
    half8* bfrag = (half8*)smem;                                              // [NQB][nbfrag][64]
    u64* lists = (u64*)(smem + (size_t)NQB * a.nbfrag * 1024);                // [NQ][KCAP]
    uint32_t* locks = (uint32_t*)((char*)lists + (size_t)NQ * KCAP * 8);      // [NQ]
    const int k = a.k;
    // blockIdx -> (corpus range, sibling group).  Blocks are dealt to the XCDs round-robin (b % 8, observed; used for speed
    // only): consecutive slots of ONE XCD are the G siblings of a range, so they share its lines through that XCD's L2.
    const int G = a.groups;
    int range = blockIdx.x, grp = 0;
    if (G > 1) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        grp = slot % G;
        range = (slot / G) * 8 + xcd;
    }
    const int qbase = grp * NQ;  // first query of this workgroup

    // ---- queries -> LDS as B fragments: block qb, fragment f, lane l = query qbase + qb*32 + (l & 31), halfs .. + (l>>5)*8
    const _Float16* xq = (const _Float16*)a.xq;
    for (int idx = tid; idx < NQB * a.nbfrag * 64; idx += WAVES * 64) {
        const int l = idx & 63, fq = idx >> 6;
        const int qb = fq / a.nbfrag, f = fq - qb * a.nbfrag;
        const int part = f / a.jper, jj = f - part * a.jper;  // part 0: columns [0, dpad), part 1: [dpad, 2 dpad)
        int qrow = qbase + qb * SQ + (l & 31);
        if (qrow > a.nq - 1) qrow = a.nq - 1;
        bfrag[idx] = *(const half8*)(xq + (long long)qrow * a.ldq + part * a.jper * 16 + jj * 16 + (l >> 5) * 8);
    }
    if (!SEED) {
        for (int i = tid; i < NQ * KCAP; i += WAVES * 64) lists[i] = 0;
        for (int i = tid; i < NQ; i += WAVES * 64) locks[i] = 0;
    }
    __syncthreads();
    float seedbest[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) seedbest[qb] = -INFINITY;

    // per query block: this lane's query, its validity, running threshold, shared threshold, |q|^2
    int qi[NQB];   // list index inside this workgroup; the query is qbase + qi
    bool qvalid[NQB];
    float tauf[NQB], qnv[NQB];
    uint32_t gord[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        qi[qb] = qb * SQ + (lane & 31);
        qvalid[qb] = qbase + qi[qb] < a.nq;
        tauf[qb] = -INFINITY;
        // start from the shared threshold: zero on a fresh call, the k-th best score of the sample on a seeded one
        gord[qb] = (!SEED && qvalid[qb]) ? a.gtau[qbase + qi[qb]] : 0u;
        tauf[qb] = tau_float(gord[qb]);
        qnv[qb] = (a.metric == LVS_METRIC_L2 && qvalid[qb]) ? a.qn[qbase + qi[qb]] : 0.f;
    }

    const _Float16* xb = (const _Float16*)a.xb;
    const long long nblocks = (a.nb + 31) / 32;
    const long long b0 = (long long)range * a.blocks_per_wg;
    const long long b1 = b0 + a.blocks_per_wg < nblocks ? b0 + a.blocks_per_wg : nblocks;
    const int nj = a.nj, jper = a.jper;
    const long long qbstride = (long long)a.nbfrag * 64;  // half8 elements between the fragment sets of two query blocks

    // ---- fragment stream: one continuous software pipeline over ALL of this wave's row blocks --------------------
    // The A-fragment loads run UNROLL fragments ahead of the MFMAs and cross block boundaries (the first fragments of
    // the next block are in flight while this block's last MFMAs and its epilogue run), so the HBM latency is paid once
    // per wave, not once per 32-row block.  A block's fragment count is padded to a multiple of UNROLL with repeats of
    // its last fragment (d = 768: 48 fragments, no padding).  All stream state below is wave-uniform except the
    // per-lane row pointers.
    const int njp = (nj + UNROLL - 1) / UNROLL * UNROLL;
    const int segc0 = a.seg_c[0], segc1 = a.seg_c[1], segc2 = a.seg_c[2];
    const int segb0 = a.seg_b[0], segb1 = a.seg_b[1], segb2 = a.seg_b[2];
    auto row_ptr = [&](long long blk) {
        long long arow = blk * 32 + (lane & 31);
        if (arow > a.nb - 1) arow = a.nb - 1;
        return xb + arow * a.ldb + (lane >> 5) * 8;  // + seg_c[seg] + jj * 16
    };
    // load stream position: block l_blk, step l_s in [0, njp), fragment (l_seg, l_jj) while l_s < nj
    long long l_blk = b0 + wave;
    const _Float16* l_ap = row_ptr(l_blk < b1 ? l_blk : b0 + wave);
    int l_s = 0, l_seg = 0, l_jj = 0, l_off = segc0;  // l_off = seg_c[l_seg] + l_jj * 16 (halfs)
    auto next_load = [&]() {
        const half8 v = *(const half8*)(l_ap + l_off);
        ++l_s;
        if (l_s < nj) {  // next real fragment of this block
            if (++l_jj == jper) {
                l_jj = 0;
                ++l_seg;
                l_off = l_seg == 1 ? segc1 : segc2;
            } else {
                l_off += 16;
            }
        } else if (l_s == njp) {  // block done: move to this wave's next block (or stay on the last one)
            l_s = 0;
            l_seg = 0;
            l_jj = 0;
            l_off = segc0;
            if (l_blk + WAVES < b1) {
                l_blk += WAVES;
                l_ap = row_ptr(l_blk);
            }
        }  // else: padding step, repeat the last fragment
        return v;
    };
    // the common shapes (fragments per K segment a multiple of UNROLL, fp16 or hi|lo rows) have a branch-free inner loop:
    // a block is 1..3 runs of jper contiguous fragments, the UNROLL loads of an iteration
    // share one base pointer (the same run UNROLL fragments on, the next run's start, or the next block's first run) and
    // differ by immediate offsets, like the B-fragment reads
    const bool simple = jper % UNROLL == 0;
    half8 abuf[UNROLL];
    if (b0 + wave < b1) {
        if (simple) {
            const _Float16* p0 = row_ptr(b0 + wave) + segc0;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) abuf[u] = *(const half8*)(p0 + u * 16);
        } else {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) abuf[u] = next_load();
        }
    }

    for (long long blk = b0 + wave; blk < b1; blk += WAVES) {
        const long long row0 = blk * 32;
        f32x16 acc[NQB];
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qb][r] = 0.f;
        if (simple) {
            const _Float16* ap_cur = row_ptr(blk);
            const _Float16* ap_nxt = row_ptr(blk + WAVES < b1 ? blk + WAVES : blk) + segc0;
            for (int run = 0; run < a.nseg; ++run) {
                const int c_off = run == 0 ? segc0 : (run == 1 ? segc1 : segc2);
                const int b_off = run == 0 ? segb0 : (run == 1 ? segb1 : segb2);
                const _Float16* run_next = run + 1 < a.nseg ? ap_cur + (run == 0 ? segc1 : segc2) : ap_nxt;
                for (int j0 = 0; j0 < jper; j0 += UNROLL) {
                    const _Float16* nsrc = j0 + UNROLL < jper ? ap_cur + c_off + (j0 + UNROLL) * 16 : run_next;
                    const half8* bsrc = bfrag + (long long)(b_off + j0) * 64 + lane;
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        const half8 av = abuf[u];
                        abuf[u] = *(const half8*)(nsrc + u * 16);
#ifdef LVS_TUNING
                        if (a.debug == 1) {  // timing ablation: no MFMA / B reads (results are wrong)
                            acc[0][0] += (float)av[0];
                            continue;
                        }
#endif
#pragma unroll
                        for (int qb = 0; qb < NQB; ++qb)
                            acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bsrc[qb * qbstride + u * 64], acc[qb], 0, 0, 0);
                    }
                }
            }
        } else {
        int c_seg = 0, c_jj = 0, c_b = segb0;  // compute stream: B fragment index c_b = seg_b[c_seg] + c_jj
        for (int j0 = 0; j0 < njp; j0 += UNROLL) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const half8 av = abuf[u];
                abuf[u] = next_load();  // the fragment UNROLL steps ahead (possibly of the next block)
                if (j0 + u < nj) {
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb) {
                        const half8 bv = bfrag[qb * qbstride + c_b * 64 + lane];
                        acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[qb], 0, 0, 0);
                    }
                    if (++c_jj == jper) {
                        c_jj = 0;
                        ++c_seg;
                        c_b = c_seg == 1 ? segb1 : segb2;
                    } else {
                        ++c_b;
                    }
                }
            }
        }
        }
        // ---- block epilogue: 32 rows x NQ queries; lane holds queries qi[*], rows row0 + (r&3) + 8*(r>>2) + 4*(lane>>5)
#ifdef LVS_TUNING
        if (a.debug == 2) continue;  // timing ablation: no block epilogue (results are wrong)
#endif
        const long long rbase = row0 + 4 * (lane >> 5);
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            const int q = qi[qb];
            if (a.metric == LVS_METRIC_L2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = rbase + (r & 3) + 8 * (r >> 2);
                    const float bnv = row < a.nb ? a.bn[row] : 0.f;
                    acc[qb][r] = -fmaxf((qnv[qb] + bnv) - 2.0f * acc[qb][r], 0.f);
                }
            }
            if constexpr (SEED) {
                seedbest[qb] = fmaxf(seedbest[qb], max16(acc[qb]));  // the sample holds whole 32-row blocks only: every row is real
                continue;
            }
            {
                const uint32_t lo = (uint32_t)(lists[q * KCAP + k - 1] >> 32);
                tauf[qb] = fmaxf(tauf[qb], tau_float(lo));
            }
            const bool th = qvalid[qb] && (max16(acc[qb]) >= tauf[qb]);
            if (__any(th)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s = acc[qb][r];
                    bool pending = false;
                    u64 key = 0;
                    if (th && s >= tauf[qb]) {
                        const long long row = rbase + (r & 3) + 8 * (r >> 2);
                        if (row < a.nb) {
                            const uint32_t id = a.row_ids ? a.row_ids[row] : (uint32_t)(row + a.id_offset);
                            key = lvs_pack_key(s, id);
                            pending = (uint32_t)(key >> 32) >= gord[qb];
                        }
                    }
                    unsigned long long pm = __ballot(pending);
                    while (pm) {  // wave-cooperative sorted insertion (see lvs_tile.hip)
                        const int src = __ffsll((long long)pm) - 1;
                        pm &= pm - 1;
                        const uint32_t klo = __builtin_amdgcn_readlane((uint32_t)key, src);
                        const uint32_t khi = __builtin_amdgcn_readlane((uint32_t)(key >> 32), src);
                        const u64 ukey = ((u64)khi << 32) | klo;
                        const int uq = __builtin_amdgcn_readlane(q, src);
                        u64* UL = lists + uq * KCAP;
                        u64 mine = 0, prev = ~0ull;
                        for (;;) {
                            uint32_t seen = 0;
                            if (lane == 0)
                                __hip_atomic_compare_exchange_strong(&locks[uq], &seen, 1u, __ATOMIC_RELAXED,
                                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            asm volatile("" ::: "memory");
                            if (lane < k) {
                                mine = UL[lane];
                                if (lane > 0) prev = UL[lane - 1];
                            }
                            if (__builtin_amdgcn_readfirstlane(seen) == 0) break;
                        }
                        u64 newv = 0;
                        if (lane < k) newv = mine > ukey ? mine : (prev > ukey ? ukey : prev);
                        __builtin_amdgcn_wave_barrier();
                        if (lane < k) UL[lane] = newv;
                        const uint32_t ntau = __builtin_amdgcn_readlane((uint32_t)(newv >> 32), k - 1);
                        asm volatile("" ::: "memory");  // slot writes stay ahead of the unlock (LDS is in-order per wave)
                        if (lane == 0) __hip_atomic_store(&locks[uq], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (q == uq) tauf[qb] = fmaxf(tauf[qb], tau_float(ntau));
                    }
                }
            }
        }
        // exchange thresholds with the other workgroups every 32 blocks of this wave (the global load drains the
        // in-flight fragment loads - vmcnt is in-order - so this must stay rare)
        const bool exch = !SEED && (((blk - b0) / WAVES) & 31) == 31;
#pragma unroll
        for (int qb = 0; qb < NQB && !SEED; ++qb) {
            if (exch && qvalid[qb] && lane < 32) {
                const uint32_t lo = (uint32_t)(lists[qi[qb] * KCAP + k - 1] >> 32);
                if (lo > gord[qb]) atomicMax(&a.gtau[qbase + qi[qb]], lo);
                const uint32_t g = a.gtau[qbase + qi[qb]];
                gord[qb] = g > gord[qb] ? g : gord[qb];
            }
            const uint32_t gl = __shfl(gord[qb], lane & 31, 64);  // lanes l and l+32 share the query
            gord[qb] = gl > gord[qb] ? gl : gord[qb];
            tauf[qb] = fmaxf(tauf[qb], tau_float(gord[qb]));
        }
    }
    if constexpr (SEED) {
        float* red = (float*)lists;  // [WAVES * 2][NQ] (the list area is unused in this mode)
        __syncthreads();
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) red[(wave * 2 + (lane >> 5)) * NQ + qi[qb]] = seedbest[qb];
        __syncthreads();
        for (int qq = tid; qq < NQ; qq += WAVES * 64) {
            float m = -INFINITY;
            for (int w = 0; w < WAVES * 2; ++w) m = fmaxf(m, red[w * NQ + qq]);
            if (qbase + qq < a.nq) a.seed_out[(long long)range * a.nq + qbase + qq] = m;
        }
        return;
    }
    __syncthreads();
    for (int i = tid; i < NQ * k; i += WAVES * 64) {
        const int qq = i / k, j = i - qq * k;
        if (qbase + qq < a.nq) a.out[((long long)range * a.nq + qbase + qq) * k + j] = lists[qq * KCAP + j];
    }
    for (int qq = tid; qq < NQ; qq += WAVES * 64) {
        const uint32_t lo = (uint32_t)(lists[qq * KCAP + k - 1] >> 32);
        if (lo && qbase + qq < a.nq) atomicMax(&a.gtau[qbase + qq], lo);
    }
}

// host side -----------------------------------------------------------------------------------------------------
// corpus ranges of a launch: one per CU when every workgroup has its own range, 256 / G with G sibling groups
int lvs_stream_ranges(int64_t nb, int groups) {
    const int64_t nblocks = (nb + 31) / 32;
    int64_t r = 256 / (groups > 0 ? groups : 1) / (groups > 1 ? 8 : 1) * (groups > 1 ? 8 : 1);  // siblings: multiples of 8
    if (groups <= 1 && lvs_tune("LVS_STREAM_WGS", 0) > 0) r = lvs_tune("LVS_STREAM_WGS", 0);  // -DLVS_TUNING builds only
    if (r > LVS_STREAM_MAXWG) r = LVS_STREAM_MAXWG;
    if (groups > 1) {
        // siblings are consecutive slots of one XCD: ranges come in multiples of 8 (one per XCD and slot group)
        int64_t cap = (nblocks + 3) / 4;
        cap = cap / 8 * 8;
        if (r > cap) r = cap;
        if (r < 8) r = 8;
    } else {
        if (r > (nblocks + 3) / 4) r = (nblocks + 3) / 4;
        if (r < 1) r = 1;
    }
    return (int)r;
}

size_t lvs_stream_lds_bytes(int nbfrag, int nqb, int kcap) {
    return (size_t)nqb * nbfrag * 1024 + (size_t)nqb * SQ * kcap * 8 + (size_t)nqb * SQ * 4;
}

// How a call is laid out: *out_nqb = 32-query blocks per workgroup (1, 2 or 3), *out_groups = sibling workgroups per corpus
// range (1 .. 4), *out_kcap = list slots per query.  Fewest groups first: every group is one more pass of the corpus
// through the fabric (measured, 1 M x 768: 64 queries in one group 0.26 ms, 128 in two 0.38 ms, 256 in four 0.70 ms - the
// siblings drift apart by more than an L2's worth of stream, so the memory-side cache, not the L2, serves the re-reads);
// then the fewest blocks per workgroup (more list slots, less LDS).  Returns 0 when the call does not fit the streaming
// kernel (too many queries / k for the LDS left beside the query fragments).
int lvs_stream_plan(int64_t nq, int k, int nbfrag, int* out_kcap, int* out_nqb, int* out_groups) {
    if (nq < 1 || nq > LVS_STREAM_MAXQ || k < 1 || k > LVS_KPASS) return 0;
    const int blocks = (int)((nq + SQ - 1) / SQ);
    for (int groups = 1; groups <= 4; ++groups) {
        const int nqb = (blocks + groups - 1) / groups;
        if (nqb > 3) continue;
        // one block keeps the full 64 slots (any k <= 56 costs the same insertion step); more blocks trade slots for queries
        const int kcap = nqb == 1 ? 64 : (k <= 16 ? 16 : (k <= 32 ? 32 : 64));
        if (lvs_stream_lds_bytes(nbfrag, nqb, kcap) > 160 * 1024) continue;  // a workgroup may use the whole 160 KiB
        if (out_kcap) *out_kcap = kcap;
        if (out_nqb) *out_nqb = nqb;
        if (out_groups) *out_groups = groups;
        return nqb;
    }
    return 0;
}

template <int U, int NQB, int WAVES, bool SEED>
static hipError_t stream_launch_one(const LvsStreamArgs& a, int grid, size_t lds, hipStream_t stream) {
    static LvsPerDeviceOnce attr;  // the attribute is a per-device property (one per instantiation)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!attr.done(dev, lds)) {
        e = hipFuncSetAttribute((const void*)lvs_stream_kernel<U, NQB, WAVES, SEED>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr.set(dev, lds);
    }
    hipLaunchKernelGGL((lvs_stream_kernel<U, NQB, WAVES, SEED>), dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}

template <int NQB, int WAVES, bool SEED>
static hipError_t stream_launch_nqb(const LvsStreamArgs& a, int grid, size_t lds, hipStream_t stream) {
    // fragments in flight: the largest of 16 / 24 / 8 that divides the fragments per K segment gives the branch-free
    // inner loop; anything else runs the general state machine with 16
    if (a.jper % 16 == 0) return stream_launch_one<16, NQB, WAVES, SEED>(a, grid, lds, stream);
    if constexpr (NQB == 1 && !SEED) {  // 24 fragment registers + 2 accumulator sets: one query block only
        if (a.jper % 24 == 0) return stream_launch_one<24, NQB, WAVES, SEED>(a, grid, lds, stream);
    }
    if (a.jper % 8 == 0) return stream_launch_one<8, NQB, WAVES, SEED>(a, grid, lds, stream);
    return stream_launch_one<16, NQB, WAVES, SEED>(a, grid, lds, stream);
}

template <bool SEED>
static hipError_t stream_launch_mode(const LvsStreamArgs& a, int grid, size_t lds, hipStream_t stream) {
    if (a.nqb == 1) return stream_launch_nqb<1, 4, SEED>(a, grid, lds, stream);
    if (a.nqb == 2) return stream_launch_nqb<2, 8, SEED>(a, grid, lds, stream);
    return stream_launch_nqb<3, 8, SEED>(a, grid, lds, stream);
}

// a.nqb / a.groups / a.kcap come from lvs_stream_plan.  On return a.nparts = candidate lists per query in a.out
// ([nparts][nq][k]).
hipError_t lvs_stream_launch(LvsStreamArgs& a, hipStream_t stream) {
    const int64_t nblocks = (a.nb + 31) / 32;
    if (a.groups < 1 || a.groups > 4) return hipErrorInvalidValue;
    if (a.nqb < 1 || a.nqb > 3 || a.kcap < a.k || a.kcap > 64) return hipErrorInvalidValue;
    if ((long long)a.nqb * SQ * a.groups < a.nq) return hipErrorInvalidValue;
    int ranges = lvs_stream_ranges(a.nb, a.groups);
    a.blocks_per_wg = (int)((nblocks + ranges - 1) / ranges);
    a.debug = (int)lvs_tune("LVS_STREAM_DEBUG", 0);
    if (a.groups > 1) {
        ranges = (int)((((nblocks + a.blocks_per_wg - 1) / a.blocks_per_wg) + 7) / 8 * 8);  // empty ranges write empty lists
    } else {
        ranges = (int)((nblocks + a.blocks_per_wg - 1) / a.blocks_per_wg);
    }
    a.nparts = ranges;
    const int grid = ranges * a.groups;
    const size_t lds = lvs_stream_lds_bytes(a.nbfrag, a.nqb, a.kcap);
    if (a.seed_out) return stream_launch_mode<true>(a, grid, lds, stream);
    return stream_launch_mode<false>(a, grid, lds, stream);
}

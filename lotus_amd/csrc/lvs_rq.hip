// Mid-size batches - 97 .. 256 queries per call (a sem_search K-doubling loop on a few hundred survivors, a sim-join of a
// small left frame: lotus/sem_ops/sem_search.py:120-138, sem_sim_join.py:132-134) - with the QUERIES RESIDENT IN REGISTERS.
//
// Below ~100 queries a call is HBM-bound and lvs_stream_kernel keeps the queries in LDS as B fragments (48 KB per 32 queries
// at d = 768: three blocks fill the 160 KB).  Beyond, rounds 3-4 fell back to the list kernel with one query tile x 245 slabs,
// which re-stages the query tile from L2 for every K-step and runs at its K-step pace: 0.39 / 0.50 ms at 128 / 256 queries x
// 1 M rows (50 / 39 % of the HBM roof, 21 / 33 % of the MFMA roof).  Here the roles of the two memories are swapped:
//   * every wave keeps the B fragments of ITS OWN 32 queries in registers for the whole kernel - 48 fragments x 4 VGPRs =
//     192 registers at d = 768.  Up to 128 queries: four waves, one per SIMD (a wave may then use the full 512-entry
//     register file); up to 256: eight waves, two per SIMD, 256 registers each;
//   * the CORPUS streams HBM -> LDS once (global_load_lds, 16 B per lane; units of 32 rows x 24 k-slices = 25 KB with padded
//     rows, a ring of five units = four in flight) and every wave reads each A fragment from LDS once per MFMA - plain linear
//     addresses, bank-conflict-free by the padding (SQ_LDS_BANK_CONFLICT 1 % of SQ_LDS_IDX_ACTIVE); no query traffic at all
//     after the prologue;
//   * a query belongs to exactly one wave, so its candidate list (LDS, 12 or 16 slots) needs no lock and no atomics;
//   * one 4-byte DMA per block and wave brings the rows' |y|^2 (squared L2) and the other workgroups' thresholds into LDS: no
//     vector load, which would drain the staging queue (vmcnt is in-order), ever sits in the loop.
// Same operand roles (corpus rows = A, queries = B), same MFMA, same k-slice order 0 .. K/16 - 1 into one accumulator as every
// other kernel of the library: keys are bit-identical to the list kernel's (tools/rq_probe.py: 150 shapes, both metrics,
// k = 1 / 10 / 16, ragged last block, duplicate rows); thresholds are seeded by this kernel's own SEED mode.
// Measured (1 M x 768 fp16, k = 10, same box, kernel ms; profiles/r08j_rq_probe.log, profiles/r07_tuning.md), structureless unit
// rows - the worst case for thresholds: 97 / 128 queries 0.380 / 0.388 -> 0.278 / 0.284 ms (5.4 TB/s of corpus stream), 160 /
// 192 / 256 queries 0.460 / 0.470 / 0.499 -> 0.379 / 0.381 / 0.398 ms; on the bench's data (planted neighbours) the legs
// q128 / q192 / q256 read ~0.29 / 0.40 / 0.42 ms = 65 / 48 / 45 % of the HBM roof (list kernel: 50 / 41 / 39 %).
// What bounds it (s_memtime stamps of an instrumented copy, tools/lvs_rq_instrumented.hip.txt, first complete version): per unit
// of 24 MFMAs a wave spent ~1 180 cycles in the read + MFMA phase, ~550 issuing its share of the staging loads, ~500 at the
// unit's barrier and ~800 per unit in the block epilogue (whose slow path was entered for every second to fourth block: a
// workgroup sees 3 900 rows, so its lists never fill and its thresholds stay at the seed) - the staging data was always there
// (34 cycles at the vmcnt wait).  With one or two waves per SIMD none of that hides behind another wave's MFMAs.
#include "lvs_common.h"
#include "lvs_kstep.h"
#include "lvs_tile.h"

namespace {

using lvs_kstep::glds16;
using lvs_kstep::lds_read16;
using lvs_kstep::static_for;

constexpr int RQ_KMAX = LVS_RQ_KMAX;  // most list slots per query (k <= 16); calls with k <= 12 use 12 and spend the LDS on the ring

__device__ inline float rq_tau_float(uint32_t ord) { return ord == 0 ? -INFINITY : lvs_unord32(ord); }
__device__ inline float rq_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ inline float rq_max16(const f32x16& v) {
    const float a = rq_max3(v[0], v[1], v[2]), b = rq_max3(v[3], v[4], v[5]), c = rq_max3(v[6], v[7], v[8]);
    const float d = rq_max3(v[9], v[10], v[11]), e = rq_max3(v[12], v[13], v[14]);
    return rq_max3(rq_max3(a, b, c), rq_max3(d, e, v[15]), v[15]);
}
// which staging piece (0 .. lpw) goes out at MFMA step jj of a unit of uk steps: piece p at step p * uk / (lpw + 1) + 1
constexpr int rq_piece_at(int jj, int uk, int lpw) {
    for (int p = 0; p <= lpw; ++p)
        if (jj == p * uk / (lpw + 1) + 1) return p;
    return -1;
}
template <int N>
__device__ inline void rq_lds_wait(half8& a) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}

template <int NJ, int UK, int WAVES, int NQW, int KCAP>
struct RqGeom {
    static_assert(NJ % UK == 0, "a block is a whole number of units");
    static constexpr int U = NJ / UK;               // units per 32-row block
    // A unit is [32 rows][UK k-slices of 32 B] with every row PADDED by 16 B: the row stride (UK * 32 + 16) is an odd multiple
    // of 16 B, so the 16 lanes of a ds_read_b128 phase - 16 consecutive rows, one k-slice - hit 16 different 16-byte bank
    // groups with plain linear addresses: lane base + an immediate, no per-read address arithmetic at all
    static constexpr int ROWB = UK * 32 + 16;
    static constexpr int UDATA = 32 * ROWB;                      // bytes of a unit = UK KiB + 512
    static constexpr int NLOAD = UK + 1;                         // 1 KiB staging loads per unit (the last one half used)
    static constexpr int UB = NLOAD * 1024;                      // slot stride: the last load's unused half falls into the gap
    static constexpr int LPW = UK / WAVES;                       // full loads per wave and unit (+ one extra, taken in turns)
    static_assert(UK % WAVES == 0, "every wave stages the same number of 1 KiB pieces");
    static constexpr int nb_ring(int ring) { return (ring - 1 + U - 1) / U + 2; }  // blocks whose side words are in flight or in use
    // LDS = ring + the workgroup's candidate lists + its waves' side rings (row norms | shared thresholds), at most 160 KiB.
    // The ring is what keeps HBM busy: RING - 1 units (25 KB each at d = 768) are in flight per CU
    static constexpr int lds_bytes(int ring) { return ring * UB + WAVES * NQW * 32 * KCAP * 8 + WAVES * NQW * nb_ring(ring) * 256; }
    static constexpr int RING = lds_bytes(6) <= 160 * 1024 ? 6 : (lds_bytes(5) <= 160 * 1024 ? 5 : (lds_bytes(4) <= 160 * 1024 ? 4 : 3));
    static_assert(lds_bytes(RING) <= 160 * 1024, "LDS budget");
    static_assert(RING >= 3, "ring too short");
    static constexpr int NB_RING = nb_ring(RING);
};

__device__ inline void rq_glds4(const void* gsrc, void* ldst) {
    __builtin_amdgcn_global_load_lds((lvs_kstep::gbl_void_t*)gsrc, (lvs_kstep::lds_void_t*)ldst, 4, 0, 0);
}

// NJ = K / 16 (k-slices of a row), UK = k-slices per staged unit, WAVES = 4 (one wave per SIMD, up to 512 registers each) or
// 8 (two per SIMD, 256 each), NQW = 32-query blocks per wave (1; 2 with four waves), AD = A fragments in flight per wave
template <int NJ, int UK, int WAVES, int NQW, int AD, int KCAP, bool SEED>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void lvs_rq_kernel(const LvsRqArgs a) {
    constexpr bool RQ_SETPRIO = true;
    constexpr bool SPREAD = true;  // the staging loads of a unit go out between the MFMAs of the unit being computed
    constexpr bool PHASED = false;  // WAVES == 8 measured slower (0.51 vs 0.46 ms at 256 queries): see the comment at the block loop
    using G = RqGeom<NJ, UK, WAVES, NQW, KCAP>;
    constexpr int RQ_WAVES = WAVES;
    constexpr int RQ_KCAP = KCAP;
    constexpr int NB_RING = G::NB_RING;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    u64* lists = (u64*)(smem + G::RING * G::UB);  // [NQ][RQ_KCAP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // per wave and block in flight, 64 words that ride in with the block's rows (ONE 4-byte DMA per block): words 0 .. 31 the
    // |y|^2 of the block's rows (squared L2), words 32 .. 63 the shared thresholds gtau[] of the wave's 32 queries as the other
    // workgroups have left them - read every block, for free, instead of through a vector load that would drain the staging
    float* side = (float*)(smem + G::RING * G::UB + (size_t)WAVES * NQW * 32 * RQ_KCAP * 8) + wave * (NB_RING * 64);
    const int k = a.k;
    // Beyond 256 queries the call is cut into GROUPS of 256: a corpus range is scanned by `groups` sibling workgroups, one per
    // group.  Workgroup b lands on XCD b % 8; the siblings of a range take consecutive slots of ONE XCD, so the range is read
    // from HBM once and by the siblings through that XCD's L2 (they run in step: same rows, same work)
    int range, group;  // (lvs_rq_item, lvs_tile.h: the siblings of a range share an XCD; beyond 32 groups one range per XCD)
    if (!lvs_rq_item(blockIdx.x, a.groups, a.nparts, range, group)) return;
    const int qbase = group * (WAVES * NQW * 32);
    const _Float16* xq = (const _Float16*)a.xq;
    const char* xb = (const char*)a.xb;
    const long long ldb2 = a.ldb * 2;  // bytes per corpus row
    const bool l2 = a.metric == LVS_METRIC_L2;

    // ---- this wave's queries -> registers as B fragments: block qb*WAVES + wave, fragment j, lane l = query (l & 31), halfs (l >> 5) * 8
    half8 breg[NQW][NJ];
    int qidx[NQW];      // query number (call-wide)
    bool qvalid[NQW];
    float tauf[NQW], qnv[NQW];
    uint32_t gord[NQW];
#pragma unroll
    for (int qb = 0; qb < NQW; ++qb) {
        qidx[qb] = qbase + (qb * RQ_WAVES + wave) * 32 + (lane & 31);
        qvalid[qb] = qidx[qb] < a.nq;
        const int qrow = qvalid[qb] ? qidx[qb] : a.nq - 1;
        const _Float16* qp = xq + (long long)qrow * a.ldq + (lane >> 5) * 8;
#pragma unroll
        for (int j = 0; j < NJ; ++j) breg[qb][j] = *(const half8*)(qp + j * 16);
        gord[qb] = (!SEED && qvalid[qb]) ? a.gtau[qidx[qb]] : 0u;
        tauf[qb] = rq_tau_float(gord[qb]);
        qnv[qb] = l2 ? a.qn[qrow] : 0.f;
    }
    u64* mylists = lists + (long long)(wave * NQW * 32) * RQ_KCAP;  // wave-private: queries are never shared between waves
    if (!SEED) {
        for (int i = lane; i < NQW * 32 * RQ_KCAP; i += 64) mylists[i] = 0;
    }
    float seedbest[NQW];
#pragma unroll
    for (int qb = 0; qb < NQW; ++qb) seedbest[qb] = -INFINITY;

    // ---- corpus range of this workgroup, in 32-row blocks
    const long long nblocks = (a.nb + 31) / 32;
    const long long b0 = (long long)range * a.blocks_per_wg;
    const long long b1 = b0 + a.blocks_per_wg < nblocks ? b0 + a.blocks_per_wg : nblocks;
    if (b0 >= b1) {  // an empty range still owns its slice of the output (its group's queries)
        const int q1 = a.nq < qbase + WAVES * NQW * 32 ? a.nq : qbase + WAVES * NQW * 32;
        if (!SEED) {
            for (int i = qbase * k + tid; i < q1 * k; i += RQ_WAVES * 64) a.out[(long long)range * a.nq * k + i] = 0;
        } else {
            for (int i = qbase + tid; i < q1; i += RQ_WAVES * 64) a.seed_out[(long long)range * a.nq + i] = -INFINITY;
        }
        return;
    }
    const int nblk = (int)(b1 - b0);
    const int total_units = nblk * G::U;

    // ---- staging: load m of a unit fills LDS bytes [m * 1024, + 1024) of the unit's slot, lane l the 16-byte granule
    // g = m * 64 + l = (row, piece) with 2 UK + 1 granules per padded row (the last one is the pad: it re-reads the row's last
    // piece).  Wave w issues loads w * LPW .. + LPW - 1 of every unit and, in turns, the half-used last one (m = UK).  The
    // per-lane source offsets are constants of the kernel: LPW + 1 registers.
    auto src_off = [&](int m, int ln, int last_row) {
        const int g = m * 64 + ln;
        int row = g / (2 * UK + 1);
        int pc = g - row * (2 * UK + 1);
        pc = pc < 2 * UK ? pc : 2 * UK - 1;
        row = row < 31 ? row : 31;            // (granules past the unit's end: lanes 32 .. 63 of the last load)
        row = row < last_row ? row : last_row;
        return (unsigned)(row * (int)ldb2 + pc * 16);
    };
    unsigned soff[G::LPW + 1];
#pragma unroll
    for (int i = 0; i < G::LPW; ++i) soff[i] = src_off(wave * G::LPW + i, lane, 31);
    soff[G::LPW] = src_off(UK, lane, 31);
    // Staging of unit n is issued in LPW + 1 PIECES (one 1 KiB load each; the last piece is the half-used load - when it is this
    // wave's turn - and the block's side words): `issue_unit` issues them in one go (prologue), the unit loop spreads them between
    // the MFMAs of the unit being computed, where their address arithmetic and issue slots cost nothing.
    struct IssuePrep {
        char* dst;
        const char* src;
        long long row0;
        int last, blk, kh;
        bool extra, real;
    };
    auto issue_prep = [&](int n) {  // wave-uniform: lives in scalar registers
        IssuePrep pr;
        const int nn = n < total_units ? n : total_units - 1;  // past the end the last unit is loaded again into a free slot
        pr.blk = nn / G::U;
        pr.kh = nn - pr.blk * G::U;
        pr.row0 = (b0 + pr.blk) * 32;
        pr.dst = ring + (n % G::RING) * G::UB;
        pr.src = xb + pr.row0 * ldb2 + (long long)pr.kh * UK * 32;
        pr.last = (int)(a.nb - 1 - pr.row0);  // rows past the corpus' end (its last, partial block) re-read the last row
        pr.extra = (n % WAVES) == wave;       // whose turn the unit's last, half-used load is
        pr.real = n < total_units;
        return pr;
    };
    auto issue_piece = [&](auto pc, const IssuePrep& pr) {
        constexpr int i = decltype(pc)::value;
        if constexpr (i < G::LPW) {
            const unsigned off = pr.last >= 31 ? soff[i] : src_off(wave * G::LPW + i, lane, pr.last);
            glds16(pr.src + off, pr.dst + (wave * G::LPW + i) * 1024);
        } else {
            if (pr.extra) {
                const unsigned off = pr.last >= 31 ? soff[G::LPW] : src_off(UK, lane, pr.last);
                glds16(pr.src + off, pr.dst + UK * 1024);
            }
            if (!SEED && pr.kh == 0 && pr.real) {
                // the block's side words: lanes 0 .. 31 -> |y|^2 of its rows (any valid word under inner product), lanes 32 .. 63 ->
                // the shared thresholds of this wave's queries.  NQW = 2: one DMA per query block, slots 64 words apart
                int row = lane & 31;
                row = row < pr.last ? row : pr.last;
#pragma unroll
                for (int qb = 0; qb < NQW; ++qb) {
                    const int q = qvalid[qb] ? qidx[qb] : 0;
                    const void* p = (lane < 32 && l2) ? (const void*)(a.bn + pr.row0 + row) : (const void*)(a.gtau + q);
                    rq_glds4(p, side + ((pr.blk % NB_RING) * NQW + qb) * 64);
                }
            } else if (SEED && l2 && pr.kh == 0 && pr.real) {
                int row = lane & 31;
                row = row < pr.last ? row : pr.last;
                rq_glds4(a.bn + pr.row0 + row, side + (pr.blk % NB_RING) * NQW * 64);
            }
        }
    };
    auto issue_unit = [&](int n) {
        const IssuePrep pr = issue_prep(n);
        static_for<G::LPW + 1>([&](auto pc) { issue_piece(pc, pr); });
    };
    // ---- fragment reads: lane (r = l & 31, h = l >> 5) reads bytes [(2 jj + h) * 16, + 16) of padded row r: lane base + jj * 32
    unsigned o_base = (unsigned)(unsigned long long)ring + (unsigned)((lane & 31) * G::ROWB + (lane >> 5) * 16);
    int slot = 0;
    f32x16 acc[NQW];
    auto zero_acc = [&]() {
#pragma unroll
        for (int qb = 0; qb < NQW; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qb][r] = 0.f;
    };
    // one unit of the current block: UK fragment reads (AD - 1 steps ahead of their MFMAs through a ring of AD registers) and
    // UK x NQW MFMAs; then the fragment base moves to the ring's next slot
    auto mfma_phase = [&](auto khc, const IssuePrep& pr) {
        constexpr int kh = decltype(khc)::value;
        half8 Af[AD];
        static_for<AD - 1>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            if constexpr (jj < UK) lds_read16<jj * 32>(Af[jj % AD], o_base);
        });
        static_for<UK>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            constexpr int ahead = jj + AD - 1;
            if constexpr (ahead < UK) lds_read16<ahead * 32>(Af[ahead % AD], o_base);
            if constexpr (SPREAD) {  // piece p of the NEXT-but-three unit's staging goes out at step p * UK / (LPW + 1) + 1
                constexpr int piece = rq_piece_at(jj, UK, G::LPW);
                if constexpr (piece >= 0) issue_piece(std::integral_constant<int, piece>{}, pr);
            }
            __builtin_amdgcn_sched_barrier(0);
            rq_lds_wait<(ahead < UK) ? AD - 1 : (UK - 1 - jj)>(Af[jj % AD]);
#pragma unroll
            for (int qb = 0; qb < NQW; ++qb)
                acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[jj % AD], breg[qb][kh * UK + jj], acc[qb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        ++slot;
        const unsigned delta = slot == G::RING ? (unsigned)(-(G::RING - 1) * G::UB) : (unsigned)G::UB;
        if (slot == G::RING) slot = 0;
        o_base += delta;
    };
    // ---- block epilogue: 32 rows x this wave's queries; lane holds query qidx[*], rows row0 + (r&3) + 8*(r>>2) + 4*(lane>>5)
    auto epilogue = [&](int blk) {
        const long long row0 = (b0 + blk) * 32;
        const long long rbase = row0 + 4 * (lane >> 5);
        const float* sideb = side + (blk % NB_RING) * NQW * 64;
        if (l2) {
            // |y|^2 of this lane's 16 rows from the wave's side words (rows 4 (lane >> 5) + {0..3} + 8 m: four 16-byte reads)
            const float* bnb = sideb + 4 * (lane >> 5);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 bn4 = *(const f32x4*)(bnb + 8 * m);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int qb = 0; qb < NQW; ++qb)
                        acc[qb][m * 4 + e] = -fmaxf((qnv[qb] + bn4[e]) - 2.0f * acc[qb][m * 4 + e], 0.f);
            }
        }
#pragma unroll
        for (int qb = 0; qb < NQW; ++qb) {
            if constexpr (SEED) {
                seedbest[qb] = fmaxf(seedbest[qb], rq_max16(acc[qb]));  // the sample holds whole 32-row blocks only
                continue;
            }
            {   // the shared threshold as it was when this block's rows were requested (a lower bound of the k-th best score
                // over ALL workgroups' rows so far): every block, every workgroup tightens from what the others have found
                // (reading it only every 8th block measured 7 % slower)
                const uint32_t g = __float_as_uint(sideb[qb * 64 + 32 + (lane & 31)]);
                gord[qb] = g > gord[qb] ? g : gord[qb];
                tauf[qb] = fmaxf(tauf[qb], rq_tau_float(gord[qb]));
            }
            const int ql = qb * 32 + (lane & 31);  // list index inside this wave
            const bool th = qvalid[qb] && (rq_max16(acc[qb]) >= tauf[qb]);
            if (__any(th)) {
                uint32_t best_tau = 0;  // the tightest k-th key this lane's query reached in this block
                // one candidate of one lane is the usual case: every lane finds its FIRST candidate (value and register index) in
                // straight-line code - three VALU operations per score, no ballots - and only when some lane holds more than one
                // do the remaining registers get their turn
                float fs = 0.f;
                int fr = 16, cnt = 0;
#pragma unroll
                for (int r = 15; r >= 0; --r) {
                    const bool c = th && acc[qb][r] >= tauf[qb];
                    fs = c ? acc[qb][r] : fs;
                    fr = c ? r : fr;
                    cnt += c ? 1 : 0;
                }
                auto offer = [&](bool cand, float s, int r) __attribute__((always_inline)) {
                    bool pending = false;
                    u64 key = 0;
                    if (cand && s >= tauf[qb]) {
                        const long long row = rbase + (r & 3) + 8 * (r >> 2);
                        if (row < a.nb) {
                            const uint32_t id = a.row_ids ? a.row_ids[row] : (uint32_t)(row + a.id_offset);
                            key = lvs_pack_key(s, id);
                            pending = (uint32_t)(key >> 32) >= gord[qb];
                        }
                    }
                    unsigned long long pm = __ballot(pending);
                    while (pm) {  // wave-cooperative sorted insertion: lane j < k owns slot j (no lock: the list is this wave's)
                        const int src = __ffsll((long long)pm) - 1;
                        pm &= pm - 1;
                        const uint32_t klo = __builtin_amdgcn_readlane((uint32_t)key, src);
                        const uint32_t khi = __builtin_amdgcn_readlane((uint32_t)(key >> 32), src);
                        const u64 ukey = ((u64)khi << 32) | klo;
                        const int uq = __builtin_amdgcn_readlane(ql, src);
                        u64* UL = mylists + uq * RQ_KCAP;
                        u64 mine = 0, prev = ~0ull;
                        if (lane < k) {
                            mine = UL[lane];
                            if (lane > 0) prev = UL[lane - 1];
                        }
                        u64 newv = 0;
                        if (lane < k) newv = mine > ukey ? mine : (prev > ukey ? ukey : prev);
                        __builtin_amdgcn_wave_barrier();
                        if (lane < k) UL[lane] = newv;
                        const uint32_t ntau = __builtin_amdgcn_readlane((uint32_t)(newv >> 32), k - 1);
                        if (ql == uq) {
                            tauf[qb] = fmaxf(tauf[qb], rq_tau_float(ntau));
                            best_tau = ntau > best_tau ? ntau : best_tau;
                        }
                    }
                };
                offer(fr < 16, fs, fr);
                if (__any(cnt > 1)) {
#pragma unroll
                    for (int r = 1; r < 16; ++r) offer(th && r > fr, acc[qb][r], r);
                }
                // lanes l and l + 32 hold the same query: both continue from the tighter threshold; a full list's k-th key is
                // published to the other workgroups (fire and forget: nothing waits for the atomic)
                tauf[qb] = fmaxf(tauf[qb], __shfl_xor(tauf[qb], 32, 64));
                if (best_tau > gord[qb] && qvalid[qb]) {
                    atomicMax(&a.gtau[qidx[qb]], best_tau);
                    gord[qb] = best_tau;
                }
            }
        }
    };

    for (int n = 0; n < G::RING - 1; ++n) issue_unit(n);
    // two waves per SIMD: the second-dispatched half loses issue arbitration to the older half on every unit; static priority
    // for it evens that out (the tile kernels' T5 recipe)
    if (WAVES == 8 && wave >= 4 && RQ_SETPRIO) __builtin_amdgcn_s_setprio(1);

    // PHASED (eight waves, two per SIMD): the two waves of a SIMD run a unit's pieces in OPPOSITE order, so that one wave's
    // instruction-bound pieces - issuing its share of the staging loads, the block epilogue - run beside the other's MFMAs
    // instead of both queueing for the same issue slots while the matrix pipe idles (stamps of the first version, DESIGN.md
    // 3.3c: per unit ~550 cycles of issuing and ~800 of epilogue against 750 of MFMAs, all eight waves in the same phase):
    //   waves 4 .. 7 ("early"):  barrier | issue loads | MFMAs            and, after a block's last unit, its epilogue
    //   waves 0 .. 3 ("late"):   barrier | epilogue of the PREVIOUS block (first unit only) | MFMAs | issue loads
    // Every wave still issues once per unit after the unit's barrier (the slot it fills was read in the unit before) and
    // waits for the same number of younger loads; a late wave's accumulators and side words outlive the block by one unit
    // (the side ring holds the blocks in flight + 2).
    const bool late = PHASED && wave < WAVES / 2;
    for (int blk = 0; blk < nblk; ++blk) {
        static_for<G::U>([&](auto khc) {
            constexpr int kh = decltype(khc)::value;
            const int n = blk * G::U + kh;
            // unit n has landed once this wave's loads of the RING - 2 younger units are all that is in flight (the extra
            // half-load and the 4-byte side words ride in the same queue: with them more than LPW loads per unit are in flight
            // behind unit n, so the wait is only more conservative); the barrier makes that true for every wave's share and
            // tells everybody that unit n - 1's slot is free again
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((G::RING - 2) * G::LPW) : "memory");
            __builtin_amdgcn_s_barrier();
            const IssuePrep pr = issue_prep(n + G::RING - 1);
            if (!late) {
                if (!SPREAD) {
                    static_for<G::LPW + 1>([&](auto pc) { issue_piece(pc, pr); });
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kh == 0) zero_acc();
                mfma_phase(khc, pr);
            } else {
                if (kh == 0) {
                    if (blk > 0) epilogue(blk - 1);
                    zero_acc();
                }
                mfma_phase(khc, pr);
                __builtin_amdgcn_sched_barrier(0);
                if (!SPREAD) static_for<G::LPW + 1>([&](auto pc) { issue_piece(pc, pr); });
            }
        });
        if (!late) epilogue(blk);
    }
    if (late) epilogue(nblk - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail loads still target the ring
    if constexpr (SEED) {
#pragma unroll
        for (int qb = 0; qb < NQW; ++qb) {
            const float m = fmaxf(seedbest[qb], __shfl_xor(seedbest[qb], 32, 64));
            if (lane < 32 && qvalid[qb]) a.seed_out[(long long)range * a.nq + qidx[qb]] = m;
        }
        return;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < NQW * 32 * k; i += 64) {
        const int ql = i / k, j = i - ql * k;
        const int q = qbase + ((ql >> 5) * RQ_WAVES + wave) * 32 + (ql & 31);
        if (q < a.nq) a.out[((long long)range * a.nq + q) * k + j] = mylists[ql * RQ_KCAP + j];
    }
    if (lane < 32) {
#pragma unroll
        for (int qb = 0; qb < NQW; ++qb) {
            const uint32_t lo = (uint32_t)(mylists[(qb * 32 + lane) * RQ_KCAP + k - 1] >> 32);
            if (lo && qvalid[qb]) atomicMax(&a.gtau[qidx[qb]], lo);
        }
    }
}

template <int NJ, int UK, int WAVES, int NQW, int AD, int KCAP, bool SEED>
hipError_t rq_launch_k(const LvsRqArgs& a, int grid, hipStream_t stream) {
    using G = RqGeom<NJ, UK, WAVES, NQW, KCAP>;
    const size_t lds = (size_t)G::lds_bytes(G::RING);
    static LvsPerDeviceOnce attr;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!attr.done(dev, lds)) {
        e = hipFuncSetAttribute((const void*)lvs_rq_kernel<NJ, UK, WAVES, NQW, AD, KCAP, SEED>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr.set(dev, lds);
    }
    if (a.groups > 1 && WAVES * NQW * 32 != LVS_RQ_GROUPQ) return hipErrorInvalidValue;  // groups are the eight-wave variant's
    hipLaunchKernelGGL((lvs_rq_kernel<NJ, UK, WAVES, NQW, AD, KCAP, SEED>), dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError();
}
template <int NJ, int UK, int WAVES, int NQW, int AD, bool SEED>
hipError_t rq_launch_one(const LvsRqArgs& a, int grid, hipStream_t stream) {
    if (a.k <= 12) return rq_launch_k<NJ, UK, WAVES, NQW, AD, 12, SEED>(a, grid, stream);
    return rq_launch_k<NJ, UK, WAVES, NQW, AD, RQ_KMAX, SEED>(a, grid, stream);
}

// up to 128 queries: four waves (one per SIMD), one query block each; up to 256: eight waves (two per SIMD, 256 registers
// each - 192 of them B fragments at d = 768), one query block each.  (-DLVS_TUNING, LVS_RQ_MODE=2: four waves with two
// query blocks each - 384 registers of B fragments per wave - measured slower: half of them end up in accumulation VGPRs
// and are copied back one MFMA at a time.)
template <int NJ, int UK, bool SEED>
hipError_t rq_launch_shape(const LvsRqArgs& a, int grid, hipStream_t stream) {
#ifdef LVS_TUNING
    if (a.nq <= 128 && lvs_tune("LVS_RQ_AD4", 0) == 3) return rq_launch_one<NJ, UK, 4, 1, 3, SEED>(a, grid, stream);
#endif
    if (a.nq <= 128) return rq_launch_one<NJ, UK, 4, 1, 6, SEED>(a, grid, stream);
#ifdef LVS_TUNING
    if (a.groups == 1 && lvs_tune("LVS_RQ_MODE", 0) == 2) return rq_launch_one<NJ, UK, 4, 2, 3, SEED>(a, grid, stream);
#endif
    // (the depth of the fragment read-ahead makes no difference with two waves per SIMD - 2 / 3 / 4 registers sets measured
    // alike, profiles/r07_tuning.md - and at d = 768 the B fragments leave 64 registers for everything else: two sets)
    return rq_launch_one<NJ, UK, 8, 1, (NJ >= 48 ? 2 : 3), SEED>(a, grid, stream);
}

}  // namespace

// Corpus ranges a launch of `groups` query groups uses at most: every range has one workgroup per group, all on one XCD
// (32 CUs), a workgroup fills a CU
static int rq_max_ranges(int groups) { return lvs_rq_ranges_for(groups); }

// Does the register-resident-queries kernel take this call?  fp16 k-slices of one K segment (d padded to 256, 384, 512 or
// 768 halfs), k <= 16, a corpus long enough to give every CU a few blocks, and 97 .. 256 queries - or up to LVS_RQ_MAXQ in
// groups of 256 when the groups' workgroups fill (nearly) every CU (>= 224 of 256): 2 .. 8, 10, 14, 15 or 16 groups.
bool lvs_rq_shape_ok(int dpad, int k) {  // operand shapes the register-resident kernels are built for
    const int nj = dpad / 16;
    return dpad % 16 == 0 && k >= 1 && k <= RQ_KMAX && (nj == 16 || nj == 24 || nj == 32 || nj == 48);
}
bool lvs_rq_fits(int64_t nq, int64_t nb, int dpad, int k) {
    const int nj = dpad / 16;
    if (!(nq > 96 && nq <= LVS_RQ_MAXQ && k >= 1 && k <= RQ_KMAX && nb >= 32768 && (nj == 16 || nj == 24 || nj == 32 || nj == 48))) return false;
    const int groups = (int)((nq + LVS_RQ_GROUPQ - 1) / LVS_RQ_GROUPQ);
    if (groups == 1) return true;
    return groups <= lvs_tune("LVS_RQ_MAXG", LVS_RQ_MAXQ / LVS_RQ_GROUPQ) && groups * rq_max_ranges(groups) >= 224 &&
           nb >= (int64_t)32768 * groups;
}

// On return a.nparts = candidate lists per query in a.out ([nparts][nq][k]) (SEED: rows of a.seed_out [nparts][nq]).
hipError_t lvs_rq_launch(LvsRqArgs& a, int dpad, hipStream_t stream) {
    const int64_t nblocks = (a.nb + 31) / 32;
    a.groups = (a.nq + LVS_RQ_GROUPQ - 1) / LVS_RQ_GROUPQ;
    if (a.groups < 1) a.groups = 1;
    int64_t ranges = rq_max_ranges(a.groups);
    if (ranges < 1) return hipErrorInvalidValue;
    if (ranges > (nblocks + 3) / 4) ranges = (nblocks + 3) / 4;
    if (ranges < 1) ranges = 1;
    a.blocks_per_wg = (int)((nblocks + ranges - 1) / ranges);
    ranges = (nblocks + a.blocks_per_wg - 1) / a.blocks_per_wg;
    a.nparts = (int)ranges;
    if (a.groups > 32 && a.groups % 32 != 0) return hipErrorInvalidValue;  // (chunks of a larger call: 32 x 2^i groups)
    const int grid = lvs_rq_grid(a.groups, (int)ranges);
    const bool seed = a.seed_out != nullptr;
    switch (dpad / 16) {
        case 48: return seed ? rq_launch_shape<48, 24, true>(a, grid, stream) : rq_launch_shape<48, 24, false>(a, grid, stream);
        case 32: return seed ? rq_launch_shape<32, 32, true>(a, grid, stream) : rq_launch_shape<32, 32, false>(a, grid, stream);
        case 24: return seed ? rq_launch_shape<24, 24, true>(a, grid, stream) : rq_launch_shape<24, 24, false>(a, grid, stream);
        case 16: return seed ? rq_launch_shape<16, 16, true>(a, grid, stream) : rq_launch_shape<16, 16, false>(a, grid, stream);
        default: return hipErrorInvalidValue;
    }
}

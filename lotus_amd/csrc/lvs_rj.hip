// Join-scale batches (from a few hundred queries up: lotus/sem_ops/sem_sim_join.py:132-134 -> faiss_vs.py:67) with the queries
// resident in registers and ONE wave per SIMD - lvs_rq_kernel's successor for the calls it serves in groups of 256 queries (r6).
//
// lvs_rq_kernel keeps 32 queries per wave (192 registers of B fragments at d = 768, two waves per SIMD): every A fragment read
// from LDS feeds ONE MFMA.  Here a wave owns the whole register file of its SIMD - 256 VGPRs + 256 accumulation registers - and
// keeps 64 queries: 96 B fragments = 384 registers, of which 60 live in NAMED accumulation registers a[0:239] (hipcc's own
// allocation keeps MFMA operands in VGPRs and copies such fragments back four v_accvgpr_read per MFMA), the ring of four A
// fragments in a[240:255] (ds_read_b128 writes AGPRs, the MFMA reads both operands there), 36 B fragments and the two
// accumulators in VGPRs.  An A fragment now feeds TWO MFMAs: 0.5 LDS fragment reads per MFMA against 0.75 in the list kernel
// and 1.0 in lvs_rq_kernel, and - as there - no query traffic at all after the prologue and the corpus staged once per 256
// queries (half the list kernel's L2 -> LDS bytes per flop).  Measured on one box (tools/rj_ablate.py, 4 096 queries x 1 M x
// 768): fragment reads + MFMAs alone run at 1.53 PFLOP/s (the MFMA-only ceiling on random operands is 1.6-1.68,
// tools/probe_mfma_chain.hip); what separates the kernel from that is everything a single wave cannot hide behind a partner.
// So, unlike lvs_rq_kernel:
//   * the fragment pipeline runs ACROSS the unit barriers (a barrier certifies the unit after the one about to be computed);
//   * a block's first MFMAs take the constant 0 as C (no accumulator clears);
//   * the staging cursor is advanced incrementally in scalar registers, and the kernel only sees WHOLE 32-row blocks - no
//     clamping arithmetic per load; the caller searches the corpus' last nb % 32 rows with lvs_rq_kernel (same arithmetic);
//   * candidates are not inserted where they are found: a lane whose score reaches its query's threshold APPENDS the key to a
//     per-wave buffer in LDS (one ballot + one write), and the buffer is drained - sorted insertions, thresholds, publication -
//     once it holds enough of them.  A wave-wide slow path per candidate (~1 500 cycles with nobody to hide it) becomes ~100
//     cycles per candidate; the thresholds lag by at most a buffer's worth, which only admits a few more candidates.
// Same operand roles, same MFMA, same k-slice order into one accumulator (the first with C = 0) as every other kernel of the
// library: keys bit-identical to the list kernel's.  Exactness of the deferred insertion: a score is dropped only when it is
// below a threshold that is the k-th best key's score of a list holding real rows of the searched set (own list or another
// workgroup's through gtau) - never above the final k-th best - and everything appended is inserted before the lists are written.
#include "lvs_common.h"
#include "lvs_kstep.h"
#include "lvs_tile.h"

namespace {

using lvs_kstep::glds16;
using lvs_kstep::static_for;

#define RJ_C10(p) "a" #p "0", "a" #p "1", "a" #p "2", "a" #p "3", "a" #p "4", "a" #p "5", "a" #p "6", "a" #p "7", "a" #p "8", "a" #p "9"
#define RJ_CLOBBER_AGPRS                                                                                                          \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", RJ_C10(1), RJ_C10(2), RJ_C10(3), RJ_C10(4), RJ_C10(5), RJ_C10(6), \
        RJ_C10(7), RJ_C10(8), RJ_C10(9), RJ_C10(10), RJ_C10(11), RJ_C10(12), RJ_C10(13), RJ_C10(14), RJ_C10(15), RJ_C10(16),       \
        RJ_C10(17), RJ_C10(18), RJ_C10(19), RJ_C10(20), RJ_C10(21), RJ_C10(22), RJ_C10(23), RJ_C10(24), "a250", "a251", "a252",    \
        "a253", "a254", "a255"
constexpr int RJ_A0 = 240;  // first register of the A-fragment ring (four fragments)
constexpr int RJ_AD = 4;
constexpr int RJ_CAP = 128;        // candidate buffer entries per wave
constexpr int RJ_DRAIN_EVERY = 16;  // every wave drains every that many blocks (LvsRqArgs::drain_every)

template <int I>
__device__ inline void rj_load_b(const void* p) {  // 16 bytes per lane -> a[4 I : 4 I + 3]
    asm volatile("global_load_dwordx4 a[%1:%2], %0, off" ::"v"(p), "n"(4 * I), "n"(4 * I + 3) : "memory", RJ_CLOBBER_AGPRS);
}
template <int R, int OFFSET>
__device__ inline void rj_read_a(unsigned addr) {  // A fragment -> ring register R
    static_assert(OFFSET >= 0 && OFFSET < 65536, "ds_read offset field is 16 bits");
    asm volatile("ds_read_b128 a[%1:%2], %0 offset:%3" ::"v"(addr), "n"(RJ_A0 + 4 * R), "n"(RJ_A0 + 4 * R + 3), "n"(OFFSET)
                 : "memory", RJ_CLOBBER_AGPRS);
}
// acc (+)= A(ring R) x B.  ZERO: the block's first k-slice - C is the constant 0
template <int R, int I, bool ZERO>
__device__ inline void rj_mfma_aa(f32x16& acc) {  // B = fragment I in accumulation registers
    if constexpr (ZERO)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%1:%2], a[%3:%4], 0"
                     : "=v"(acc)
                     : "n"(RJ_A0 + 4 * R), "n"(RJ_A0 + 4 * R + 3), "n"(4 * I), "n"(4 * I + 3)
                     : RJ_CLOBBER_AGPRS);
    else
        asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%1:%2], a[%3:%4], %0"
                     : "+v"(acc)
                     : "n"(RJ_A0 + 4 * R), "n"(RJ_A0 + 4 * R + 3), "n"(4 * I), "n"(4 * I + 3)
                     : RJ_CLOBBER_AGPRS);
}
template <int R, bool ZERO>
__device__ inline void rj_mfma_av(f32x16& acc, const half8& b) {  // B in VGPRs
    if constexpr (ZERO)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%1:%2], %3, 0" : "=v"(acc) : "n"(RJ_A0 + 4 * R), "n"(RJ_A0 + 4 * R + 3), "v"(b) : RJ_CLOBBER_AGPRS);
    else
        asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%1:%2], %3, %0" : "+v"(acc) : "n"(RJ_A0 + 4 * R), "n"(RJ_A0 + 4 * R + 3), "v"(b) : RJ_CLOBBER_AGPRS);
}
// one 1 KiB staging load: 16 bytes per lane from base (wave-uniform) + off (per lane) to LDS bytes [lds, lds + 1024) - the
// scalar-base form, spelled out (hipcc adds the base to a 64-bit copy of every lane offset instead: two registers per offset)
__device__ inline void rj_glds16(const char* base, unsigned off, unsigned lds) {
    // (readfirstlane: a wave-uniform value hipcc happens to keep in a VECTOR register is handed to an "s" operand as it is)
    const unsigned long long b = (unsigned long long)base;
    const unsigned long long bs = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned ls = (unsigned)__builtin_amdgcn_readfirstlane((int)lds);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(bs), "s"(ls) : "memory", "m0");
}
__device__ inline void rj_glds4(const void* gsrc, void* ldst) {
    __builtin_amdgcn_global_load_lds((lvs_kstep::gbl_void_t*)gsrc, (lvs_kstep::lds_void_t*)ldst, 4, 0, 0);
}
__device__ inline float rj_tau_float(uint32_t ord) { return ord == 0 ? -INFINITY : lvs_unord32(ord); }
__device__ inline float rj_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ inline float rj_max16(const f32x16& v) {
    const float a = rj_max3(v[0], v[1], v[2]), b = rj_max3(v[3], v[4], v[5]), c = rj_max3(v[6], v[7], v[8]);
    const float d = rj_max3(v[9], v[10], v[11]), e = rj_max3(v[12], v[13], v[14]);
    return rj_max3(rj_max3(a, b, c), rj_max3(d, e, v[15]), v[15]);
}
constexpr int rj_piece_at(int jj, int uk, int lpw) {  // staging piece p (0 .. lpw) goes out at MFMA step p * uk / (lpw + 1) + 1
    for (int p = 0; p <= lpw; ++p)
        if (jj == p * uk / (lpw + 1) + 1) return p;
    return -1;
}

#ifdef LVS_TUNING
// tuning builds: cycle / event counters of the DBG = 32 instantiation (tools/rj_ablate.py), summed over all waves
__device__ unsigned long long rj_dbg[16];
#endif

template <int NJ>
struct RjGeom {
    static constexpr int UK = NJ % 24 == 0 ? 24 : 16;  // k-slices per staged unit
    static_assert(NJ % UK == 0 && UK % RJ_AD == 0 && UK % 4 == 0, "units tile a row; the A ring closes over a unit");
    static constexpr int U = NJ / UK;
    static constexpr int ROWB = UK * 32 + 16;   // padded row of a unit (odd multiple of 16 B: conflict-free linear fragment reads)
    static constexpr int NLOAD = UK + 1;        // 1 KiB staging loads per unit (the last one half used)
    static constexpr int UB = NLOAD * 1024;     // ring slot stride
    static constexpr int LPW = UK / 4;          // full loads per wave and unit (+ the half-used one, taken in turns)
    static constexpr int NB_AGPR = 2 * NJ < 60 ? 2 * NJ : 60;  // B fragments in accumulation registers
    static constexpr int NB_VGPR = 2 * NJ - NB_AGPR;
    static constexpr int nb_ring(int ring) { return (ring - 1 + U - 1) / U + 2; }  // blocks whose side words are in flight or in use
    static constexpr int lds_bytes(int ring, int kcap) {
        return ring * UB + 256 * kcap * 8 + 8 * nb_ring(ring) * 256 + 4 * RJ_CAP * 12;
    }
    static constexpr int ring(int kcap) {
        for (int r = 7; r >= 4; --r)
            if (lds_bytes(r, kcap) <= 160 * 1024) return r;
        return 0;
    }
};

// NJ = K / 16; KCAP = list slots per query (k <= KCAP); L2: squared-L2 scores (the rows' |y|^2 ride in with the block)
// RANGE: the threshold join of sem_dedup (lvs_range_join) on the same frame - no lists, no thresholds to exchange: a score above
// the (one, constant) threshold goes straight to the global pair list
template <int NJ, int KCAP, bool L2, int DBG = 0, bool RANGE = false>
__global__ __launch_bounds__(256, 1) void lvs_rj_kernel(const LvsRqArgs a) {
    static_assert(!RANGE || !L2, "the RANGE epilogue is built for inner-product scores");
    using G = RjGeom<NJ>;
    constexpr int UK = G::UK, U = G::U, LPW = G::LPW, AD = RJ_AD, RING = G::ring(KCAP), NB_RING = G::nb_ring(RING);
    static_assert(RING >= 4, "LDS budget: the ring must hold the unit being read, the certified one and two in flight");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    u64* lists = (u64*)(smem + RING * G::UB);                         // [256][KCAP]
    float* side_all = (float*)(lists + 256 * KCAP);                   // [4 waves][NB_RING][2][64]
    u64* ckey_all = (u64*)(side_all + 8 * NB_RING * 64);              // [4][RJ_CAP] candidate keys
    uint32_t* cql_all = (uint32_t*)(ckey_all + 4 * RJ_CAP);           // [4][RJ_CAP] ... and the wave-local query (0 .. 63) of each
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* side = side_all + wave * (NB_RING * 2 * 64);
    u64* ckey = ckey_all + wave * RJ_CAP;
    uint32_t* cql = cql_all + wave * RJ_CAP;
    u64* mylists = lists + (long long)(wave * 64) * KCAP;
    const int k = a.k;
    // groups of 256 queries; the siblings of a corpus range take consecutive slots of ONE XCD (lvs_rq_kernel's deal)
    int range, group;
    if (!lvs_rq_item(blockIdx.x, a.groups, a.nparts, range, group)) return;
    const int qbase = group * 256;
    const _Float16* xq = (const _Float16*)a.xq;
    const char* xb = (const char*)a.xb;
    const long long ldb2 = a.ldb * 2;

    // ---- this wave's 64 queries -> registers: query block qb * 4 + wave of the group, fragment j; lane l = query l & 31, halfs (l >> 5) * 8
    half8 bv[G::NB_VGPR > 0 ? G::NB_VGPR : 1];
    auto qidx_of = [&](int qb) { return qbase + (qb * 4 + wave) * 32 + (lane & 31); };  // (recomputed where needed: registers)
    float tauf[2], qnv[2];
    uint32_t pub[2];  // the k-th key's score (ordered) this lane last saw published for its query
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = qidx_of(qb);
        const bool valid = qi < a.nq;
        const int qrow = valid ? qi : a.nq - 1;
        const _Float16* qp = xq + (long long)qrow * a.ldq + (lane >> 5) * 8;
        static_for<NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (qb == 0) {
                if constexpr (j < G::NB_AGPR) rj_load_b<(j < G::NB_AGPR ? j : 0)>(qp + j * 16);
                else bv[j - G::NB_AGPR < 0 ? 0 : j - G::NB_AGPR] = *(const half8*)(qp + j * 16);
            } else {
                if constexpr (NJ + j < G::NB_AGPR) rj_load_b<(NJ + j < G::NB_AGPR ? NJ + j : 0)>(qp + j * 16);
                else bv[NJ + j - G::NB_AGPR < 0 ? 0 : NJ + j - G::NB_AGPR] = *(const half8*)(qp + j * 16);
            }
        });
        pub[qb] = (valid && !RANGE) ? a.gtau[qi] : 0u;
        tauf[qb] = valid ? (RANGE ? a.threshold : rj_tau_float(pub[qb])) : INFINITY;  // lanes without a query never hold a candidate
        qnv[qb] = L2 ? a.qn[qrow] : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory", RJ_CLOBBER_AGPRS);  // (the loads into named registers are asm: nobody else waits for them)
    if constexpr (!RANGE)
        for (int i = lane; i < 64 * KCAP; i += 64) mylists[i] = 0;

    // ---- corpus range of this workgroup, in WHOLE 32-row blocks (the caller searches the last nb % 32 rows elsewhere)
    const long long nblocks = a.nb / 32;
    const long long b0 = (long long)range * a.blocks_per_wg;
    const long long b1 = b0 + a.blocks_per_wg < nblocks ? b0 + a.blocks_per_wg : nblocks;
    if (b0 >= b1) {  // an empty range still owns its slice of the output
        const int q1 = a.nq < qbase + 256 ? a.nq : qbase + 256;
        if constexpr (!RANGE)
            for (int i = qbase * k + tid; i < q1 * k; i += 256) a.out[(long long)range * a.nq * k + i] = 0;
        return;
    }
    const int nblk = (int)(b1 - b0);
    const int total_units = nblk * U;

    // ---- staging: load m of a unit fills LDS bytes [m * 1024, + 1024) of the unit's slot, lane l the 16-byte granule
    // g = m * 64 + l = (row, piece) with 2 UK + 1 granules per padded row (the last one is the pad: it re-reads the row's last piece)
    auto src_off = [&](int m, int ln) {
        const int g = m * 64 + ln;
        int row = g / (2 * UK + 1);
        int pc = g - row * (2 * UK + 1);
        pc = pc < 2 * UK ? pc : 2 * UK - 1;
        row = row < 31 ? row : 31;  // (granules past the unit's end: lanes 32 .. 63 of the last load)
        return (unsigned)(row * (int)ldb2 + pc * 16);
    };
    unsigned soff[LPW + 1];
#pragma unroll
    for (int i = 0; i < LPW; ++i) soff[i] = src_off(wave * LPW + i, lane);
    soff[LPW] = src_off(UK, lane);
    // the staging cursor (wave-uniform: scalar registers): the unit issued next
    const char* is_src = xb + b0 * 32 * ldb2;
    int is_n = 0, is_kh = 0, is_blk = 0, is_slot = 0;
    auto issue_piece = [&](auto pc) {
        constexpr int i = decltype(pc)::value;
        const unsigned dst = (unsigned)(unsigned long long)ring + (unsigned)(is_slot * G::UB);
        if constexpr (i < LPW) {
            rj_glds16(is_src, soff[i], dst + (unsigned)((wave * LPW + i) * 1024));
        } else {
            if ((is_n & 3) == wave) rj_glds16(is_src, soff[LPW], dst + (unsigned)(UK * 1024));
            if (!RANGE && is_kh == 0 && is_n < total_units) {
                // the block's side words, one 4-byte DMA per query block: lanes 0 .. 31 -> |y|^2 of its rows (any valid word under
                // inner product), lanes 32 .. 63 -> the shared thresholds of this wave's queries as the other workgroups left them
                const long long row0 = (b0 + is_blk) * 32;
                // (inner product: ONE 4-byte DMA - words 0 .. 31 the thresholds of query block 0, words 32 .. 63 those of query
                // block 1; squared L2: two - |y|^2 | thresholds 0, then thresholds 1 in the upper half of the second)
                if constexpr (L2) {
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        const int q = qidx_of(qb) < a.nq ? qidx_of(qb) : 0;
                        const void* p = lane < 32 ? (const void*)(a.bn + row0 + (lane & 31)) : (const void*)(a.gtau + q);
                        rj_glds4(p, side + ((is_blk % NB_RING) * 2 + qb) * 64);
                    }
                } else {
                    const int qi = qidx_of(lane >> 5);
                    rj_glds4(a.gtau + (qi < a.nq ? qi : 0), side + ((is_blk % NB_RING) * 2) * 64);
                }
            }
            // advance the cursor; past the range's end the last unit is loaded again (into a free slot, never read)
            ++is_n;
            if (is_n < total_units) {
                if (++is_kh == U) {
                    is_kh = 0;
                    ++is_blk;
                    is_src += 32 * ldb2 - (long long)(U - 1) * UK * 32;
                } else {
                    is_src += UK * 32;
                }
            }
            is_slot = is_slot + 1 == RING ? 0 : is_slot + 1;
        }
    };
    // ---- fragment reads: lane (r = l & 31, h = l >> 5) reads bytes [(2 jj + h) * 16, + 16) of padded row r: lane base + jj * 32
    unsigned o_base = (unsigned)(unsigned long long)ring + (unsigned)((lane & 31) * G::ROWB + (lane >> 5) * 16);
    int slot = 0;
    f32x16 acc[2];
    const unsigned side_lds = (unsigned)(unsigned long long)side + (unsigned)((lane & 31) * 4);

    // ---- deferred insertion: candidates wait in the wave's buffer; `drain` inserts them (wave-cooperative sorted insertion: lane
    // j < k owns slot j of the query's list - no lock, the list is this wave's), tightens the lanes' thresholds and publishes
    int count = 0;  // wave-uniform
    [[maybe_unused]] unsigned long long d_visit = 0, d_drain = 0, d_bar = 0, d_nvisit = 0, d_ndrain = 0, d_ncand = 0, d_epi = 0;  // (tuning builds: DBG = 32)
    auto drain = [&]() {
        const unsigned long long t_d0 = (DBG & 32) ? __builtin_amdgcn_s_memtime() : 0;
        d_ndrain += 1;
        d_ncand += count;
        uint32_t best[2] = {0u, 0u};
        const int grp = lane >> 4, slot16 = lane & 15;
        for (int i0 = 0; i0 < count; i0 += 64) {
            // every lane fetches ONE waiting entry (a single LDS round trip for up to 64 of them); then FOUR entries per round:
            // the 16 lanes of group g own the list slots of entry i + g's query (k <= 16), so one list read + write serves four
            // insertions - unless two of the four hit the same query, which then take their turns
            const int n = count - i0 < 64 ? count - i0 : 64;
            u64 ek = 0;
            uint32_t eq = 0xFFFFFFFFu;
            if (lane < n) {
                ek = ckey[i0 + lane];
                eq = cql[i0 + lane];
            }
            if constexpr (L2) {  // (squared L2 has no registers to spare for the four-at-a-time form: one entry per round)
                for (int i = 0; i < n; ++i) {
                    const u64 ukey = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ek >> 32), i) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ek, i);
                    const int uq = __builtin_amdgcn_readlane((int)eq, i);
                    u64* UL = mylists + uq * KCAP;
                    u64 mine = 0;
                    if (lane < k) mine = UL[lane];
                    u64 prev = ((u64)(uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)(mine >> 32), 0x138, 0xf, 0xf, false) << 32) |
                               (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)mine, 0x138, 0xf, 0xf, false);
                    if (lane == 0) prev = ~0ull;
                    u64 newv = 0;
                    if (lane < k) newv = mine > ukey ? mine : (prev > ukey ? ukey : prev);
                    if (lane < k) UL[lane] = newv;
                    const uint32_t ntau = __builtin_amdgcn_readlane((uint32_t)(newv >> 32), k - 1);
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb)
                        if (qb * 32 + (lane & 31) == uq) best[qb] = ntau > best[qb] ? ntau : best[qb];
                }
            } else
            for (int i = 0; i < n; i += 4) {
                uint32_t sq[4];  // (scalar) the four entries' queries; 0xFFFFFFFF = no entry
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) sq[gg] = i + gg < n ? (uint32_t)__builtin_amdgcn_readlane((int)eq, (i + gg) & 63) : 0xFFFFFFFFu;
                const bool clash = (sq[0] == sq[1] && sq[1] != 0xFFFFFFFFu) || (sq[0] == sq[2] && sq[2] != 0xFFFFFFFFu) ||
                                   (sq[0] == sq[3] && sq[3] != 0xFFFFFFFFu) || (sq[1] == sq[2] && sq[2] != 0xFFFFFFFFu) ||
                                   (sq[1] == sq[3] && sq[3] != 0xFFFFFFFFu) || (sq[2] == sq[3] && sq[3] != 0xFFFFFFFFu);
                const int e = (i + grp) & 63;
                const u64 ukey = ((u64)(uint32_t)__shfl((int)(uint32_t)(ek >> 32), e, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)ek, e, 64);
                const uint32_t uq = grp == 0 ? sq[0] : (grp == 1 ? sq[1] : (grp == 2 ? sq[2] : sq[3]));
                for (int turn = 0; turn < (clash ? 4 : 1); ++turn) {
                    const bool act = uq != 0xFFFFFFFFu && slot16 < k && (!clash || grp == turn);
                    u64* UL = mylists + (act ? uq : 0u) * KCAP;
                    u64 mine = 0;
                    if (act) mine = UL[slot16];
                    // the neighbour slot through a row shift instead of a second LDS read
                    u64 prev = ((u64)(uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)(mine >> 32), 0x111, 0xf, 0xf, false) << 32) |
                               (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)mine, 0x111, 0xf, 0xf, false);
                    if (slot16 == 0) prev = ~0ull;
                    const u64 newv = mine > ukey ? mine : (prev > ukey ? ukey : prev);
                    if (act) UL[slot16] = newv;
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) {
                        if (sq[gg] == 0xFFFFFFFFu || (clash && gg != turn)) continue;  // (wave-uniform)
                        const uint32_t ntau = __builtin_amdgcn_readlane((uint32_t)(newv >> 32), gg * 16 + k - 1);
#pragma unroll
                        for (int qb = 0; qb < 2; ++qb)
                            if ((uint32_t)(qb * 32 + (lane & 31)) == sq[gg]) best[qb] = ntau > best[qb] ? ntau : best[qb];
                    }
                }
            }
        }
        count = 0;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (best[qb] > pub[qb]) {  // a full list's k-th key is published to the other workgroups (fire and forget)
                tauf[qb] = fmaxf(tauf[qb], rj_tau_float(best[qb]));
                pub[qb] = best[qb];
                if (lane < 32 && !(DBG & 512)) atomicMax(&a.gtau[qidx_of(qb)], best[qb]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the unit loop counts its fragment reads from zero
        if (DBG & 32) d_drain += __builtin_amdgcn_s_memtime() - t_d0;
    };
    // what the other workgroups have found meanwhile (a lower bound of the k-th best over ALL rows), as it rode in with the block
    auto take_shared = [&](int qb, uint32_t g) {
        if (g > pub[qb]) {
            pub[qb] = g;
            tauf[qb] = fmaxf(tauf[qb], rj_tau_float(g));
        }
    };
    // The lanes whose scores of query block qb reach their thresholds append them to the wave's buffer; lane holds query
    // qidx_of(qb), rows row0 + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    auto append = [&](const f32x16& v, int qb, long long row0) {
        // all sixteen "which lanes hold a candidate in register r?" masks first, back to back: a compare whose result steers a
        // scalar branch costs ~45 cycles of VALU -> SALU latency, and one per register (the obvious loop) made a visit ~1 400
        // cycles on a wave that has nobody to hide behind (stamps: tools/rj_ablate.py).  ONE drain site behind the loop (a full
        // buffer: hundreds of equal scores in a block) - inlined at every register it made the visit 25 KB of code, walked once
        // per visit straight out of a cold instruction cache.
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {  // (four masks at a time, back to back: see above)
            u64 m[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = __ballot(v[g4 * 4 + e] >= tauf[qb]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = g4 * 4 + e;
                if (m[e] == 0) continue;
                if (count + __popcll(m[e]) > RJ_CAP) {  // (only with hundreds of equal scores in a block)
                    drain();
                    m[e] = __ballot(v[r] >= tauf[qb]);
                    if (m[e] == 0) continue;
                }
                const int pos = count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m[e] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[e], 0u));
                if ((m[e] >> lane) & 1) {
                    const long long row = row0 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
                    ckey[pos] = lvs_pack_key(v[r], (uint32_t)(row + a.id_offset));
                    cql[pos] = (uint32_t)(qb * 32 + (lane & 31));
                }
                count += __popcll(m[e]);
            }
        }
    };
    // every wave drains at the SAME blocks: a drain (hundreds of cycles per candidate) holds up the other three waves at the next
    // barrier, so four drains at four different moments cost the workgroup four times what four at once do
    auto block_end = [&](int blk) {
        if constexpr (DBG & 128) {
            if ((blk & (a.drain_every - 1)) == a.drain_every - 1) count = 0;
        } else {
            if (count > 0 && (count >= RJ_CAP - 64 || (blk & (a.drain_every - 1)) == a.drain_every - 1)) drain();
        }
    };
    // ---- block epilogue: 32 rows x this wave's 64 queries.  With one wave per SIMD nothing hides it, so as much of it as
    // possible has already happened inside the block's last MFMA steps (see `unit`): the side words are in registers (gsh) and the
    // thresholds the other workgroups published are taken in.  Left for here: the wait states behind the last MFMA, the lanes'
    // maxima, "does any lane hold a candidate?" - and, rarely, the visit.
    unsigned gsh[2] = {0u, 0u};
    auto epilogue = [&](int blk) {
        const long long row0 = (b0 + blk) * 32;
        const unsigned long long t_e0 = (DBG & 32) ? __builtin_amdgcn_s_memtime() : 0;
        // (inline-asm MFMAs are invisible to hipcc's hazard recogniser: the wait states between the block's last MFMA and the
        // first VALU read of its result are spelled out)
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]));
        if constexpr (L2) {
            const float* bnb = side + (blk % NB_RING) * 2 * 64 + 4 * (lane >> 5);  // |y|^2 of this lane's 16 rows
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 bn4 = *(const f32x4*)(bnb + 8 * m);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb)
                        acc[qb][m * 4 + e] = -fmaxf((qnv[qb] + bn4[e]) - 2.0f * acc[qb][m * 4 + e], 0.f);
            }
        }
        if constexpr (RANGE) {
            // strict ">" as the reference compares (sem_dedup.py:46); a self-join keeps the pairs with row id > query row
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if (!__any(rj_max16(acc[qb]) > tauf[qb])) continue;
                const long long qg = qidx_of(qb);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    u64 m[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = __ballot(acc[qb][g4 * 4 + e] > tauf[qb]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (m[e] == 0) continue;
                        const int r = g4 * 4 + e;
                        const long long jg = row0 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2) + a.id_offset;
                        if (((m[e] >> lane) & 1) && !(a.q_row0 >= 0 && jg <= a.q_row0 + qg)) {
                            const unsigned long long pos = atomicAdd(a.pair_count, 1ull);  // (keeps counting past the capacity)
                            if ((long long)pos < a.pair_capacity) {
                                a.pair_q[pos] = qg + a.q_base;
                                a.pair_j[pos] = jg;
                                a.pair_s[pos] = acc[qb][r] * a.out_scale;
                            }
                        }
                    }
                }
            }
        } else if constexpr (!(DBG & 4)) {
            bool wrote = false;
            const unsigned long long t_v0 = (DBG & 32) ? __builtin_amdgcn_s_memtime() : 0;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
                if (__any(rj_max16(acc[qb]) >= tauf[qb])) {
                    if constexpr (!(DBG & 64)) {
                        append(acc[qb], qb, row0);
                        wrote = true;
                    } else {
                        count += 1;
                    }
                }
            // (LDS writes of its own only behind a visit: the unit loop counts its fragment reads from zero)
            if (wrote) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if ((DBG & 32) && wrote) {
                d_visit += __builtin_amdgcn_s_memtime() - t_v0;
                d_nvisit += 1;
            }
        } else {
            if (__any(rj_max16(acc[0]) >= 3.0e38f) || __any(rj_max16(acc[1]) >= 3.0e38f)) count = 1;
        }
        if constexpr (!RANGE) block_end(blk);
        if (DBG & 32) d_epi += __builtin_amdgcn_s_memtime() - t_e0;
    };

    // One unit of block `blk`.  In the block's LAST unit the epilogue's memory part rides along: step UK - 7 reads the block's side
    // words (ONE more LDS read in flight: the counted waits of the next three steps allow for it), steps UK - 3 / UK - 2 take the
    // thresholds in.
    auto unit = [&](auto khc, int blk) {
        constexpr int kh = decltype(khc)::value;
        constexpr bool F = kh == U - 1 && !(DBG & 2) && !(DBG & 256) && !RANGE;
        constexpr int JS = UK - 7;  // the step that reads the side words
        const unsigned o_next = o_base + (slot + 1 == RING ? (unsigned)(-(RING - 1) * G::UB) : (unsigned)G::UB);
        static_for<UK>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            constexpr int ahead = jj + AD - 1;
            if constexpr (ahead < UK) rj_read_a<ahead % AD, (ahead < UK ? ahead : 0) * 32>(o_base);
            else rj_read_a<ahead % AD, (ahead >= UK ? ahead - UK : 0) * 32>(o_next);  // the next unit's slot: certified at this unit's barrier
            if constexpr (F && jj == JS)
            {
                if constexpr (L2)
                    asm volatile("ds_read2_b32 %0, %1 offset0:32 offset1:96" : "=v"(*(unsigned long long*)gsh) : "v"(side_lds + (unsigned)((blk % NB_RING) * 512)) : "memory");
                else
                    asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:32" : "=v"(*(unsigned long long*)gsh) : "v"(side_lds + (unsigned)((blk % NB_RING) * 512)) : "memory");
            }
            constexpr int piece = rj_piece_at(jj, UK, LPW);
            if constexpr (piece >= 0 && !(DBG & 1)) issue_piece(std::integral_constant<int, piece>{});
            __builtin_amdgcn_sched_barrier(0);
            constexpr int extra = (F && jj >= JS && jj < JS + AD - 1) ? 1 : 0;
            if constexpr (F && jj == JS + AD - 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(*(unsigned long long*)gsh) : "n"(AD - 1) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(AD - 1 + extra) : "memory");
            constexpr bool zero = kh == 0 && jj == 0;
            constexpr int f0 = kh * UK + jj, f1 = NJ + kh * UK + jj;
            if constexpr (f0 < G::NB_AGPR) rj_mfma_aa<jj % AD, (f0 < G::NB_AGPR ? f0 : 0), zero>(acc[0]);
            else rj_mfma_av<jj % AD, zero>(acc[0], bv[f0 - G::NB_AGPR < 0 ? 0 : f0 - G::NB_AGPR]);
            if constexpr (f1 < G::NB_AGPR) rj_mfma_aa<jj % AD, (f1 < G::NB_AGPR ? f1 : 0), zero>(acc[1]);
            else rj_mfma_av<jj % AD, zero>(acc[1], bv[f1 - G::NB_AGPR < 0 ? 0 : f1 - G::NB_AGPR]);
            if constexpr (F && jj == UK - 3) take_shared(0, gsh[0]);
            if constexpr (F && jj == UK - 2) take_shared(1, gsh[1]);
            __builtin_amdgcn_sched_barrier(0);
        });
        slot = slot + 1 == RING ? 0 : slot + 1;
        o_base = o_next;
    };

    // ---- prologue: RING - 1 units in flight, unit 0 certified by a barrier of its own, its first fragments read
    for (int n = 0; n < RING - 1; ++n) static_for<LPW + 1>([&](auto pc) { issue_piece(pc); });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * LPW) : "memory");
    __builtin_amdgcn_s_barrier();
    static_for<AD - 1>([&](auto jc) {
        constexpr int jj = decltype(jc)::value;
        rj_read_a<jj % AD, jj * 32>(o_base);
    });
    for (int blk = 0; blk < nblk; ++blk) {
        static_for<U>([&](auto khc) {
            // unit n + 1 has landed once this wave's loads of the RING - 3 younger units are all that is in flight (the half-used
            // load and the side words ride in the same queue: the wait is only more conservative); the barrier makes that true
            // for every wave's share and tells everybody that unit n - 1's slot is free again
            const unsigned long long t_b0 = (DBG & 32) ? __builtin_amdgcn_s_memtime() : 0;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 3) * LPW) : "memory");
            if constexpr (!(DBG & 16)) __builtin_amdgcn_s_barrier();
            if (DBG & 32) d_bar += __builtin_amdgcn_s_memtime() - t_b0;
            unit(khc, blk);
        });
        if constexpr (!(DBG & 2)) epilogue(blk);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the clamped tail loads still target the ring; the last fragment reads
    if constexpr (RANGE) return;
    drain();
#ifdef LVS_TUNING
    if ((DBG & 32) && lane == 0) {
        atomicAdd(&rj_dbg[0], d_visit);
        atomicAdd(&rj_dbg[1], d_drain);
        atomicAdd(&rj_dbg[2], d_bar);
        atomicAdd(&rj_dbg[3], d_nvisit);
        atomicAdd(&rj_dbg[4], d_ndrain);
        atomicAdd(&rj_dbg[5], d_ncand);
        atomicAdd(&rj_dbg[6], d_epi);
        atomicAdd(&rj_dbg[7], (unsigned long long)nblk);
        atomicAdd(&rj_dbg[8], 1ull);
    }
#endif
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 64 * k; i += 64) {
        const int ql = i / k, j = i - ql * k;
        const int q = qbase + ((ql >> 5) * 4 + wave) * 32 + (ql & 31);
        if (q < a.nq) a.out[((long long)range * a.nq + q) * k + j] = mylists[ql * KCAP + j];
    }
    if (lane < 32) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const uint32_t lo = (uint32_t)(mylists[(qb * 32 + lane) * KCAP + k - 1] >> 32);
            if (lo && qidx_of(qb) < a.nq) atomicMax(&a.gtau[qidx_of(qb)], lo);
        }
    }
}

template <int NJ, int KCAP, bool L2, int DBG = 0, bool RANGE = false>
hipError_t rj_launch_k(const LvsRqArgs& a, int grid, hipStream_t stream) {
    using G = RjGeom<NJ>;
    const size_t lds = (size_t)G::lds_bytes(G::ring(KCAP), KCAP);
    static LvsPerDeviceOnce attr;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!attr.done(dev, lds)) {
        e = hipFuncSetAttribute((const void*)lvs_rj_kernel<NJ, KCAP, L2, DBG, RANGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr.set(dev, lds);
    }
    hipLaunchKernelGGL((lvs_rj_kernel<NJ, KCAP, L2, DBG, RANGE>), dim3(grid), dim3(256), lds, stream, a);
    return hipGetLastError();
}
template <int NJ>
hipError_t rj_launch_nj(const LvsRqArgs& a, int grid, hipStream_t stream) {
    const bool l2 = a.metric == LVS_METRIC_L2;
#ifdef LVS_TUNING
    if (NJ == 48 && !l2 && a.k <= 10) {  // timing ablations (LVS_RQ_DEBUG; WRONG results): 1 no staging loads, 2 no epilogue, 4 + filter only, 16 no barrier
        switch ((int)lvs_tune("LVS_RQ_DEBUG", 0)) {
            case 1: return rj_launch_k<NJ, 10, false, 1>(a, grid, stream);
            case 2: return rj_launch_k<NJ, 10, false, 2>(a, grid, stream);
            case 3: return rj_launch_k<NJ, 10, false, 3>(a, grid, stream);
            case 6: return rj_launch_k<NJ, 10, false, 6>(a, grid, stream);
            case 18: return rj_launch_k<NJ, 10, false, 18>(a, grid, stream);
            case 19: return rj_launch_k<NJ, 10, false, 19>(a, grid, stream);
            case 32: return rj_launch_k<NJ, 10, false, 32>(a, grid, stream);
            case 64: return rj_launch_k<NJ, 10, false, 64>(a, grid, stream);
            case 512: return rj_launch_k<NJ, 10, false, 512>(a, grid, stream);
            case 128: return rj_launch_k<NJ, 10, false, 128>(a, grid, stream);
            case 192: return rj_launch_k<NJ, 10, false, 192>(a, grid, stream);
            case 448: return rj_launch_k<NJ, 10, false, 448>(a, grid, stream);
            default: break;
        }
    }
#endif
    if (a.k <= 10) return l2 ? rj_launch_k<NJ, 10, true>(a, grid, stream) : rj_launch_k<NJ, 10, false>(a, grid, stream);
    return l2 ? rj_launch_k<NJ, LVS_RQ_KMAX, true>(a, grid, stream) : rj_launch_k<NJ, LVS_RQ_KMAX, false>(a, grid, stream);
}

}  // namespace

#ifdef LVS_TUNING
// tuning builds: read and clear the DBG = 32 counters (visit / drain / barrier cycles, visits, drains, candidates, epilogue cycles,
// blocks, waves)
extern "C" int32_t lvs_rj_debug_read(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rj_dbg), sizeof(unsigned long long) * 16) != hipSuccess) return LVS_EDEVICE;
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(rj_dbg), z, sizeof(z)) != hipSuccess) return LVS_EDEVICE;
    return LVS_OK;
}
#endif

// Does the one-wave-per-SIMD form take this launch?  Everything lvs_rq_kernel's grouped launch takes (lvs_rq_fits) with row ids
// that are positions (no id table: a candidate's key is built where it is found) and at least one whole 32-row block per range.
bool lvs_rj_fits(int64_t nq, int64_t nb, int dpad, int k, bool has_row_ids) {
    if (has_row_ids || !(dpad == 256 || dpad == 384 || dpad == 512 || dpad == 768) || nq <= 128 || k < 1 || k > LVS_RQ_KMAX) return false;
    if (nq <= LVS_RQ_MAXQ) return lvs_rq_fits(nq, nb, dpad, k) || (nq > 2048 && nb >= LVS_RQ_JOIN_MINROWS);  // (the last chunk of a call)
    const int64_t groups = (nq + LVS_RQ_GROUPQ - 1) / LVS_RQ_GROUPQ;  // a chunk of a larger call: 32 x 2^i groups
    return nq <= LVS_RQ_CHUNK_MAX && groups % 32 == 0 && nb >= LVS_RQ_JOIN_MINROWS;
}

static hipError_t rj_launch_any(LvsRqArgs& a, int dpad, hipStream_t stream, bool range_mode);
// a.nb rows are searched in whole 32-row blocks: the caller runs the last a.nb % 32 rows through lvs_rq_launch (one more list per
// query).  On return a.nparts = candidate lists per query in a.out ([nparts][nq][k]).
hipError_t lvs_rj_launch(LvsRqArgs& a, int dpad, hipStream_t stream) { return rj_launch_any(a, dpad, stream, false); }
// The threshold join on the same launch geometry (whole 32-row blocks; inner product; a.pair_* / a.threshold / a.q_row0 set)
hipError_t lvs_rj_range_launch(LvsRqArgs& a, int dpad, hipStream_t stream) { return rj_launch_any(a, dpad, stream, true); }
static hipError_t rj_launch_any(LvsRqArgs& a, int dpad, hipStream_t stream, bool range_mode) {
    const int64_t nblocks = a.nb / 32;
    a.groups = (a.nq + LVS_RQ_GROUPQ - 1) / LVS_RQ_GROUPQ;
    if (a.groups < 1) a.groups = 1;
    if (a.groups > 32 && a.groups % 32 != 0) return hipErrorInvalidValue;
    int64_t ranges = lvs_rq_ranges_for(a.groups);
    if (ranges < 1 || nblocks < 1) return hipErrorInvalidValue;
    a.drain_every = (int)lvs_tune("LVS_RJ_EVERY", RJ_DRAIN_EVERY);
    if (a.drain_every < 1 || (a.drain_every & (a.drain_every - 1))) a.drain_every = RJ_DRAIN_EVERY;
    if (ranges > (nblocks + 3) / 4) ranges = (nblocks + 3) / 4;
    if (ranges < 1) ranges = 1;
    a.blocks_per_wg = (int)((nblocks + ranges - 1) / ranges);
    ranges = (nblocks + a.blocks_per_wg - 1) / a.blocks_per_wg;
    a.nparts = (int)ranges;
    const int grid = lvs_rq_grid(a.groups, (int)ranges);
    if (range_mode) {
        if (a.metric != LVS_METRIC_IP) return hipErrorInvalidValue;
        switch (dpad / 16) {
            case 48: return rj_launch_k<48, 10, false, 0, true>(a, grid, stream);
            case 32: return rj_launch_k<32, 10, false, 0, true>(a, grid, stream);
            case 24: return rj_launch_k<24, 10, false, 0, true>(a, grid, stream);
            case 16: return rj_launch_k<16, 10, false, 0, true>(a, grid, stream);
            default: return hipErrorInvalidValue;
        }
    }
    switch (dpad / 16) {
        case 48: return rj_launch_nj<48>(a, grid, stream);
        case 32: return rj_launch_nj<32>(a, grid, stream);
        case 24: return rj_launch_nj<24>(a, grid, stream);
        case 16: return rj_launch_nj<16>(a, grid, stream);
        default: return hipErrorInvalidValue;
    }
}

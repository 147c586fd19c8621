// The asm-scheduled K-step shared by the tile kernels (lvs_tile.hip: list / range / score epilogues, corpus streaming;
// lvs_assign.hip: nearest-row search with the queries streaming) - one 64-half K-step of a 256 x BQ score tile on the
// matrix cores: LDS fragment reads issued in a fixed order with counted lgkmcnt waits, the staging loads of the NEXT
// K-step interleaved with the first MFMA pairs.  Tuning history: profiles/r01_tuning.md, r02_tuning.md.
#pragma once
#include <type_traits>
#include <utility>

#include "lvs_common.h"

namespace lvs_kstep {

constexpr int BC = 256, BK = 64;
constexpr int ROWB = BK * 2;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ inline void glds16(const void* gsrc, void* ldst) {
    __builtin_amdgcn_global_load_lds((gbl_void_t*)gsrc, (lds_void_t*)ldst, 16, 0, 0);
}

// ---- asm-scheduled K-step ------------------------------------------------------------------------------------
// hipcc sinks every LDS fragment read to just before its first use and waits with lgkmcnt(0), which undoes the
// software pipelining written in the source.  The fragment reads and their waits are therefore inline asm: reads are
// issued in source order (pinned by sched_barrier), waits are COUNTED (LDS returns a wave's reads in issue order).
template <int... Is, class F>
__device__ inline void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ inline void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
template <int OFFSET>  // OFFSET: compile-time byte offset (16-bit immediate field of the instruction)
__device__ inline void lds_read16(half8& dst, unsigned addr) {
    static_assert(OFFSET >= 0 && OFFSET < 65536, "ds_read offset field is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFFSET));
}
template <int N>
__device__ inline void lds_wait(half8& a, half8& b0, half8& b1) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b0), "+v"(b1) : "n"(N));
}
// Issue order of one K-step (4 * MI steps f = kk * MI + mi): prologue B(0)[0], B(0)[1], A(0) .. A(DEPTH-1); step f issues
// A(f + DEPTH) (if any), then B(kk+1)[0..1] when mi == BPOS.  wait(f) = reads that may still be outstanding when step
// f's MFMAs start = (reads issued so far) - 1 - (issue index of the last read step f needs).
template <int MI, int DEPTH, int BPOS>
struct KStepOrder {
    static constexpr int NF = 4 * MI;
    static constexpr int issued(int f) {
        int c = 2 + DEPTH;
        for (int g = 0; g <= f; ++g) {
            if (g + DEPTH < NF) ++c;
            if (g % MI == BPOS && g / MI + 1 < 4) c += 2;
        }
        return c;
    }
    static constexpr int pos_A(int f) { return f < DEPTH ? 2 + f : issued(f - DEPTH - 1); }
    static constexpr int pos_B(int kk) { return kk == 0 ? 1 : issued((kk - 1) * MI + BPOS) - 1; }
    static constexpr int wait(int f) {
        const int a = pos_A(f), b = pos_B(f / MI);
        return issued(f) - 1 - (a > b ? a : b);
    }
};

// One K-step of the workgroup's 256 x (64 MI) score tile.  sb: the staged operands of THIS K-step (corpus rows at 0,
// query rows at BC * ROWB); n_base: the other staging buffer, filled meanwhile with the NEXT K-step's operands from
// c_sbase / q_sbase (wave-uniform 64-bit bases) + the per-lane byte offsets c_loff[4] / q_loff[QG]; foff: per-lane fragment
// offsets of the four 16-wide k-slices.  A fragments are read KDEPTH = 2 steps ahead through a 3-deep register ring, B
// fragments double-buffered per k-slice.
template <int MI, int QG>
__device__ __forceinline__ void run(const char* sb, char* n_base, const char* c_sbase, const char* q_sbase,
                                    const unsigned (&c_loff)[4], const unsigned (&q_loff)[QG], int wave, int a_base, int b_base,
                                    const int (&foff)[4], f32x16 (&acc)[MI][2]) {
    constexpr int NF = 4 * MI;
    constexpr int KDEPTH = 2, KBPOS = MI == 4 ? 2 : 0;
    using Ord = KStepOrder<MI, KDEPTH, KBPOS>;
    half8 Bf[2][2], Af[KDEPTH + 1];
    // one address VGPR per (operand, kk); the 32-row block (mi / second query block) goes into the offset field
    const unsigned sbu = (unsigned)(unsigned long long)sb;
    unsigned a_addr[4], b_addr[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        a_addr[kk] = (sbu + a_base) + foff[kk];
        b_addr[kk] = (sbu + b_base) + foff[kk];
    }
    lds_read16<0>(Bf[0][0], b_addr[0]);
    lds_read16<32 * ROWB>(Bf[0][1], b_addr[0]);
    static_for<KDEPTH>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        lds_read16<(f % MI) * 32 * ROWB>(Af[f], a_addr[f / MI]);
    });
    static_for<NF>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        constexpr int kk = f / MI, mi = f % MI;
        if constexpr (f + KDEPTH < NF) {
            constexpr int f2 = f + KDEPTH;
            lds_read16<(f2 % MI) * 32 * ROWB>(Af[f2 % (KDEPTH + 1)], a_addr[f2 / MI]);
        }
        if constexpr (mi == KBPOS && kk + 1 < 4) {
            lds_read16<0>(Bf[(kk + 1) & 1][0], b_addr[kk + 1]);
            lds_read16<32 * ROWB>(Bf[(kk + 1) & 1][1], b_addr[kk + 1]);
        }
        if constexpr (f < 4)
            glds16(c_sbase + c_loff[f], n_base + (wave * 32 + f * 8) * ROWB);
        else if constexpr (f < 4 + QG)
            glds16(q_sbase + q_loff[f - 4], n_base + BC * ROWB + (wave * (8 * QG) + (f - 4) * 8) * ROWB);
        __builtin_amdgcn_sched_barrier(0);
        lds_wait<Ord::wait(f)>(Af[f % (KDEPTH + 1)], Bf[kk & 1][0], Bf[kk & 1][1]);
        acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[f % (KDEPTH + 1)], Bf[kk & 1][0], acc[mi][0], 0, 0, 0);
        acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[f % (KDEPTH + 1)], Bf[kk & 1][1], acc[mi][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    });
}

}  // namespace lvs_kstep

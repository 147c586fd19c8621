// Design-space probe (development aid, not part of the library): the tile kernel's main loop WITHOUT any top-k, as a
// plain C[q][j] max-reduction GEMM, templated on the wave layout.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_gemm.hip -o /tmp/probe_gemm && /tmp/probe_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <utility>
#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)


template <int... Is, class F>
__device__ inline void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ inline void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

__device__ inline void lds_read16(half8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
template <int N>
__device__ inline void lds_wait5(half8& a, half8& b0, half8& b1, half8& b2, half8& b3) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "n"(N));
}
template <int N>
__device__ inline void lds_wait(half8& a, half8& b0, half8& b1) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b0), "+v"(b1) : "n"(N));
}
// issue-order bookkeeping for the asm-scheduled K-step (NI == 2): reads are issued in this order -
//   prologue: B(0)[0], B(0)[1], A(0) .. A(DEPTH-1);  step f: A(f+DEPTH) (if any), then B(kk+1)[0..1] when mi == BPOS.
// pos_A(f) / pos_B(kk): 0-based issue index of the LAST read that MFMA step f needs; issued(f): reads issued up to and
// including step f's own issues.  The counted wait before step f's MFMAs is lgkmcnt(issued(f) - 1 - needed).
template <int MI, int DEPTH, int BPOS, int NB = 2>
struct KStepOrder {
    static constexpr int NF = 4 * MI;
    static constexpr int issued(int f) {
        int c = NB + DEPTH;
        for (int g = 0; g <= f; ++g) {
            if (g + DEPTH < NF) ++c;
            if (g % MI == BPOS && g / MI + 1 < 4) c += NB;
        }
        return c;
    }
    static constexpr int pos_A(int f) {
        if (f < DEPTH) return NB + f;
        // issued at step g = f - DEPTH as the first read of that step
        return issued(f - DEPTH - 1 < 0 ? -1 : f - DEPTH - 1);
    }
    static constexpr int pos_B(int kk) {
        if (kk == 0) return NB - 1;
        const int g = (kk - 1) * MI + BPOS;  // step that issued B(kk): after that step's A read (if any)
        return issued(g) - 1;
    }
    static constexpr int wait(int f) {
        const int a = pos_A(f), b = pos_B(f / MI);
        const int need = a > b ? a : b;
        return issued(f) - 1 - need;
    }
};

constexpr int BC = 256, BQ = 256, BK = 64, ROWB = 128, STAGE = (BC + BQ) * ROWB;

__device__ inline void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)l, 16, 0, 0);
}

// WM x WN waves, each wave MI x NI accumulator blocks of 32x32:  WM*MI*32 == 256, WN*NI*32 == 256
template <int WM, int WN, int MI, int NI, int WPE, int DEPTH, int BPOS, int PRIO, int SPREAD, int STAMP = 0, int ABL = 0>
__global__ __launch_bounds__(WM * WN * 64, WPE) void gemm_probe(const _Float16* __restrict__ xb, const _Float16* __restrict__ xq,
                                                                 float* __restrict__ out, int ntiles, int nk, long long ld, unsigned long long* __restrict__ stamps, int qmod) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WM * WN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
        // XCD x = b % 8 runs 32 blocks at a time (one per CU): local index i = (b / 8) % 32, generation g = b / 256
    const int xcd = blockIdx.x & 7, li = (blockIdx.x >> 3) & 31, gen = blockIdx.x >> 8;
    const int gq = qmod;                       // query tiles per XCD group; 32 / gq corpus phases
    const int qt = (gen * 8 + xcd) * gq + (li % gq);
    const int slab = li / gq, nsl = 32 / gq;
    const int tile0 = slab * (ntiles / nsl);
    const long long q0 = (long long)qt * BQ;
    constexpr int RPW = 512 / NW;        // staged rows per wave per K-step (corpus + queries)
    constexpr int GL = RPW / 8;          // glds per wave per K-step
    // wave stages rows [wave*RPW, +RPW) of the 512-row (corpus | query) stack
    unsigned loff[GL];
    const char* sbase[GL];
#pragma unroll
    for (int i = 0; i < GL; ++i) {
        int row = wave * RPW + i * 8 + (lane >> 3);
        int col = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        bool isq = row >= BC;
        long long grow = isq ? q0 + (row - BC) : row;
        loff[i] = (unsigned)((grow * ld + col) * 2);
        sbase[i] = (const char*)(isq ? xq : xb);
    }
    int foff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        foff[kk] = (lane & 31) * ROWB + ((((kk * 2) + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    const int a_base = wm * MI * 32 * ROWB;
    const int b_base = BC * ROWB + wn * NI * 32 * ROWB;
    f32x16 acc[MI][NI];
    float best[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) best[ni] = -1e30f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int T = ntiles * nk;
    auto issue = [&](int t, int buf) {
        int ti = t / nk, ks = t - ti * nk; ti = (ti + tile0) % ntiles;
#pragma unroll
        for (int i = 0; i < GL; ++i) {
            int row = wave * RPW + i * 8;
            bool isq = row >= BC;
            long long tile_off = isq ? 0 : (long long)ti * BC * ld * 2;
            glds16(sbase[i] + tile_off + loff[i] + ks * BK * 2, smem + buf * STAGE + row * ROWB);
        }
    };
    issue(0, 0);
    int ksin = 0;
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        unsigned long long ta = 0, tb = 0;
        const bool st = STAMP && stamps && t >= 200 && t < 264 && (blockIdx.x == 0 || blockIdx.x == 1001);
        if (st) ta = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st) tb = __builtin_amdgcn_s_memtime();
        const char* sb = smem + buf * STAGE;
        const int tn = t + 1 < T ? t + 1 : T - 1;
        const int nti0 = tn / nk, nks = tn - nti0 * nk; const int nti = (nti0 + tile0) % ntiles;
        if constexpr (SPREAD == 4) {
            // one wave per SIMD: 4 waves of 128 x 128 (MI = NI = 4), asm-scheduled, 16 staging loads per wave and K-step
            static_assert(NI == 4 && MI == 4, "written for 4 x 4 accumulator blocks");
            using Ord = KStepOrder<MI, DEPTH, BPOS, 4>;
            half8 Bf[2][4], Af[DEPTH + 1];
            const unsigned sbu = (unsigned)(unsigned long long)sb;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) lds_read16(Bf[0][ni], sbu + b_base + ni * 32 * ROWB + foff[0]);
#pragma unroll
            for (int f = 0; f < DEPTH; ++f) lds_read16(Af[f], sbu + a_base + (f % MI) * 32 * ROWB + foff[f / MI]);
            static_for<16>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                constexpr int kk = f / 4, mi = f % 4;
                if constexpr (f + DEPTH < 16) {
                    constexpr int f2 = f + DEPTH;
                    lds_read16(Af[f2 % (DEPTH + 1)], sbu + a_base + (f2 % 4) * 32 * ROWB + foff[f2 / 4]);
                }
                if constexpr (mi == BPOS && kk + 1 < 4) {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) lds_read16(Bf[(kk + 1) & 1][ni], sbu + b_base + ni * 32 * ROWB + foff[kk + 1]);
                }
                if (f < GL) {
                    int row = wave * RPW + f * 8;
                    bool isq = row >= BC;
                    long long tile_off = isq ? 0 : (long long)nti * BC * ld * 2;
                    glds16(sbase[f] + tile_off + loff[f] + nks * BK * 2, smem + (buf ^ 1) * STAGE + row * ROWB);
                }
                __builtin_amdgcn_sched_barrier(0);
                lds_wait5<Ord::wait(f)>(Af[f % (DEPTH + 1)], Bf[kk & 1][0], Bf[kk & 1][1], Bf[kk & 1][2], Bf[kk & 1][3]);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[f % (DEPTH + 1)], Bf[kk & 1][ni], acc[mi][ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        } else if constexpr (SPREAD == 3) {
            // asm-scheduled K-step: LDS reads and their counted waits are inline asm, order pinned by sched_barrier
            static_assert(NI == 2, "asm K-step is written for NI == 2");
            using Ord = KStepOrder<MI, DEPTH, BPOS>;
            half8 Bf[2][2], Af[DEPTH + 1];
            if (PRIO == 1) { if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1); }
            const unsigned sbu = (unsigned)(unsigned long long)sb;
            lds_read16(Bf[0][0], sbu + b_base + foff[0]);
            lds_read16(Bf[0][1], sbu + b_base + 32 * ROWB + foff[0]);
            constexpr int NF = 4 * MI;
#pragma unroll
            for (int f = 0; f < DEPTH; ++f) lds_read16(Af[f], sbu + a_base + (f % MI) * 32 * ROWB + foff[f / MI]);
            static_for<NF>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                constexpr int kk = f / MI, mi = f % MI;
                if (f + DEPTH < NF && !(ABL & 1)) {
                    constexpr int f2 = f + DEPTH;
                    lds_read16(Af[f2 % (DEPTH + 1)], sbu + a_base + (f2 % MI) * 32 * ROWB + foff[f2 / MI]);
                }
                if (mi == BPOS && kk + 1 < 4 && !(ABL & 1)) {
                    lds_read16(Bf[(kk + 1) & 1][0], sbu + b_base + foff[kk + 1]);
                    lds_read16(Bf[(kk + 1) & 1][1], sbu + b_base + 32 * ROWB + foff[kk + 1]);
                }
                if (f < GL && !(ABL & 2)) {
                    int row = wave * RPW + f * 8;
                    bool isq = row >= BC;
                    long long tile_off = isq ? 0 : (long long)nti * BC * ld * 2;
                    glds16(sbase[f] + tile_off + loff[f] + nks * BK * 2, smem + (buf ^ 1) * STAGE + row * ROWB);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(ABL & 1)) lds_wait<Ord::wait(f)>(Af[f % (DEPTH + 1)], Bf[kk & 1][0], Bf[kk & 1][1]);
                acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[f % (DEPTH + 1)], Bf[kk & 1][0], acc[mi][0], 0, 0, 0);
                acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[f % (DEPTH + 1)], Bf[kk & 1][1], acc[mi][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
        half8 Bf[2][NI], Af[DEPTH + 1];
        if (PRIO == 1) { if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1); }  // static priority for the younger half
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) Bf[0][ni] = *(const half8*)(sb + b_base + ni * 32 * ROWB + foff[0]);
        constexpr int NF = 4 * MI;
#pragma unroll
        for (int f = 0; f < DEPTH; ++f) Af[f] = *(const half8*)(sb + a_base + (f % MI) * 32 * ROWB + foff[f / MI]);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int kk = f / MI, mi = f % MI;
            if (f + DEPTH < NF) {
                const int f2 = f + DEPTH;
                Af[f2 % (DEPTH + 1)] = *(const half8*)(sb + a_base + (f2 % MI) * 32 * ROWB + foff[f2 / MI]);
            }
            if (mi == BPOS && kk + 1 < 4) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) Bf[(kk + 1) & 1][ni] = *(const half8*)(sb + b_base + ni * 32 * ROWB + foff[kk + 1]);
            }
            if ((SPREAD == 1 ? ((f & 1) == 0 && (f >> 1) < GL) : (f < GL))) {
                const int gi = SPREAD == 1 ? (f >> 1) : f;
                int row = wave * RPW + gi * 8;
                bool isq = row >= BC;
                long long tile_off = isq ? 0 : (long long)nti * BC * ld * 2;
                glds16(sbase[gi] + tile_off + loff[gi] + nks * BK * 2, smem + (buf ^ 1) * STAGE + row * ROWB);
            }
            if (SPREAD == 2) __builtin_amdgcn_sched_barrier(0);  // pin: this step's reads are issued BEFORE its MFMAs
            if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[f % (DEPTH + 1)], Bf[kk & 1][ni], acc[mi][ni], 0, 0, 0);
            if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
            if (SPREAD == 2) __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (st) {
            unsigned long long tc = __builtin_amdgcn_s_memtime();
            if (lane == 0) {
                unsigned long long* o = stamps + (((blockIdx.x ? 1 : 0) * NW + wave) * 64 + (t - 200)) * 3;
                o[0] = ta; o[1] = tb; o[2] = tc;
            }
        }
        if (++ksin < nk) continue;
        ksin = 0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int r = 0; r < 16; ++r) best[ni] = fmaxf(best[ni], acc[mi][ni][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
            }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) out[((long long)blockIdx.x * NW + wave) * 64 * NI + ni * 64 + lane] = best[ni];
}


// ---- role-split ("ping-pong") K-step: the two waves of a SIMD alternate between a LOAD section (LDS fragment reads of
// the next quadrant + staging loads of the next K-step) and an MFMA section (one quadrant: 2 corpus blocks x 1 query block
// x K = 64 -> 8 MFMAs), separated by workgroup barriers; waves 4-7 run one section behind waves 0-3, so on every SIMD one
// wave computes while its partner loads.  Same tile geometry, LDS image and swizzle as the product kernel (2 x 4 waves,
// 128 x 64 per wave).  Quadrant order per K-step: (A01,B0) (A01,B1) (A23,B1) (A23,B0); B0 keeps its own registers, so a
// K-step reads 24 fragments as before.  VARIANT bit 0: s_setprio(1) around the MFMA sections; bit 1: no stagger (all
// waves in lockstep - the control); bit 2: staging loads in LOAD(0)/(1) instead of (1)/(2).
template <int VARIANT>
__global__ __launch_bounds__(512, 2) void gemm_pingpong(const _Float16* __restrict__ xb, const _Float16* __restrict__ xq,
                                                        float* __restrict__ out, int ntiles, int nk, long long ld, int qmod,
                                                        unsigned* __restrict__ simd_ids) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / 4, wn = wave % 4;
    const int grp = (VARIANT & 2) ? 0 : wm;  // waves 4-7 = second group
    if (simd_ids && blockIdx.x == 0 && lane == 0) simd_ids[wave] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
    const int xcd = blockIdx.x & 7, li = (blockIdx.x >> 3) & 31, gen = blockIdx.x >> 8;
    const int gq = qmod;
    const int qt = (gen * 8 + xcd) * gq + (li % gq);
    const int slab = li / gq, nsl = 32 / gq;
    const int tile0 = slab * (ntiles / nsl);
    const long long q0 = (long long)qt * BQ;
    constexpr int RPW = 64, GL = 8;  // staged rows / glds per wave per K-step
    unsigned loff[GL];
    const char* sbase[GL];
#pragma unroll
    for (int i = 0; i < GL; ++i) {
        int row = wave * RPW + i * 8 + (lane >> 3);
        int col = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        bool isq = row >= BC;
        long long grow = isq ? q0 + (row - BC) : row;
        loff[i] = (unsigned)((grow * ld + col) * 2);
        sbase[i] = (const char*)(isq ? xq : xb);
    }
    int foff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        foff[kk] = (lane & 31) * ROWB + ((((kk * 2) + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    const int a_base = wm * 4 * 32 * ROWB;
    const int b_base = BC * ROWB + wn * 2 * 32 * ROWB;
    f32x16 acc[4][2];
    float best[2] = {-1e30f, -1e30f};
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int T = ntiles * nk;
    auto glds_part = [&](int t, int buf, int i0, int i1) {  // staging loads i0..i1-1 of K-step t into buffer buf
        int ti = t / nk, ks = t - ti * nk;
        ti = (ti + tile0) % ntiles;
#pragma unroll
        for (int i = 0; i < GL; ++i) {
            if (i < i0 || i >= i1) continue;
            int row = wave * RPW + i * 8;
            bool isq = row >= BC;
            long long tile_off = isq ? 0 : (long long)ti * BC * ld * 2;
            glds16(sbase[i] + tile_off + loff[i] + ks * BK * 2, smem + buf * STAGE + row * ROWB);
        }
    };
    glds_part(0, 0, 0, 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();  // the second group runs one section behind
    half8 A[2][4], B0[4], B1[4];
    int ksin = 0;
    constexpr int G0 = (VARIANT & 4) ? 0 : 1;  // LOAD section that issues the first half of the staging loads
    for (int t = 0; t < T; ++t) {
        if (simd_ids && blockIdx.x == 1001 && lane == 0 && (t == 200 || t == 1224)) {  // cycles per K-step (wave-level clock)
            unsigned long long c = __builtin_amdgcn_s_memtime();
            simd_ids[16 + wave * 4 + (t == 200 ? 0 : 2)] = (unsigned)c;
            simd_ids[16 + wave * 4 + (t == 200 ? 1 : 3)] = (unsigned)(c >> 32);
        }
        const unsigned sbu = (unsigned)(unsigned long long)(smem + (t & 1) * STAGE);
        const int tn = t + 1 < T ? t + 1 : T - 1;
        // ---------------- LOAD(0): A blocks 0,1 and B block 0 ----------------
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) lds_read16(B0[kk], sbu + b_base + foff[kk]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            lds_read16(A[0][kk], sbu + a_base + foff[kk]);
            lds_read16(A[1][kk], sbu + a_base + 32 * ROWB + foff[kk]);
        }
        if (G0 == 0 && !(VARIANT & 8)) glds_part(tn, (t & 1) ^ 1, 0, 4);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (VARIANT & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][kk], B0[kk], acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][kk], B0[kk], acc[1][0], 0, 0, 0);
            if (VARIANT & 8) {  // staging loads inside the MFMA section, one per MFMA pair (as the product loop does)
                __builtin_amdgcn_sched_barrier(0);
                glds_part(tn, (t & 1) ^ 1, kk, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (VARIANT & 1) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- LOAD(1): B block 1 ----------------
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) lds_read16(B1[kk], sbu + b_base + 32 * ROWB + foff[kk]);
        if (!(VARIANT & 8)) glds_part(tn, (t & 1) ^ 1, G0 == 0 ? 4 : 0, G0 == 0 ? 8 : 4);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (VARIANT & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][kk], B1[kk], acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][kk], B1[kk], acc[1][1], 0, 0, 0);
            if (VARIANT & 8) {
                __builtin_amdgcn_sched_barrier(0);
                glds_part(tn, (t & 1) ^ 1, 4 + kk, 5 + kk);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (VARIANT & 1) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- LOAD(2): A blocks 2,3 ----------------
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            lds_read16(A[0][kk], sbu + a_base + 64 * ROWB + foff[kk]);
            lds_read16(A[1][kk], sbu + a_base + 96 * ROWB + foff[kk]);
        }
        if (G0 == 1 && !(VARIANT & 8)) glds_part(tn, (t & 1) ^ 1, 4, 8);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (VARIANT & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][kk], B1[kk], acc[2][1], 0, 0, 0);
            acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][kk], B1[kk], acc[3][1], 0, 0, 0);
        }
        if (VARIANT & 1) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- LOAD(3): nothing to read (B0 is still in registers); the second group's staging loads must
        // have landed before the barrier that lets the first group start the next K-step ----------------
        if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (VARIANT & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][kk], B0[kk], acc[2][0], 0, 0, 0);
            acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][kk], B0[kk], acc[3][0], 0, 0, 0);
        }
        if (VARIANT & 1) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... and the first group's, same barrier
        __builtin_amdgcn_s_barrier();
        if (++ksin < nk) continue;
        ksin = 0;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                for (int r = 0; r < 16; ++r) best[ni] = fmaxf(best[ni], acc[mi][ni][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
            }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();  // even out the barrier count of the two groups
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) out[((long long)blockIdx.x * 8 + wave) * 64 * 2 + ni * 64 + lane] = best[ni];
}

// ---- one wave per SIMD: 4 waves of 128 x 128, the 16 accumulator blocks pinned in AGPRs a[0:255] by inline asm (hipcc
// cannot keep 256 accumulator registers in the AGPR half on its own: it copies them around every K-step).  A third fewer
// LDS fragment reads per MFMA than the 128 x 64 wave tile (32 reads per 64 MFMAs instead of 24 per 32).
#define AG_MFMA_0(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[0:15], %0, %1, a[0:15]" ::"v"(A), "v"(B) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15")
#define AG_MFMA_1(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[16:31], %0, %1, a[16:31]" ::"v"(A), "v"(B) : "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31")
#define AG_MFMA_2(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[32:47], %0, %1, a[32:47]" ::"v"(A), "v"(B) : "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47")
#define AG_MFMA_3(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[48:63], %0, %1, a[48:63]" ::"v"(A), "v"(B) : "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63")
#define AG_MFMA_4(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[64:79], %0, %1, a[64:79]" ::"v"(A), "v"(B) : "a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79")
#define AG_MFMA_5(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[80:95], %0, %1, a[80:95]" ::"v"(A), "v"(B) : "a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95")
#define AG_MFMA_6(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[96:111], %0, %1, a[96:111]" ::"v"(A), "v"(B) : "a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111")
#define AG_MFMA_7(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[112:127], %0, %1, a[112:127]" ::"v"(A), "v"(B) : "a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127")
#define AG_MFMA_8(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[128:143], %0, %1, a[128:143]" ::"v"(A), "v"(B) : "a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143")
#define AG_MFMA_9(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[144:159], %0, %1, a[144:159]" ::"v"(A), "v"(B) : "a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159")
#define AG_MFMA_10(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[160:175], %0, %1, a[160:175]" ::"v"(A), "v"(B) : "a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175")
#define AG_MFMA_11(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[176:191], %0, %1, a[176:191]" ::"v"(A), "v"(B) : "a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191")
#define AG_MFMA_12(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[192:207], %0, %1, a[192:207]" ::"v"(A), "v"(B) : "a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207")
#define AG_MFMA_13(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[208:223], %0, %1, a[208:223]" ::"v"(A), "v"(B) : "a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223")
#define AG_MFMA_14(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[224:239], %0, %1, a[224:239]" ::"v"(A), "v"(B) : "a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239")
#define AG_MFMA_15(A, B) asm volatile("v_mfma_f32_32x32x16_f16 a[240:255], %0, %1, a[240:255]" ::"v"(A), "v"(B) : "a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255")
#define AG_MFMA(blk, A, B) AG_MFMA_##blk(A, B)
#define AG_ZERO_ALL() asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\tv_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\tv_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0\n\tv_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0\n\tv_accvgpr_write_b32 a128, 0\n\tv_accvgpr_write_b32 a129, 0\n\tv_accvgpr_write_b32 a130, 0\n\tv_accvgpr_write_b32 a131, 0\n\tv_accvgpr_write_b32 a132, 0\n\tv_accvgpr_write_b32 a133, 0\n\tv_accvgpr_write_b32 a134, 0\n\tv_accvgpr_write_b32 a135, 0\n\tv_accvgpr_write_b32 a136, 0\n\tv_accvgpr_write_b32 a137, 0\n\tv_accvgpr_write_b32 a138, 0\n\tv_accvgpr_write_b32 a139, 0\n\tv_accvgpr_write_b32 a140, 0\n\tv_accvgpr_write_b32 a141, 0\n\tv_accvgpr_write_b32 a142, 0\n\tv_accvgpr_write_b32 a143, 0\n\tv_accvgpr_write_b32 a144, 0\n\tv_accvgpr_write_b32 a145, 0\n\tv_accvgpr_write_b32 a146, 0\n\tv_accvgpr_write_b32 a147, 0\n\tv_accvgpr_write_b32 a148, 0\n\tv_accvgpr_write_b32 a149, 0\n\tv_accvgpr_write_b32 a150, 0\n\tv_accvgpr_write_b32 a151, 0\n\tv_accvgpr_write_b32 a152, 0\n\tv_accvgpr_write_b32 a153, 0\n\tv_accvgpr_write_b32 a154, 0\n\tv_accvgpr_write_b32 a155, 0\n\tv_accvgpr_write_b32 a156, 0\n\tv_accvgpr_write_b32 a157, 0\n\tv_accvgpr_write_b32 a158, 0\n\tv_accvgpr_write_b32 a159, 0\n\tv_accvgpr_write_b32 a160, 0\n\tv_accvgpr_write_b32 a161, 0\n\tv_accvgpr_write_b32 a162, 0\n\tv_accvgpr_write_b32 a163, 0\n\tv_accvgpr_write_b32 a164, 0\n\tv_accvgpr_write_b32 a165, 0\n\tv_accvgpr_write_b32 a166, 0\n\tv_accvgpr_write_b32 a167, 0\n\tv_accvgpr_write_b32 a168, 0\n\tv_accvgpr_write_b32 a169, 0\n\tv_accvgpr_write_b32 a170, 0\n\tv_accvgpr_write_b32 a171, 0\n\tv_accvgpr_write_b32 a172, 0\n\tv_accvgpr_write_b32 a173, 0\n\tv_accvgpr_write_b32 a174, 0\n\tv_accvgpr_write_b32 a175, 0\n\tv_accvgpr_write_b32 a176, 0\n\tv_accvgpr_write_b32 a177, 0\n\tv_accvgpr_write_b32 a178, 0\n\tv_accvgpr_write_b32 a179, 0\n\tv_accvgpr_write_b32 a180, 0\n\tv_accvgpr_write_b32 a181, 0\n\tv_accvgpr_write_b32 a182, 0\n\tv_accvgpr_write_b32 a183, 0\n\tv_accvgpr_write_b32 a184, 0\n\tv_accvgpr_write_b32 a185, 0\n\tv_accvgpr_write_b32 a186, 0\n\tv_accvgpr_write_b32 a187, 0\n\tv_accvgpr_write_b32 a188, 0\n\tv_accvgpr_write_b32 a189, 0\n\tv_accvgpr_write_b32 a190, 0\n\tv_accvgpr_write_b32 a191, 0\n\tv_accvgpr_write_b32 a192, 0\n\tv_accvgpr_write_b32 a193, 0\n\tv_accvgpr_write_b32 a194, 0\n\tv_accvgpr_write_b32 a195, 0\n\tv_accvgpr_write_b32 a196, 0\n\tv_accvgpr_write_b32 a197, 0\n\tv_accvgpr_write_b32 a198, 0\n\tv_accvgpr_write_b32 a199, 0\n\tv_accvgpr_write_b32 a200, 0\n\tv_accvgpr_write_b32 a201, 0\n\tv_accvgpr_write_b32 a202, 0\n\tv_accvgpr_write_b32 a203, 0\n\tv_accvgpr_write_b32 a204, 0\n\tv_accvgpr_write_b32 a205, 0\n\tv_accvgpr_write_b32 a206, 0\n\tv_accvgpr_write_b32 a207, 0\n\tv_accvgpr_write_b32 a208, 0\n\tv_accvgpr_write_b32 a209, 0\n\tv_accvgpr_write_b32 a210, 0\n\tv_accvgpr_write_b32 a211, 0\n\tv_accvgpr_write_b32 a212, 0\n\tv_accvgpr_write_b32 a213, 0\n\tv_accvgpr_write_b32 a214, 0\n\tv_accvgpr_write_b32 a215, 0\n\tv_accvgpr_write_b32 a216, 0\n\tv_accvgpr_write_b32 a217, 0\n\tv_accvgpr_write_b32 a218, 0\n\tv_accvgpr_write_b32 a219, 0\n\tv_accvgpr_write_b32 a220, 0\n\tv_accvgpr_write_b32 a221, 0\n\tv_accvgpr_write_b32 a222, 0\n\tv_accvgpr_write_b32 a223, 0\n\tv_accvgpr_write_b32 a224, 0\n\tv_accvgpr_write_b32 a225, 0\n\tv_accvgpr_write_b32 a226, 0\n\tv_accvgpr_write_b32 a227, 0\n\tv_accvgpr_write_b32 a228, 0\n\tv_accvgpr_write_b32 a229, 0\n\tv_accvgpr_write_b32 a230, 0\n\tv_accvgpr_write_b32 a231, 0\n\tv_accvgpr_write_b32 a232, 0\n\tv_accvgpr_write_b32 a233, 0\n\tv_accvgpr_write_b32 a234, 0\n\tv_accvgpr_write_b32 a235, 0\n\tv_accvgpr_write_b32 a236, 0\n\tv_accvgpr_write_b32 a237, 0\n\tv_accvgpr_write_b32 a238, 0\n\tv_accvgpr_write_b32 a239, 0\n\tv_accvgpr_write_b32 a240, 0\n\tv_accvgpr_write_b32 a241, 0\n\tv_accvgpr_write_b32 a242, 0\n\tv_accvgpr_write_b32 a243, 0\n\tv_accvgpr_write_b32 a244, 0\n\tv_accvgpr_write_b32 a245, 0\n\tv_accvgpr_write_b32 a246, 0\n\tv_accvgpr_write_b32 a247, 0\n\tv_accvgpr_write_b32 a248, 0\n\tv_accvgpr_write_b32 a249, 0\n\tv_accvgpr_write_b32 a250, 0\n\tv_accvgpr_write_b32 a251, 0\n\tv_accvgpr_write_b32 a252, 0\n\tv_accvgpr_write_b32 a253, 0\n\tv_accvgpr_write_b32 a254, 0\n\tv_accvgpr_write_b32 a255, 0\n\ts_nop 3" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191","a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223","a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239","a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255")
template <int REG> __device__ inline float ag_read() { float v; asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(REG)); return v; }

template <int DEPTHK>
__global__ __launch_bounds__(256, 1) void gemm_agpr(const _Float16* __restrict__ xb, const _Float16* __restrict__ xq,
                                                    float* __restrict__ out, int ntiles, int nk, long long ld, int qmod,
                                                    unsigned* __restrict__ stamps32) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7, li = (blockIdx.x >> 3) & 31, gen = blockIdx.x >> 8;
    const int gq = qmod;
    const int qt = (gen * 8 + xcd) * gq + (li % gq);
    const int slab = li / gq, nsl = 32 / gq;
    const int tile0 = slab * (ntiles / nsl);
    const long long q0 = (long long)qt * BQ;
    constexpr int RPW = 128, GL = 16;  // staged rows / glds per wave per K-step
    unsigned loff[GL];
    const char* sbase[GL];
#pragma unroll
    for (int i = 0; i < GL; ++i) {
        int row = wave * RPW + i * 8 + (lane >> 3);
        int col = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        bool isq = row >= BC;
        long long grow = isq ? q0 + (row - BC) : row;
        loff[i] = (unsigned)((grow * ld + col) * 2);
        sbase[i] = (const char*)(isq ? xq : xb);
    }
    int foff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        foff[kk] = (lane & 31) * ROWB + ((((kk * 2) + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    const int a_base = wm * 128 * ROWB;
    const int b_base = BC * ROWB + wn * 128 * ROWB;
    float best[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
    AG_ZERO_ALL();
    const int T = ntiles * nk;
    auto glds_one = [&](int t, int buf, int i) {
        int ti = t / nk, ks = t - ti * nk;
        ti = (ti + tile0) % ntiles;
        int row = wave * RPW + i * 8;
        bool isq = row >= BC;
        long long tile_off = isq ? 0 : (long long)ti * BC * ld * 2;
        glds16(sbase[i] + tile_off + loff[i] + ks * BK * 2, smem + buf * STAGE + row * ROWB);
    };
#pragma unroll
    for (int i = 0; i < GL; ++i) glds_one(0, 0, i);
    int ksin = 0;
    half8 Af[2][4], Bf[2][4];
    for (int t = 0; t < T; ++t) {
        if (stamps32 && blockIdx.x == 1001 && lane == 0 && (t == 200 || t == 1224)) {
            unsigned long long c = __builtin_amdgcn_s_memtime();
            stamps32[16 + wave * 4 + (t == 200 ? 0 : 2)] = (unsigned)c;
            stamps32[16 + wave * 4 + (t == 200 ? 1 : 3)] = (unsigned)(c >> 32);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned sbu = (unsigned)(unsigned long long)(smem + (t & 1) * STAGE);
        const int tn = t + 1 < T ? t + 1 : T - 1;
        // fragments of k-slices 0 and 1; A0 and the B's first: the first MFMA needs only A0 and B0
        lds_read16(Af[0][0], sbu + a_base + foff[0]);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) lds_read16(Bf[0][ni], sbu + b_base + ni * 32 * ROWB + foff[0]);
#pragma unroll
        for (int mi = 1; mi < 4; ++mi) lds_read16(Af[0][mi], sbu + a_base + mi * 32 * ROWB + foff[0]);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) lds_read16(Af[1][mi], sbu + a_base + mi * 32 * ROWB + foff[1]);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) lds_read16(Bf[1][ni], sbu + b_base + ni * 32 * ROWB + foff[1]);
        static_for<4>([&](auto kc) {
            constexpr int kk = decltype(kc)::value;
            constexpr int s = kk & 1;
            // the set of this k-slice must have landed; the other set (8 reads) may still be in flight
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (kk == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            else if constexpr (kk == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            static_for<4>([&](auto mc) {
                constexpr int mi = decltype(mc)::value;
                if constexpr (mi == 0) { AG_MFMA(0, Af[s][0], Bf[s][0]); AG_MFMA(1, Af[s][0], Bf[s][1]); AG_MFMA(2, Af[s][0], Bf[s][2]); AG_MFMA(3, Af[s][0], Bf[s][3]); }
                if constexpr (mi == 1) { AG_MFMA(4, Af[s][1], Bf[s][0]); AG_MFMA(5, Af[s][1], Bf[s][1]); AG_MFMA(6, Af[s][1], Bf[s][2]); AG_MFMA(7, Af[s][1], Bf[s][3]); }
                if constexpr (mi == 2) { AG_MFMA(8, Af[s][2], Bf[s][0]); AG_MFMA(9, Af[s][2], Bf[s][1]); AG_MFMA(10, Af[s][2], Bf[s][2]); AG_MFMA(11, Af[s][2], Bf[s][3]); }
                if constexpr (mi == 3) { AG_MFMA(12, Af[s][3], Bf[s][0]); AG_MFMA(13, Af[s][3], Bf[s][1]); AG_MFMA(14, Af[s][3], Bf[s][2]); AG_MFMA(15, Af[s][3], Bf[s][3]); }
                // staging loads of the next K-step: all 16 within the first two k-slices (two per 4 MFMAs)
                if constexpr (kk < 2) {
                    glds_one(tn, (t & 1) ^ 1, kk * 8 + mi * 2);
                    glds_one(tn, (t & 1) ^ 1, kk * 8 + mi * 2 + 1);
                }
            });
            // this set's registers are free again (its MFMAs have been issued): fetch k-slice kk + 2 into it
            if constexpr (kk + 2 < 4) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) lds_read16(Af[s][mi], sbu + a_base + mi * 32 * ROWB + foff[kk + 2]);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) lds_read16(Bf[s][ni], sbu + b_base + ni * 32 * ROWB + foff[kk + 2]);
            }
        });
        if (++ksin < nk) continue;
        ksin = 0;
        // tile epilogue: per-lane maxima per query block, accumulators back to zero
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        static_for<16>([&](auto bc) {
            constexpr int blk = decltype(bc)::value;
            constexpr int ni = blk & 3;
            static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                best[ni] = fmaxf(best[ni], ag_read<blk * 16 + r>());
            });
        });
        AG_ZERO_ALL();
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) out[((long long)blockIdx.x * 4 + wave) * 256 + ni * 64 + lane] = best[ni];
}

void run_agpr(const char* name, const _Float16* xb, const _Float16* xq, float* out, int nqt, int ntiles, int nk, long long ld,
              const float* ref_out, unsigned* stamps32) {
    auto k = gemm_agpr<2>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(nqt), dim3(256), 2 * STAGE, 0, xb, xq, out, ntiles, nk, ld, 32, stamps32);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    double fl = 2.0 * nqt * 256.0 * ntiles * 256.0 * nk * 64.0;
    // per-query maxima must equal the reference kernel's: reduce both layouts to [block][256 queries] on the host
    size_t n = (size_t)nqt * 1024, bad = 0;
    std::vector<float> a(n), b(n);
    CHECK(hipMemcpy(a.data(), out, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), ref_out, n * 4, hipMemcpyDeviceToHost));
    for (int blk = 0; blk < nqt; ++blk) {
        float qa[256], qb[256];
        for (int q = 0; q < 256; ++q) qa[q] = qb[q] = -3e38f;
        for (int w = 0; w < 4; ++w)       // new layout: wave = wm * 2 + wn, [ni 0..3][lane]: query wn*128 + ni*32 + (lane & 31)
            for (int ni = 0; ni < 4; ++ni)
                for (int l = 0; l < 64; ++l) {
                    int q = (w & 1) * 128 + ni * 32 + (l & 31);
                    float v = a[((size_t)blk * 4 + w) * 256 + ni * 64 + l];
                    qa[q] = v > qa[q] ? v : qa[q];
                }
        for (int w = 0; w < 8; ++w)       // reference layout: wave = wm * 4 + wn, [ni 0..1][lane]: query wn*64 + ni*32 + (lane & 31)
            for (int ni = 0; ni < 2; ++ni)
                for (int l = 0; l < 64; ++l) {
                    int q = (w & 3) * 64 + ni * 32 + (l & 31);
                    float v = b[((size_t)blk * 8 + w) * 128 + ni * 64 + l];
                    qb[q] = v > qb[q] ? v : qb[q];
                }
        for (int q = 0; q < 256; ++q) bad += qa[q] != qb[q];
    }
    unsigned hs[64];
    CHECK(hipMemcpy(hs, stamps32, sizeof(hs), hipMemcpyDeviceToHost));
    auto u64of = [&](int i) { return ((unsigned long long)hs[i + 1] << 32) | hs[i]; };
    double cyc0 = (double)(u64of(16 + 2) - u64of(16)) / 1024.0;
    double us_per_kstep = best * 1e3 / ((double)ntiles * nk * (nqt / 256.0));
    printf("%-44s %8.2f ms  %7.1f TFLOP/s   cyc/K-step %.0f  clock %.2f GHz   query maxima differing from the reference: %zu of %zu\n",
           name, best, fl / (best * 1e-3) / 1e12, cyc0, cyc0 / us_per_kstep / 1e3, bad, (size_t)nqt * 256);
    fflush(stdout);
}

template <int VARIANT>
void run_pp(const char* name, const _Float16* xb, const _Float16* xq, float* out, int nqt, int ntiles, int nk, long long ld,
            const float* ref_out, unsigned* simd_ids) {
    auto k = gemm_pingpong<VARIANT>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(nqt), dim3(512), 2 * STAGE, 0, xb, xq, out, ntiles, nk, ld, 32, simd_ids);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    double fl = 2.0 * nqt * 256.0 * ntiles * 256.0 * nk * 64.0;
    // same maxima as the reference kernel?  (the reduction order inside a dot product is identical: K ascending)
    size_t n = (size_t)nqt * 8 * 128, bad = 0;
    std::vector<float> a(n), b(n);
    CHECK(hipMemcpy(a.data(), out, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), ref_out, n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) bad += a[i] != b[i];
    unsigned hs[64];
    CHECK(hipMemcpy(hs, simd_ids, sizeof(hs), hipMemcpyDeviceToHost));
    auto u64of = [&](int i) { return ((unsigned long long)hs[i + 1] << 32) | hs[i]; };
    double cyc0 = (double)(u64of(16 + 2) - u64of(16)) / 1024.0, cyc4 = (double)(u64of(16 + 16 + 2) - u64of(16 + 16)) / 1024.0;
    double us_per_kstep = best * 1e3 / ((double)ntiles * nk * (nqt / 256.0));
    printf("%-44s %8.2f ms  %7.1f TFLOP/s   cyc/K-step wave0 %.0f wave4 %.0f  clock %.2f GHz   mismatches: %zu\n", name, best,
           fl / (best * 1e-3) / 1e12, cyc0, cyc4, cyc0 / us_per_kstep / 1e3, bad);
    fflush(stdout);
}


// ---- query operand straight from global memory: the B fragments (queries, shared by only WM = 2 waves) are loaded one
// K-step ahead with global_load_dwordx4 into registers (the natural row-major layout already gives lane l the 16 bytes
// row l % 32, k-chunk l / 32 of a 32x32x16 fragment), so only the corpus goes through LDS: 16 instead of 24 fragment
// reads per wave and K-step, 32 KB instead of 64 KB staged per K-step, and room for a ring of STAGES 32 KB stages
// (STAGES = 3: the top-of-step wait is vmcnt(4), the staging loads get more than a full K-step to land).
// Register sets of the B fragments alternate between two K-steps (loop unrolled by two), 64 VGPRs in all.
__device__ inline void gload16(half8& dst, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ inline void vm_wait8(half8 (&b)[4][2]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]),
                 "+v"(b[3][0]), "+v"(b[3][1]) : "n"(N) : "memory");
}
template <int N>
__device__ inline void lds_wait1(half8& a) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}
template <int DEPTH, int STAGES, int BFIRST, int STAMP, int SWZ = 0>
__global__ __launch_bounds__(512, 2) void gemm_bdirect(const _Float16* __restrict__ xb, const _Float16* __restrict__ xq,
                                                        float* __restrict__ out, int ntiles, int nk, long long ld,
                                                        unsigned long long* __restrict__ stamps, int qmod) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MI = 4, NI = 2, WN = 4, NW = 8, NF = 16;
    constexpr int CSTAGE = BC * ROWB;                      // 32 KB: corpus rows only
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int xcd = blockIdx.x & 7, li = (blockIdx.x >> 3) & 31, gen = blockIdx.x >> 8;
    const int gq = qmod;
    const int qt = (gen * 8 + xcd) * gq + (li % gq);
    const int slab = li / gq, nsl = 32 / gq;
    const int tile0 = slab * (ntiles / nsl);
    const long long q0 = (long long)qt * BQ;
    unsigned loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = wave * 32 + i * 8 + (lane >> 3);
        int col = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        loff[i] = (unsigned)((row * ld + col) * 2);
    }
    int foff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        foff[kk] = (lane & 31) * ROWB + ((((kk * 2) + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    const int a_base = wm * MI * 32 * ROWB;
    const char* bq[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
        bq[ni] = SWZ ? (const char*)xq + (((long long)qt * 4 + wn) * nk * 8 + ni) * 1024 + lane * 16   // [64-query block][ks][kk][ni][lane]
                     : (const char*)(xq + (q0 + wn * 64 + ni * 32 + (lane & 31)) * ld + (lane >> 5) * 8);
    f32x16 acc[MI][NI];
    float best[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) best[ni] = -1e30f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int T = ntiles * nk;
    auto stage = [&](int t, int i) {                       // one staging load (8 corpus rows of this wave) of K-step t
        int tc = t < T ? t : T - 1;
        int ti = tc / nk, ks = tc - ti * nk; ti = (ti + tile0) % ntiles;
        glds16((const char*)xb + (long long)ti * BC * ld * 2 + loff[i] + ks * BK * 2,
               smem + (t % STAGES) * CSTAGE + (wave * 32 + i * 8) * ROWB);
    };
    auto bload = [&](int t, int j, half8 (&B)[4][2]) {    // fragment j = kk * 2 + ni of K-step t
        int tc = t < T ? t : T - 1;
        int ks = tc % nk;
        if (SWZ) gload16(B[j >> 1][j & 1], bq[j & 1] + ks * 8192 + (j >> 1) * 2048);
        else gload16(B[j >> 1][j & 1], bq[j & 1] + ks * BK * 2 + (j >> 1) * 32);
    };
    half8 B0[4][2], B1[4][2];
    // prologue: stages 0 .. STAGES-2 and the fragments of K-step 0; the LAST group issued is the one the steady-state
    // wait may leave outstanding (4 staging loads if the B loads come first in a K-step, else the 8 B loads)
    if (BFIRST) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bload(0, j, B0);
        for (int s = 0; s + 1 < STAGES; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) stage(s, i);
    } else {
        for (int s = 0; s + 1 < STAGES; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) stage(s, i);
#pragma unroll
        for (int j = 0; j < 8; ++j) bload(0, j, B0);
    }
    int ksin = 0;
    auto kstep = [&](int t, half8 (&Bc)[4][2], half8 (&Bn)[4][2]) {
        unsigned long long ta = 0, tb = 0;
        const bool st = STAMP && stamps && t >= 200 && t < 264 && (blockIdx.x == 0 || blockIdx.x == 1001);
        if (st) ta = __builtin_amdgcn_s_memtime();
        // stage t and Bc complete: with B loads first only the youngest staging group (stage t + STAGES - 2) may be in flight
        if (BFIRST && STAGES > 2) vm_wait8<4>(Bc); else vm_wait8<0>(Bc);
        __builtin_amdgcn_s_barrier();                      // raw: __syncthreads() would add its own vmcnt(0)
        if (st) tb = __builtin_amdgcn_s_memtime();
        const unsigned sbu = (unsigned)(unsigned long long)(smem + (t % STAGES) * CSTAGE);
        half8 Af[DEPTH + 1];
        if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int f = 0; f < DEPTH; ++f) lds_read16(Af[f], sbu + a_base + (f % MI) * 32 * ROWB + foff[f / MI]);
        static_for<NF>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            constexpr int kk = f / MI, mi = f % MI;
            if constexpr (f + DEPTH < NF) {
                constexpr int f2 = f + DEPTH;
                lds_read16(Af[f2 % (DEPTH + 1)], sbu + a_base + (f2 % MI) * 32 * ROWB + foff[f2 / MI]);
            }
            if constexpr (BFIRST) {
                if constexpr (f < 8) bload(t + 1, f, Bn);
                else if constexpr (f < 12) stage(t + STAGES - 1, f - 8);
            } else {
                if constexpr (f < 4) stage(t + STAGES - 1, f);
                else if constexpr (f < 12) bload(t + 1, f - 4, Bn);
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int last = f + DEPTH < NF ? f + DEPTH : NF - 1;   // youngest A read issued so far
            lds_wait1<last - f>(Af[f % (DEPTH + 1)]);
            acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[f % (DEPTH + 1)], Bc[kk][0], acc[mi][0], 0, 0, 0);
            acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[f % (DEPTH + 1)], Bc[kk][1], acc[mi][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        if (st) {
            unsigned long long tc = __builtin_amdgcn_s_memtime();
            if (lane == 0) {
                unsigned long long* o = stamps + (((blockIdx.x ? 1 : 0) * NW + wave) * 64 + (t - 200)) * 3;
                o[0] = ta; o[1] = tb; o[2] = tc;
            }
        }
        if (++ksin < nk) return;
        ksin = 0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int r = 0; r < 16; ++r) best[ni] = fmaxf(best[ni], acc[mi][ni][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
            }
    };
    for (int t = 0; t < T; t += 2) {                       // T = ntiles * nk is even
        kstep(t, B0, B1);
        kstep(t + 1, B1, B0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) out[((long long)blockIdx.x * NW + wave) * 64 * NI + ni * 64 + lane] = best[ni];
}

template <int DEPTH, int STAGES, int BFIRST, int SWZ = 0>
void run_bdirect(const char* name, const _Float16* xb, const _Float16* xq, float* out, int nqt, int ntiles, int nk, long long ld,
                 const float* ref_out, unsigned long long* stamps) {
    auto k = gemm_bdirect<DEPTH, STAGES, BFIRST, 1, SWZ>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * BC * ROWB));
    const size_t nst = 2 * 8 * 64 * 3;
    CHECK(hipMemset(stamps, 0, nst * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(nqt), dim3(512), STAGES * BC * ROWB, 0, xb, xq, out, ntiles, nk, ld, stamps, 32);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    double fl = 2.0 * nqt * 256.0 * ntiles * 256.0 * nk * 64.0;
    size_t n = (size_t)nqt * 8 * 128, bad = 0;
    std::vector<float> a(n), b(n);
    CHECK(hipMemcpy(a.data(), out, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), ref_out, n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) bad += a[i] != b[i];
    std::vector<unsigned long long> hs(nst);
    CHECK(hipMemcpy(hs.data(), stamps, nst * 8, hipMemcpyDeviceToHost));
    double per = (double)(hs[63 * 3] - hs[0]) / 63.0;
    double wt = 0, mf = 0;
    for (int i = 0; i < 64; ++i) { wt += (double)(hs[i * 3 + 1] - hs[i * 3]); mf += (double)(hs[i * 3 + 2] - hs[i * 3 + 1]); }
    double us_per_kstep = best * 1e3 / ((double)ntiles * nk * (nqt / 256.0));
    printf("%-44s %8.2f ms  %7.1f TFLOP/s   cyc/K-step %.0f (wait %.0f, mfma %.0f)  clock %.2f GHz   mismatches: %zu\n", name, best,
           fl / (best * 1e-3) / 1e12, per, wt / 64, mf / 64, per / us_per_kstep / 1e3, bad);
    fflush(stdout);
}

template <int WM, int WN, int MI, int NI, int WPE, int DEPTH, int BPOS, int PRIO, int SPREAD, int ABL = 0>
void run(const char* name, const _Float16* xb, const _Float16* xq, float* out, int nqt, int ntiles, int nk, long long ld, unsigned long long* stamps = nullptr, int qmod = 32) {
    auto k = gemm_probe<WM, WN, MI, NI, WPE, DEPTH, BPOS, PRIO, SPREAD, 0, ABL>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(nqt), dim3(WM * WN * 64), 2 * STAGE, 0, xb, xq, out, ntiles, nk, ld, stamps, qmod);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    double fl = 2.0 * nqt * 256.0 * ntiles * 256.0 * nk * 64.0;
    printf("%-28s %8.2f ms  %7.1f TFLOP/s\n", name, best, fl / (best * 1e-3) / 1e12);
    fflush(stdout);
}

template <int SPREAD, int ABL>
void clock_of(const char* name, const _Float16* xb, const _Float16* xq, float* out, int nqt, int ntiles, int nk, long long d,
              unsigned long long* stamps) {
    const size_t nst = 2 * 8 * 64 * 3;
    CHECK(hipMemset(stamps, 0, nst * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto k = gemm_probe<2, 4, 4, 2, 2, 2, 2, 1, SPREAD, 1, ABL>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(nqt), dim3(512), 2 * STAGE, 0, xb, xq, out, ntiles, nk, d, stamps, 32);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> hs(nst);
    CHECK(hipMemcpy(hs.data(), stamps, nst * 8, hipMemcpyDeviceToHost));
    double per = (double)(hs[63 * 3] - hs[0]) / 63.0;
    double wt = 0, mf = 0;
    for (int i = 0; i < 64; ++i) { wt += (double)(hs[i * 3 + 1] - hs[i * 3]); mf += (double)(hs[i * 3 + 2] - hs[i * 3 + 1]); }
    const unsigned long long* o4 = hs.data() + (size_t)4 * 64 * 3;
    double wt4 = 0, mf4 = 0;
    for (int i = 0; i < 64; ++i) { wt4 += (double)(o4[i * 3 + 1] - o4[i * 3]); mf4 += (double)(o4[i * 3 + 2] - o4[i * 3 + 1]); }
    double fl = 2.0 * nqt * 256.0 * ntiles * 256.0 * nk * 64.0;
    double us_per_kstep = best * 1e3 / ((double)ntiles * nk * (nqt / 256.0));
    printf("%-32s %7.2f ms %7.1f TF  cyc/K-step %.0f  clock %.2f GHz  util %.1f %%  wave0 wait %.0f mfma %.0f | wave4 wait %.0f mfma %.0f\n",
           name, best, fl / (best * 1e-3) / 1e12, per, per / us_per_kstep / 1e3, 2048.0 / per * 100, wt / 64, mf / 64, wt4 / 64, mf4 / 64);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int d = 768, nk = d / 64;
    const int nqt = 2048;                 // 2048 query tiles of 256 = 512k queries (8 blocks per CU)
    const int ntiles = 256;               // 65 536 corpus rows per block
    const long long nq = (long long)nqt * 256, nb = (long long)ntiles * 256;
    std::vector<_Float16> h((size_t)(nq > nb ? nq : nb) * d);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f * 0.06f);
    _Float16 *xb, *xq;
    float *out, *ref;
    unsigned* ids;
    CHECK(hipMalloc(&xb, nb * d * 2));
    CHECK(hipMalloc(&xq, nq * d * 2));
    CHECK(hipMalloc(&out, (size_t)nqt * 16 * 64 * 4 * 4));
    CHECK(hipMalloc(&ref, (size_t)nqt * 16 * 64 * 4 * 4));
    CHECK(hipMalloc(&ids, 1024));
    CHECK(hipMemcpy(xb, h.data(), nb * d * 2, hipMemcpyHostToDevice));
    CHECK(hipMemset(xq, 0, nq * d * 2));
    CHECK(hipMemcpy(xq, h.data() + 12345 * d, nq * d * 2 - 12345 * d * 2, hipMemcpyHostToDevice));
    unsigned long long* stamps;
    CHECK(hipMalloc(&stamps, 2 * 8 * 64 * 3 * 8));
    // fragment-major copy of the queries: [64-query block][K-step][kk][ni][lane] x 8 halfs = what lane `lane` feeds the MFMA
    _Float16* xqs;
    CHECK(hipMalloc(&xqs, nq * d * 2));
    {
        const _Float16* hq = h.data() + (size_t)12345 * d;
        std::vector<_Float16> sw((size_t)nq * d);
        const size_t nrows_ok = (h.size() / d) - 12345;     // rows of hq that exist on the host (the rest was never copied)
        for (long long blk = 0; blk < nq / 64; ++blk)
            for (int ks = 0; ks < nk; ++ks)
                for (int kk = 0; kk < 4; ++kk)
                    for (int ni = 0; ni < 2; ++ni)
                        for (int l = 0; l < 64; ++l) {
                            size_t row = (size_t)blk * 64 + ni * 32 + (l & 31);
                            _Float16* dst = &sw[((((size_t)blk * nk + ks) * 4 + kk) * 2 + ni) * 512 + (size_t)l * 8];
                            for (int e = 0; e < 8; ++e)
                                dst[e] = row < nrows_ok ? hq[row * d + ks * 64 + kk * 16 + (l >> 5) * 8 + e] : (_Float16)0;
                        }
        CHECK(hipMemcpy(xqs, sw.data(), (size_t)nq * d * 2, hipMemcpyHostToDevice));
    }
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) {  // the same binaries on zero-filled operands: no data-dependent switching power
            CHECK(hipMemset(xb, 0, nb * d * 2));
            CHECK(hipMemset(xq, 0, nq * d * 2));
            CHECK(hipMemset(xqs, 0, nq * d * 2));
            printf("---- zero-filled operands ----\n");
        }
        clock_of<3, 0>("product loop (stamped)", xb, xq, ref, nqt, ntiles, nk, d, stamps);
        run<2, 4, 4, 2, 2, 2, 2, 1, 3>("8 waves 128x64, asm waits (product loop)", xb, xq, ref, nqt, ntiles, nk, d);
        if (argc > 1) {   // the earlier probes (tuning log): AGPR accumulators, ping-pong
            run_agpr("4 waves 128x128, accumulators in AGPRs (asm)", xb, xq, out, nqt, ntiles, nk, d, ref, ids);
            run_pp<8>("ping-pong, staging inside MFMA 0/1", xb, xq, out, nqt, ntiles, nk, d, ref, ids);
        }
        run_bdirect<2, 3, 1>("queries from global (row-major), 3 stages", xb, xq, out, nqt, ntiles, nk, d, ref, stamps);
        run_bdirect<2, 2, 0, 1>("queries fragment-major from global, 2 stages", xb, xqs, out, nqt, ntiles, nk, d, ref, stamps);
        run_bdirect<2, 3, 1, 1>("same, 3 stages, vmcnt(4)", xb, xqs, out, nqt, ntiles, nk, d, ref, stamps);
        run_bdirect<2, 4, 1, 1>("same, 4 stages, vmcnt(4)", xb, xqs, out, nqt, ntiles, nk, d, ref, stamps);
        run_bdirect<3, 3, 1, 1>("same, 3 stages, A depth 3", xb, xqs, out, nqt, ntiles, nk, d, ref, stamps);

    }
    unsigned hid[8];
    CHECK(hipMemcpy(hid, ids, 32, hipMemcpyDeviceToHost));
    printf("wave -> SIMD of block 0:");
    for (int w = 0; w < 8; ++w) printf(" %u:%u", w, (hid[w] >> 4) & 3);
    printf("\n");
    return 0;
}

// Design-space probe (development aid, not part of the library): the tile kernel's main loop WITHOUT any top-k, as a
// plain C[q][j] max-reduction GEMM, templated on the wave layout.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_gemm.hip -o /tmp/probe_gemm && /tmp/probe_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int BC = 256, BQ = 256, BK = 64, ROWB = 128, STAGE = (BC + BQ) * ROWB;

__device__ inline void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)l, 16, 0, 0);
}

// WM x WN waves, each wave MI x NI accumulator blocks of 32x32:  WM*MI*32 == 256, WN*NI*32 == 256
template <int WM, int WN, int MI, int NI, int WPE, int DEPTH, int BPOS, int PRIO, int SPREAD>
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
        const bool st = stamps && t >= 200 && t < 264 && (blockIdx.x == 0 || blockIdx.x == 1001);
        if (st) ta = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st) tb = __builtin_amdgcn_s_memtime();
        const char* sb = smem + buf * STAGE;
        const int tn = t + 1 < T ? t + 1 : T - 1;
        const int nti0 = tn / nk, nks = tn - nti0 * nk; const int nti = (nti0 + tile0) % ntiles;
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
            if ((SPREAD ? ((f & 1) == 0 && (f >> 1) < GL) : (f < GL))) {
                const int gi = SPREAD ? (f >> 1) : f;
                int row = wave * RPW + gi * 8;
                bool isq = row >= BC;
                long long tile_off = isq ? 0 : (long long)nti * BC * ld * 2;
                glds16(sbase[gi] + tile_off + loff[gi] + nks * BK * 2, smem + (buf ^ 1) * STAGE + row * ROWB);
            }
            if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[f % (DEPTH + 1)], Bf[kk & 1][ni], acc[mi][ni], 0, 0, 0);
            if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
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

template <int WM, int WN, int MI, int NI, int WPE, int DEPTH, int BPOS, int PRIO, int SPREAD>
void run(const char* name, const _Float16* xb, const _Float16* xq, float* out, int nqt, int ntiles, int nk, long long ld, unsigned long long* stamps = nullptr, int qmod = 32) {
    auto k = gemm_probe<WM, WN, MI, NI, WPE, DEPTH, BPOS, PRIO, SPREAD>;
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

int main() {
    const int d = 768, nk = d / 64;
    const int nqt = 2048 * 2;             // 4096 query tiles of 256 = 1M queries (16 blocks per CU)
    const int ntiles = 256;               // 65 536 corpus rows per block
    const long long nq = (long long)nqt * 256, nb = (long long)ntiles * 256;
    std::vector<_Float16> h((size_t)(nq > nb ? nq : nb) * d);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f * 0.06f);
    _Float16 *xb, *xq;
    float* out;
    CHECK(hipMalloc(&xb, nb * d * 2));
    CHECK(hipMalloc(&xq, nq * d * 2));
    CHECK(hipMalloc(&out, (size_t)nqt * 16 * 64 * 4 * 4));
    CHECK(hipMemcpy(xb, h.data(), nb * d * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(xq, h.data(), nq * d * 2, hipMemcpyHostToDevice));
    unsigned long long* stamps;
    const size_t nst = 2 * 8 * 64 * 3;
    CHECK(hipMalloc(&stamps, nst * 8));
    auto clock_of = [&](const char* name, int qmod) {
        CHECK(hipMemset(stamps, 0, nst * 8));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        auto k = gemm_probe<2, 4, 4, 2, 2, 2, 2, 1, 0>;
        CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
        float best = 1e30f;
        for (int it = 0; it < 4; ++it) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(nqt), dim3(512), 2 * STAGE, 0, xb, xq, out, ntiles, nk, (long long)d, stamps, qmod);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        std::vector<unsigned long long> hs(nst);
        CHECK(hipMemcpy(hs.data(), stamps, nst * 8, hipMemcpyDeviceToHost));
        double per = (double)(hs[63 * 3] - hs[0]) / 63.0;
        double fl = 2.0 * nqt * 256.0 * ntiles * 256.0 * nk * 64.0;
        double us_per_kstep = best * 1e3 / ((double)ntiles * nk * (nqt / 256.0));
        printf("%-34s %8.2f ms %7.1f TFLOP/s  cycles/K-step %.0f  -> clock %.2f GHz, MFMA util %.1f %%\n", name, best,
               fl / (best * 1e-3) / 1e12, per, per / us_per_kstep / 1e3, 2048.0 / per * 100);
        fflush(stdout);
    };
    clock_of("32 query tiles x 1 slab / XCD", 32);
    clock_of("16 x 2", 16);
    clock_of("8 x 4", 8);
    clock_of("4 x 8", 4);
    clock_of("2 x 16", 2);
    clock_of("32 x 1 again", 32);
    return 0;
}

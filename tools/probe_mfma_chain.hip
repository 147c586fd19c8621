// Development probe (round 6): how fast does one SIMD issue v_mfma_f32_32x32x16_f16 as a function of the number of independent
// accumulator chains per wave, their interleaving, the waves per SIMD, and what sits between two MFMAs of a chain?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_mfma_chain.hip -o /tmp/probe_mfma_chain && /tmp/probe_mfma_chain
// Prints s_memtime cycles per MFMA per SIMD (ideal 32) for each pattern.  Operands are random fp16 of unit-row magnitude.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// PATTERN 0: NCH chains round-robin (c0 c1 .. c0 c1 ..); 1: bursts of BURST per chain (c0 x BURST, c1 x BURST, ..)
// FILL: instructions between consecutive MFMAs: 0 none, 1 one s_nop 0, 2 an LDS read + counted wait (like the kernels' K-steps)
template <int NCH, int PATTERN, int BURST, int FILL, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void chain(const half8* __restrict__ x, float* __restrict__ out, int iters,
                                                               unsigned long long* __restrict__ cyc) {
    __shared__ __attribute__((aligned(16))) char smem[16384];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384 / 16; i += WAVES * 64) ((half8*)smem)[i] = x[i & 1023];
    __syncthreads();
    half8 A[4], B[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        A[i] = x[(tid * 7 + i) & 1023];
        B[i] = x[(tid * 13 + 5 + i) & 1023];
    }
    f32x16 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const unsigned la = (unsigned)(unsigned long long)smem + (tid & 63) * 16;
    half8 F;
    asm volatile("ds_read_b128 %0, %1" : "=v"(F) : "v"(la));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    constexpr int STEPS = 32;  // MFMAs per iteration per wave
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int c = PATTERN == 0 ? s % NCH : (s / BURST) % NCH;
            if (FILL == 1) asm volatile("s_nop 0" : "+v"(acc[c]));
            if (FILL == 2) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F) : "v"(la), "n"(1024));
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(F));
            }
            if (FILL == 3 && (PATTERN == 0 ? (s % NCH == 0) : (s % BURST == 0))) {  // fillers only at chain switches
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F) : "v"(la), "n"(1024));
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(F));
            }
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FILL >= 2 ? F : A[s & 3], B[(s >> 2) & 3], acc[c], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) m += acc[c][r];
    out[blockIdx.x * WAVES * 64 + tid] = m;
    if (tid == 0 && blockIdx.x == 7) cyc[0] = t1 - t0;
}

template <int NCH, int PATTERN, int BURST, int FILL, int WAVES>
void run(const char* name, const half8* x, float* out, unsigned long long* cyc) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned long long c = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((chain<NCH, PATTERN, BURST, FILL, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, x, out, iters, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    }
    const double mf = 32.0 * iters * (WAVES / 4);  // MFMAs per SIMD
    const double tf = 2.0 * 32 * 32 * 16 * 32.0 * iters * WAVES * 256 / (best * 1e-3) / 1e12;
    printf("%-64s %7.1f cycles per MFMA per SIMD  (%.2f GHz, %7.1f TFLOP/s)\n", name, (double)c / mf, (double)c / (best * 1e-3) / 1e9, tf);
    fflush(stdout);
}

// Operands from the accumulation half of the register file: MODE 0 A and B in VGPRs (inline asm), 1 A in AGPRs, 2 B in AGPRs, 3 both
template <int MODE, int NCH>
__global__ __launch_bounds__(256, 1) void chain_agpr(const half8* __restrict__ x, float* __restrict__ out, int iters,
                                                     unsigned long long* __restrict__ cyc) {
    const int tid = threadIdx.x;
    half8 A = x[(tid * 7) & 1023], B = x[(tid * 13 + 5) & 1023];
    asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_write_b32 a2, %2\n\tv_accvgpr_write_b32 a3, %3"
                 ::"v"(((const float*)&A)[0]), "v"(((const float*)&A)[1]), "v"(((const float*)&A)[2]), "v"(((const float*)&A)[3]) : "a0", "a1", "a2", "a3");
    asm volatile("v_accvgpr_write_b32 a4, %0\n\tv_accvgpr_write_b32 a5, %1\n\tv_accvgpr_write_b32 a6, %2\n\tv_accvgpr_write_b32 a7, %3"
                 ::"v"(((const float*)&B)[0]), "v"(((const float*)&B)[1]), "v"(((const float*)&B)[2]), "v"(((const float*)&B)[3]) : "a4", "a5", "a6", "a7");
    asm volatile("s_nop 7" ::: "memory");
    f32x16 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int c = s % NCH;
            if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(A), "v"(B));
            if (MODE == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[0:3], %1, %0" : "+v"(acc[c]) : "v"(B));
            if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[4:7], %0" : "+v"(acc[c]) : "v"(A));
            if (MODE == 3) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[0:3], a[4:7], %0" : "+v"(acc[c]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) m += acc[c][r];
    out[blockIdx.x * 256 + tid] = m;
    if (tid == 0 && blockIdx.x == 7) cyc[0] = t1 - t0;
}
template <int MODE, int NCH>
void run_agpr(const char* name, const half8* x, float* out, unsigned long long* cyc) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned long long c = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((chain_agpr<MODE, NCH>), dim3(256), dim3(256), 0, 0, x, out, iters, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    }
    const double tf = 2.0 * 32 * 32 * 16 * 32.0 * iters * 4 * 256 / (best * 1e-3) / 1e12;
    printf("%-64s %7.1f cycles per MFMA per SIMD  (%.2f GHz, %7.1f TFLOP/s)\n", name, (double)c / (32.0 * iters), (double)c / (best * 1e-3) / 1e9, tf);
    fflush(stdout);
}

int main() {
    std::vector<_Float16> h(1024 * 8);
    srand(3);
    for (auto& v : h) {  // ~N(0, 1/768): sum of 12 uniforms
        float s = 0;
        for (int i = 0; i < 12; ++i) s += (float)rand() / RAND_MAX;
        v = (_Float16)((s - 6.0f) * 0.036f);
    }
    half8* x;
    float* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&x, h.size() * 2));
    CHECK(hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, 256 * 512 * 4));
    CHECK(hipMalloc(&cyc, 8));
    printf("one wave per SIMD:\n");
    run<1, 0, 1, 0, 4>("1 chain, back to back", x, out, cyc);
    run<1, 0, 1, 1, 4>("1 chain, one s_nop between", x, out, cyc);
    run<1, 0, 1, 2, 4>("1 chain, ds_read + wait between", x, out, cyc);
    run<2, 0, 1, 0, 4>("2 chains alternating", x, out, cyc);
    run<2, 0, 1, 1, 4>("2 chains alternating, one s_nop between", x, out, cyc);
    run<2, 0, 1, 2, 4>("2 chains alternating, ds_read + wait between", x, out, cyc);
    run<2, 1, 4, 0, 4>("2 chains in bursts of 4", x, out, cyc);
    run<2, 1, 4, 3, 4>("2 chains in bursts of 4, ds_read + wait at the switches", x, out, cyc);
    run<2, 1, 8, 3, 4>("2 chains in bursts of 8, ds_read + wait at the switches", x, out, cyc);
    run<3, 0, 1, 2, 4>("3 chains alternating, ds_read + wait between", x, out, cyc);
    run<4, 0, 1, 0, 4>("4 chains alternating", x, out, cyc);
    run<4, 0, 1, 2, 4>("4 chains alternating, ds_read + wait between", x, out, cyc);
    run<8, 0, 1, 2, 4>("8 chains alternating, ds_read + wait between", x, out, cyc);
    printf("inline-asm MFMAs, one wave per SIMD, operands by register file half:\n");
    run_agpr<0, 2>("2 chains, A and B in VGPRs", x, out, cyc);
    run_agpr<1, 2>("2 chains, A in AGPRs", x, out, cyc);
    run_agpr<2, 2>("2 chains, B in AGPRs", x, out, cyc);
    run_agpr<3, 2>("2 chains, A and B in AGPRs", x, out, cyc);
    run_agpr<3, 1>("1 chain, A and B in AGPRs", x, out, cyc);
    run_agpr<3, 4>("4 chains, A and B in AGPRs", x, out, cyc);
    run_agpr<0, 1>("1 chain, A and B in VGPRs", x, out, cyc);
    printf("two waves per SIMD:\n");
    run<1, 0, 1, 0, 8>("1 chain, back to back", x, out, cyc);
    run<1, 0, 1, 1, 8>("1 chain, one s_nop between", x, out, cyc);
    run<1, 0, 1, 2, 8>("1 chain, ds_read + wait between", x, out, cyc);
    run<2, 0, 1, 2, 8>("2 chains alternating, ds_read + wait between", x, out, cyc);
    run<2, 1, 4, 3, 8>("2 chains in bursts of 4, ds_read + wait at the switches", x, out, cyc);
    run<4, 0, 1, 2, 8>("4 chains alternating, ds_read + wait between", x, out, cyc);
    run<8, 0, 1, 2, 8>("8 chains alternating, ds_read + wait between", x, out, cyc);
    return 0;
}

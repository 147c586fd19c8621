// Development probe (not part of the product): what HBM rate does the STAGING PATTERN of the tile kernels get when one workgroup
// per CU streams a corpus that nothing else shares (the 97 .. 1 024-query launches: one or two query tiles, every corpus byte
// comes from HBM)?  A 256-row x 768-dimension fp16 tile is one contiguous 384 KB block; the tile kernels walk it K-step by
// K-step (128 B of each of the 256 rows = 32 KB per step, twelve steps), which returns to every DRAM page twelve times, ~2 us
// apart.  The probe reads the same bytes (a) in that order and (b) front to back, 32 KB per step either way, with 1 .. 4 steps
// of loads in flight per workgroup, and reports GB/s.  build + run: hipcc --offload-arch=gfx950 -O3 tools/probe_stage_pattern.hip
// -o /tmp/psp && /tmp/psp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

constexpr int ROW_BYTES = 1536, TILE_ROWS = 256, KSTEPS = 12, STEP_BYTES = 32768;
constexpr long long TILE_BYTES = (long long)ROW_BYTES * TILE_ROWS;

template <int PATTERN>
__device__ inline long long step_offset(int ks, int j, int t) {
    if (PATTERN == 0) {  // K-step walk: row (j * 64 + t / 8), 16 B piece t % 8 of its 128-B slice ks
        const int row = j * 64 + (t >> 3);
        return (long long)row * ROW_BYTES + ks * 128 + (t & 7) * 16;
    }
    if (PATTERN == 2) {  // K-step walk, 256 B per row and step (two K-slices at once, 128 rows per half step)
        const int row = (ks & 1) * 128 + j * 32 + (t >> 4);
        return (long long)row * ROW_BYTES + (ks >> 1) * 256 + (t & 15) * 16;
    }
    return (long long)ks * STEP_BYTES + j * 8192 + t * 16;  // front to back
}

template <int PATTERN, int DEPTH>
__global__ __launch_bounds__(512) void probe_kernel(const char* __restrict__ x, int tiles_per_wg, long long ntiles, uint4* sink) {
    extern __shared__ char smem[];  // sized by the host so that ONE workgroup fits a CU, as in the tile kernels
    const int t = threadIdx.x;
    const long long tile0 = (long long)blockIdx.x * tiles_per_wg;
    long long tile1 = tile0 + tiles_per_wg;
    if (tile1 > ntiles) tile1 = ntiles;
    if (tile0 >= tile1) return;
    const int S = (int)(tile1 - tile0) * KSTEPS;
    uint4 acc = {0, 0, 0, 0};
    uint4 buf[DEPTH][4];
    auto issue = [&](int s, uint4 (&b)[4]) {
        const char* base = x + (tile0 + s / KSTEPS) * TILE_BYTES;
        const int ks = s % KSTEPS;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *(const uint4*)(base + step_offset<PATTERN>(ks, j, t));
    };
#pragma unroll
    for (int p = 0; p < DEPTH; ++p)
        if (p < S) issue(p, buf[p]);
    for (int s0 = 0; s0 < S; s0 += DEPTH) {
#pragma unroll
        for (int p = 0; p < DEPTH; ++p) {
            const int s = s0 + p;
            if (s < S) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc.x ^= buf[p][j].x;
                    acc.y += buf[p][j].y;
                    acc.z ^= buf[p][j].z;
                    acc.w += buf[p][j].w;
                }
                if (s + DEPTH < S) issue(s + DEPTH, buf[p]);
                __syncthreads();  // the K-step barrier of the tile kernels
            }
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[blockIdx.x * 512 + t] = acc;  // never true: keeps the loads alive
    if (t == 0 && smem[0] == 77) sink[0] = acc;
}

template <int PATTERN, int DEPTH>
static void run(const char* name, const char* x, long long ntiles, int tiles_per_wg, uint4* sink, size_t lds) {
    const int grid = (int)((ntiles + tiles_per_wg - 1) / tiles_per_wg);
    CHECK(hipFuncSetAttribute((const void*)probe_kernel<PATTERN, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe_kernel<PATTERN, DEPTH>), dim3(grid), dim3(512), lds, 0, x, tiles_per_wg, ntiles, sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    const double gb = (double)ntiles * TILE_BYTES / 1e9;
    printf("%-34s steps in flight %d, %2d tiles per workgroup (%5d workgroups, %3zu KB LDS): %7.3f ms  %7.1f GB/s\n", name, DEPTH,
           tiles_per_wg, grid, lds >> 10, best, gb / (best * 1e-3));
    fflush(stdout);
}

int main() {
    const long long ntiles = 3906;  // 1 M rows x 768 fp16
    const size_t bytes = (size_t)ntiles * TILE_BYTES;
    char* x = nullptr;
    uint4* sink = nullptr;
    CHECK(hipMalloc(&x, bytes));
    CHECK(hipMalloc(&sink, (size_t)4096 * 512 * sizeof(uint4)));
    CHECK(hipMemset(x, 1, bytes));
    for (size_t lds : {(size_t)140 << 10, (size_t)72 << 10, (size_t)36 << 10}) {  // 1, 2 and 4 workgroups per CU
        for (int tpw : {16, 4}) {
            run<0, 1>("K-step walk (128 B per row)", x, ntiles, tpw, sink, lds);
            run<0, 2>("K-step walk (128 B per row)", x, ntiles, tpw, sink, lds);
            run<0, 3>("K-step walk (128 B per row)", x, ntiles, tpw, sink, lds);
            run<0, 4>("K-step walk (128 B per row)", x, ntiles, tpw, sink, lds);
            run<2, 1>("K-step walk (256 B per row)", x, ntiles, tpw, sink, lds);
            run<2, 2>("K-step walk (256 B per row)", x, ntiles, tpw, sink, lds);
            run<2, 4>("K-step walk (256 B per row)", x, ntiles, tpw, sink, lds);
            run<1, 1>("front to back", x, ntiles, tpw, sink, lds);
            run<1, 2>("front to back", x, ntiles, tpw, sink, lds);
            run<1, 3>("front to back", x, ntiles, tpw, sink, lds);
            run<1, 4>("front to back", x, ntiles, tpw, sink, lds);
        }
    }
    return 0;
}

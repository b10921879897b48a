// hbm_randread.hip -- measurement tool, not product code: what random line reads can MI355X HBM3E
// sustain in the access shape k_classify uses (G lanes x 16 B = one aligned LINE-byte line per group)?
// Gives the empirical ceiling next to the 8 TB/s spec peak, and a known-byte-count workload for
// calibrating rocprofv3's FETCH_SIZE on this access pattern (MI355X_MICROARCH.md, HBM section).
//
//   hbm_randread <table_GB> <line_bytes 64|128> <unroll 1|2|4|8> [iters] [blocks] [share]
//   share = number of ADJACENT lane groups of one wave-instruction that read the same line (locality model)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

template <int LINE, int U>
__global__ void __launch_bounds__(256) k_rand(const u32x4 *tab, uint32_t nlines, uint32_t iters, uint32_t *sink, uint32_t share) {
    constexpr int G = LINE / 16;
    const uint64_t gid = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) / G / share;
    const uint32_t sub = threadIdx.x % G;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t it = 0; it < iters; ++it) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t h = (uint32_t)(mix(gid * 0x10001ull + (uint64_t)it * U + u) >> 32);
            uint32_t line = (uint32_t)(((uint64_t)h * nlines) >> 32);
            v[u] = tab[(size_t)line * G + sub];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// lane-per-line variant: every LANE owns one 64-B line and reads all of it (4 x 16 B); `share` adjacent lanes
// own the same line.  Models a probe kernel where each lane compares a whole bucket itself.
template <int U>
__global__ void __launch_bounds__(256) k_rand_lane(const u32x4 *tab, uint32_t nlines, uint32_t iters, uint32_t *sink, uint32_t share) {
    const uint64_t gid = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) / share;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t it = 0; it < iters; ++it) {
        u32x4 v[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t h = (uint32_t)(mix(gid * 0x10001ull + (uint64_t)it * U + u) >> 32);
            uint32_t line = (uint32_t)(((uint64_t)h * nlines) >> 32);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[u][q] = tab[(size_t)line * 4 + q];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc ^= v[u][q];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int LINE, int U>
double run(const u32x4 *tab, uint32_t nlines, uint32_t iters, int blocks, uint32_t *sink, uint32_t share) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_rand<LINE, U>), dim3(blocks), dim3(256), 0, 0, tab, nlines, 2u, sink, share);   // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_rand<LINE, U>), dim3(blocks), dim3(256), 0, 0, tab, nlines, iters, sink, share);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main(int argc, char **argv) {
    double gb = argc > 1 ? atof(argv[1]) : 6.4;
    int line = argc > 2 ? atoi(argv[2]) : 64;
    int unroll = argc > 3 ? atoi(argv[3]) : 4;
    uint32_t iters = argc > 4 ? (uint32_t)atoi(argv[4]) : 256;
    int blocks = argc > 5 ? atoi(argv[5]) : 2048;
    uint32_t share = argc > 6 ? (uint32_t)atoi(argv[6]) : 1;
    const int line_b = line ? line : 64;
    size_t bytes = (size_t)(gb * 1e9) / line_b * line_b;
    uint32_t nlines = (uint32_t)(bytes / line_b);
    u32x4 *tab;
    uint32_t *sink;
    CK(hipMalloc(&tab, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(tab, 0x5A, bytes));
    CK(hipDeviceSynchronize());
    double ms = 0;
    if (line == 0) {          // lane-per-line mode: <table_GB> 0 <unroll 1|2> ...
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        auto launch = [&](uint32_t it) {
            if (unroll == 1) hipLaunchKernelGGL((k_rand_lane<1>), dim3(blocks), dim3(256), 0, 0, tab, nlines, it, sink, share);
            else hipLaunchKernelGGL((k_rand_lane<2>), dim3(blocks), dim3(256), 0, 0, tab, nlines, it, sink, share);
        };
        launch(2); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a)); launch(iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float fms = 0; CK(hipEventElapsedTime(&fms, a, b)); ms = fms;
        double probes = (double)blocks * 256 * iters * unroll;
        printf("{\"mode\": \"lane_per_line\", \"table_gb\": %.2f, \"unroll\": %d, \"share\": %u, \"ms\": %.3f, \"Gprobes_per_s\": %.2f, \"Glines_per_s\": %.2f}\n", gb, unroll, share, ms, probes / ms / 1e6, probes / share / ms / 1e6);
        return 0;
    }
#define CASE(L, U) if (line == L && unroll == U) ms = run<L, U>(tab, nlines, iters, blocks, sink, share);
    CASE(64, 1) CASE(64, 2) CASE(64, 4) CASE(64, 8) CASE(128, 1) CASE(128, 2) CASE(128, 4) CASE(128, 8) CASE(256, 1) CASE(256, 2) CASE(256, 4) CASE(512, 2)
    if (ms == 0) { fprintf(stderr, "unsupported line/unroll\n"); return 1; }
    double groups = (double)blocks * 256 / (line / 16);
    double lines = groups * iters * unroll;
    printf("{\"table_gb\": %.2f, \"line\": %d, \"unroll\": %d, \"blocks\": %d, \"share\": %u, \"lines\": %.0f, \"bytes\": %.0f, \"ms\": %.3f, \"Gprobes_per_s\": %.2f, \"Glines_per_s\": %.2f, \"GB_per_s\": %.1f}\n",
           gb, line, unroll, blocks, share, lines / share, lines / share * line, ms, lines / ms / 1e6, lines / share / ms / 1e6, lines / share * line / ms / 1e6);
    return 0;
}

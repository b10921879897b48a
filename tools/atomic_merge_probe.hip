// atomic_merge_probe.hip -- measurement tool, not product code: over how many lanes does MI355X merge the atomics of one
// wave instruction that fall into the same 32-byte sector?  Each group of G consecutive lanes (starting at lane offset S
// within the wave) adds to different dwords of ONE random sector; everything else being equal, the time per instruction
// tells how many memory-side requests the instruction became.
//   atomic_merge_probe            prints one JSON line per (G, S)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// order: 0 = lane i of a group adds to dword i, 1 = to dword 5i mod 8 (same dwords, scrambled over the lanes), 2 = all to dword 0
__global__ void __launch_bounds__(256) k_merge(uint32_t *mem, uint32_t nsectors, uint32_t iters, uint32_t G, uint32_t S, uint32_t order) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint32_t shifted = (lane + 64 - S) & 63;                 // groups start at lane S
    const uint32_t grp = shifted / G, within = shifted % G;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t sector = (uint32_t)(((mix(wave * 1000003ull + it * 64ull + grp) >> 32) * nsectors) >> 32);
        const uint32_t dw = order == 0 ? (within & 7) : order == 1 ? ((within * 5) & 7) : 0u;
        atomicAdd(mem + (size_t)sector * 8 + dw, 1u);
    }
}
int main() {
    const size_t bytes = (size_t)8 << 30;
    const uint32_t nsectors = (uint32_t)(bytes / 32);
    uint32_t *mem;
    CK(hipMalloc(&mem, bytes));
    CK(hipMemset(mem, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t iters = 64, blocks = 256 * 16;
    const int cfg[][3] = {{1, 0, 0}, {2, 0, 0}, {2, 1, 0}, {4, 0, 0}, {4, 1, 0}, {4, 2, 0}, {8, 0, 0}, {8, 4, 0}, {8, 2, 0}, {16, 0, 0}, {16, 8, 0}, {64, 0, 0},
                          {4, 0, 1}, {4, 1, 1}, {8, 0, 1}, {8, 3, 1}, {3, 0, 1}, {3, 1, 1}, {4, 0, 2}, {8, 0, 2}};
    for (int rep = 0; rep < 2; ++rep)
        for (auto &c : cfg) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_merge, dim3(blocks), dim3(256), 0, 0, mem, nsectors, iters, (uint32_t)c[0], (uint32_t)c[1], (uint32_t)c[2]);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double lane_ops = (double)blocks * 256 * iters;
            if (rep) printf("{\"group\": %d, \"start\": %d, \"order\": %d, \"ms\": %.3f, \"G_lane_atomics_per_s\": %.1f, \"G_groups_per_s\": %.1f}\n", c[0], c[1], c[2], ms,
                            lane_ops / ms / 1e6, lane_ops / c[0] / ms / 1e6);
        }
    return 0;
}

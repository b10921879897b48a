// atomics_probe.hip -- measurement tool, not product code: what do the per-read commit updates of k_classify cost on
// MI355X?  Random 64-B line reads over a 16-GB table (the probes) with, per 40 reads, one update of a random 16-B record
// in a 160-MB array, done as (a) nothing, (b) a 64-bit atomic add (what k_classify does), (c) a plain 8-byte store
// (what a log-and-apply scheme would do), (d) atomics alone, (e) stores alone.
//   atomics_probe [n_updates=16000000] [reads_per_update=40]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// mode bit 0: do the reads; bits 1-2: 0 none, 1 atomic, 2 store.  A group of 4 lanes reads one 64-B line (16 B each).
__global__ void __launch_bounds__(256) k_probe(const u32x4 *tab, uint32_t nlines, unsigned long long *rec, uint32_t nrec, uint32_t n_updates,
                                               uint32_t reads_per_update, int mode, uint32_t *sink) {
    const uint32_t groups = gridDim.x * blockDim.x / 4;
    const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) / 4, sub = threadIdx.x & 3;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t u = g; u < n_updates; u += groups) {
        if (mode & 1)
            for (uint32_t r = 0; r < reads_per_update; r += 4) {
                u32x4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t line = (uint32_t)(((mix((uint64_t)u * 64 + r + q) >> 32) * nlines) >> 32);
                    v[q] = tab[(size_t)line * 4 + sub];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) acc ^= v[q];
            }
        if (sub == 0) {
            const uint32_t id = (uint32_t)(((mix(0xABCDEFull + u) >> 32) * nrec) >> 32);
            if ((mode >> 1) == 1) atomicAdd(rec + 2 * (size_t)id, 0x100000001ull);
            else if ((mode >> 1) == 2) rec[2 * (size_t)id] = 0x100000001ull + u;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main(int argc, char **argv) {
    const uint32_t n_updates = argc > 1 ? (uint32_t)atol(argv[1]) : 16000000u, rpu = argc > 2 ? (uint32_t)atoi(argv[2]) : 40u;
    const size_t table_bytes = (size_t)16 << 30;
    const uint32_t nlines = (uint32_t)(table_bytes / 64), nrec = 10000000u;
    u32x4 *tab;
    unsigned long long *rec;
    uint32_t *sink;
    CK(hipMalloc(&tab, table_bytes));
    CK(hipMemset(tab, 1, table_bytes));
    CK(hipMalloc(&rec, (size_t)nrec * 16));
    CK(hipMemset(rec, 0, (size_t)nrec * 16));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[] = {"reads only", "reads + atomic", "reads + store", "atomics only", "stores only"};
    const int modes[] = {1, 1 | 2, 1 | 4, 2, 4};
    for (int rep = 0; rep < 2; ++rep)
        for (int i = 0; i < 5; ++i) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_probe, dim3(256 * 16), dim3(256), 0, 0, tab, nlines, rec, nrec, n_updates, rpu, modes[i], sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("{\"case\": \"%s\", \"updates\": %u, \"reads_per_update\": %u, \"ms\": %.3f}\n", names[i], n_updates, rpu, ms);
        }
    return 0;
}

// l2_atomics_probe.hip -- measurement tool, not product code.  Question behind stage 00's next design:
// can counter updates stay inside ONE XCD's L2?  Agent-scope atomics execute at the memory side on this part (every update
// is an HBM transaction, TCC_EA0_WRREQ == TCC_ATOMIC).  Here every workgroup reads the id of the XCD it runs on
// (HW_REG_XCC_ID) and adds only into that XCD's own slice of a table, with atomics of a given scope:
//     l2_atomics_probe [slice_kb=2048] [adds_per_lane=4096] [dwords_per_sector: 1 = every lane its own random dword]
// prints adds/s for agent, workgroup and wavefront scope and checks that no add was lost (sum of the table == adds issued).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

template <int SCOPE>
__global__ void __launch_bounds__(256) k_add(uint32_t *tab, uint32_t slice_dwords, uint32_t adds_per_lane, uint32_t *xcc_seen) {
    const uint32_t x = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&xcc_seen[x], 1u);
    uint32_t *mine = tab + (size_t)x * slice_dwords;
    uint32_t s = mix32(blockIdx.x * 256u + threadIdx.x + 1u);
    for (uint32_t i = 0; i < adds_per_lane; ++i) {
        s = mix32(s + i);
        const uint32_t at = (uint32_t)(((uint64_t)s * slice_dwords) >> 32);
        __hip_atomic_fetch_add(&mine[at], 1u, __ATOMIC_RELAXED, SCOPE);
    }
}

int main(int argc, char **argv) {
    const uint32_t slice_kb = argc > 1 ? (uint32_t)atoi(argv[1]) : 2048u, apl = argc > 2 ? (uint32_t)atoi(argv[2]) : 4096u;
    const uint32_t slice_dwords = slice_kb * 256u;
    uint32_t *tab, *seen;
    CK(hipMalloc(&tab, (size_t)8 * slice_dwords * 4));
    CK(hipMalloc(&seen, 8 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    const char *names[] = {"agent", "workgroup", "wavefront"};
    std::vector<uint32_t> h((size_t)8 * slice_dwords);
    for (int sc = 0; sc < 3; ++sc) {
        float best = 1e30f;
        unsigned long long sum = 0;
        uint32_t hs[8];
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(tab, 0, (size_t)8 * slice_dwords * 4));
            CK(hipMemset(seen, 0, 32));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (sc == 0) hipLaunchKernelGGL(k_add<__HIP_MEMORY_SCOPE_AGENT>, dim3(grid), dim3(256), 0, 0, tab, slice_dwords, apl, seen);
            else if (sc == 1) hipLaunchKernelGGL(k_add<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(grid), dim3(256), 0, 0, tab, slice_dwords, apl, seen);
            else hipLaunchKernelGGL(k_add<__HIP_MEMORY_SCOPE_WAVEFRONT>, dim3(grid), dim3(256), 0, 0, tab, slice_dwords, apl, seen);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        CK(hipMemcpy(h.data(), tab, h.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs, seen, 32, hipMemcpyDeviceToHost));
        for (uint32_t v : h) sum += v;
        const unsigned long long want = (unsigned long long)grid * 256 * apl;
        printf("{\"scope\": \"%s\", \"slice_kb\": %u, \"adds\": %llu, \"sum\": %llu, \"lost\": %lld, \"ms\": %.3f, \"Gadds_per_s\": %.2f, \"wgs_per_xcc\": [%u,%u,%u,%u,%u,%u,%u,%u]}\n",
               names[sc], slice_kb, want, sum, (long long)want - (long long)sum, best, want / best / 1e6, hs[0], hs[1], hs[2], hs[3], hs[4], hs[5], hs[6], hs[7]);
    }
    return 0;
}

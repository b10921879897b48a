// what a HIP stream costs a short process: creates N streams (argv[1]; argv[2] = 1: half of them CU-masked, as gz_api.cpp's pass
// streams are), puts one tiny copy on each, and leaves without tearing anything down (as `classify` does).  Prints the time of the
// runtime's start, of the stream creations and of the copies; the shell's clock around the process gives what leaving costs.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/stream_cost tools/probes/stream_cost.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4, masked = argc > 2 ? atoi(argv[2]) : 0;
    const size_t mb = argc > 3 ? (size_t)atol(argv[3]) : 0;                  // device memory held at the end, MB
    const double t0 = now();
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1) return 1;
    const double t1 = now();
    std::vector<hipStream_t> s(n);
    std::vector<uint32_t> mask(8, 0xFFFFFFFFu);
    mask[7] = 0;
    for (int i = 0; i < n; i++) {
        hipError_t e = (masked && (i & 1)) ? hipExtStreamCreateWithCUMask(&s[i], 8, mask.data()) : hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
        if (e != hipSuccess) return 2;
    }
    const double t2 = now();
    void *d = nullptr, *big = nullptr;
    char h[64] = {0};
    if (hipMalloc(&d, 4096) != hipSuccess) return 3;
    if (mb && hipMalloc(&big, mb << 20) != hipSuccess) return 3;
    for (int i = 0; i < n; i++) (void)hipMemcpyAsync(d, h, 64, hipMemcpyHostToDevice, s[i]);
    for (int i = 0; i < n; i++) (void)hipStreamSynchronize(s[i]);
    const double t3 = now();
    printf("streams %2d masked %d held %5zu MB: runtime start %.3f s, streams %.3f s (%.1f ms each), first copies %.3f s, in main %.3f s\n", n, masked, mb, t1 - t0, t2 - t1,
           n ? (t2 - t1) * 1e3 / n : 0.0, t3 - t2, t3 - t0);
    fflush(stdout);
    _exit(0);
}

// hipstall -- which HIP runtime call makes a process wait?  ("name the stalls".)
//
// An LD_PRELOAD interposer over the handful of HIP runtime entry points libhast.so uses: every call is timed; a call that takes
// longer than HIPSTALL_MS (default 20) is printed at once --
//     __hipstall__ t=0.412 s thread=3 hipHostMalloc 87.3 ms (arg 16781312)
// -- and when the process leaves, a table of calls / total / longest per entry point (`__hipstall_total__` lines) and the time
// from the first call to the table.  Measurement tooling only: nothing in the product links or loads it.
//     g++ -O2 -shared -fPIC -o tools/hipstall.so tools/hipstall.cpp -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -ldl
//     LD_PRELOAD=tools/hipstall.so HIPSTALL_MS=20 hast_amd/classify ... 2> log
// The program under it must not replace itself (exec) after its first HIP call -- the same rule as everywhere on this pool.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <mutex>

namespace {
using clk = std::chrono::steady_clock;
const clk::time_point g_t0 = clk::now();
double g_limit_ms = [] { const char *e = getenv("HIPSTALL_MS"); return e ? atof(e) : 20.0; }();
std::atomic<int> g_threads{0};
thread_local int t_id = -1;
struct Site {
    const char *name;
    std::atomic<unsigned long long> calls{0}, ns{0}, worst{0};
    Site *next;
    explicit Site(const char *n);
};
Site *g_sites = nullptr;
std::mutex g_mu;
Site::Site(const char *n) : name(n), next(nullptr) {
    std::lock_guard<std::mutex> l(g_mu);
    next = g_sites;
    g_sites = this;
}
struct Timed {
    Site &s;
    unsigned long long arg;
    clk::time_point a;
    Timed(Site &site, unsigned long long x) : s(site), arg(x), a(clk::now()) {}
    ~Timed() {
        const clk::time_point b = clk::now();
        const unsigned long long ns = (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count();
        s.calls++;
        s.ns += ns;
        unsigned long long w = s.worst.load();
        while (ns > w && !s.worst.compare_exchange_weak(w, ns)) {}
        if (ns * 1e-6 >= g_limit_ms) {
            if (t_id < 0) t_id = g_threads++;
            fprintf(stderr, "__hipstall__ t=%.3f s thread=%d %s %.1f ms (arg %llu)\n", std::chrono::duration<double>(a - g_t0).count(), t_id, s.name, ns * 1e-6, arg);
        }
    }
};
struct AtExit {
    ~AtExit() {
        for (Site *s = g_sites; s; s = s->next)
            if (s->calls) fprintf(stderr, "__hipstall_total__ %-32s calls %8llu total %9.2f ms longest %8.2f ms\n", s->name, s->calls.load(), s->ns * 1e-6, s->worst * 1e-6);
        fprintf(stderr, "__hipstall_total__ the table is printed %.3f s after this library was loaded\n", std::chrono::duration<double>(clk::now() - g_t0).count());
    }
} g_at_exit;
template <class F>
F next_of(const char *name) {
    void *p = dlsym(RTLD_NEXT, name);
    if (!p) {
        fprintf(stderr, "hipstall: %s not found behind this library\n", name);
        abort();
    }
    return reinterpret_cast<F>(p);
}
}  // namespace

#define HIPSTALL(name, params, args, argword)                            \
    extern "C" hipError_t name params {                                  \
        static auto real = next_of<hipError_t(*) params>(#name);         \
        static Site site(#name);                                         \
        Timed t(site, (unsigned long long)(argword));                    \
        return real args;                                                \
    }

HIPSTALL(hipGetDeviceCount, (int *n), (n), 0)
HIPSTALL(hipSetDevice, (int d), (d), d)
HIPSTALL(hipMalloc, (void **p, size_t n), (p, n), n)
HIPSTALL(hipFree, (void *p), (p), 0)
HIPSTALL(hipHostMalloc, (void **p, size_t n, unsigned int f), (p, n, f), n)
HIPSTALL(hipHostFree, (void *p), (p), 0)
HIPSTALL(hipHostRegister, (void *p, size_t n, unsigned int f), (p, n, f), n)
HIPSTALL(hipHostUnregister, (void *p), (p), 0)
HIPSTALL(hipMemGetInfo, (size_t *a, size_t *b), (a, b), 0)
HIPSTALL(hipMemcpy, (void *d, const void *s, size_t n, hipMemcpyKind k), (d, s, n, k), n)
HIPSTALL(hipMemcpyAsync, (void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st), (d, s, n, k, st), n)
HIPSTALL(hipMemcpyPeer, (void *d, int dd, const void *s, int sd, size_t n), (d, dd, s, sd, n), n)
HIPSTALL(hipMemcpyPeerAsync, (void *d, int dd, const void *s, int sd, size_t n, hipStream_t st), (d, dd, s, sd, n, st), n)
HIPSTALL(hipMemset, (void *d, int v, size_t n), (d, v, n), n)
HIPSTALL(hipMemsetAsync, (void *d, int v, size_t n, hipStream_t st), (d, v, n, st), n)
HIPSTALL(hipStreamCreateWithFlags, (hipStream_t * s, unsigned int f), (s, f), f)
HIPSTALL(hipExtStreamCreateWithCUMask, (hipStream_t * s, uint32_t n, const uint32_t *m), (s, n, m), n)
HIPSTALL(hipStreamDestroy, (hipStream_t s), (s), 0)
HIPSTALL(hipStreamSynchronize, (hipStream_t s), (s), 0)
HIPSTALL(hipStreamWaitEvent, (hipStream_t s, hipEvent_t e, unsigned int f), (s, e, f), 0)
HIPSTALL(hipEventCreateWithFlags, (hipEvent_t * e, unsigned int f), (e, f), f)
HIPSTALL(hipEventRecord, (hipEvent_t e, hipStream_t s), (e, s), 0)
HIPSTALL(hipEventSynchronize, (hipEvent_t e), (e), 0)
HIPSTALL(hipEventDestroy, (hipEvent_t e), (e), 0)
HIPSTALL(hipDeviceSynchronize, (void), (), 0)
HIPSTALL(hipLaunchKernel, (const void *f, dim3 g, dim3 b, void **a, size_t sh, hipStream_t st), (f, g, b, a, sh, st), (unsigned long long)g.x)
HIPSTALL(hipFuncSetAttribute, (const void *f, hipFuncAttribute a, int v), (f, a, v), v)
HIPSTALL(hipPointerGetAttributes, (hipPointerAttribute_t * a, const void *p), (a, p), 0)
HIPSTALL(hipDeviceCanAccessPeer, (int *c, int a, int b), (c, a, b), b)
HIPSTALL(hipDeviceEnablePeerAccess, (int d, unsigned int f), (d, f), d)
HIPSTALL(hipGetDevicePropertiesR0600, (hipDeviceProp_tR0600 * p, int d), (p, d), d)
HIPSTALL(hipEventCreate, (hipEvent_t * e), (e), 0)

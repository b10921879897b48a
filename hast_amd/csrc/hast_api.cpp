// hast_api.cpp -- the C ABI of include/hast.h: context, table, counters, classify, synthetic data.
// Host C++ over the HIP runtime; all device work is in hast_kernels.hip.  No CPU fallback exists:
// every compute entry point needs a context, and a context needs a GPU.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/hast.h"
#include "hast_common.h"
#include "hast_device.h"
#include "kc_device.h"
#include "fq_device.h"
#include "hast_internal.h"
#include "worker_pool.h"

using namespace hast;

namespace {

thread_local char g_err[512] = "";

hast_status fail(hast_status st, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return st;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? HAST_ERR_OOM : HAST_ERR_HIP, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                                \
    } while (0)

struct Staging {          // one pinned+device buffer set of hast_classify_batch's double buffer
    uint8_t *h_bases = nullptr, *d_bases = nullptr;
    uint64_t *h_off = nullptr, *d_off = nullptr;
    uint32_t *h_ids = nullptr, *d_ids = nullptr;
    size_t cap_bases = 0, cap_reads = 0;
    hipEvent_t done = nullptr;
    bool in_flight = false;
};

}  // namespace

struct hast_ctx {
    int device = 0;
    int k = 0;
    int m = 0;                  // minimizer length used for bucket placement (fixed once a table exists)
    int n_cu = 256;
    hipStream_t stream = nullptr;
    // table
    uint64_t *d_slots = nullptr;
    uint32_t nbuckets = 0;
    // counters
    unsigned long long *d_counts = nullptr;  // [n_barcodes][4] = {c0, c1, neg, reserved}, 64-bit words
    uint32_t *h_errword = nullptr;           // pinned: where d_err[1] is read to
    unsigned long long *d_pack = nullptr;    // the three live words as arrays c0[n] | c1[n] | neg[n]: what is all-reduced and read back
    size_t pack_words = 0;
    size_t n_barcodes = 0;
    bool counts_owned = false;
    // small scratch
    uint32_t *d_err = nullptr;              // [4]
    unsigned long long *d_cnt = nullptr;    // [8]: [0..1] set sizes, [2] sink of measurement kernels, [3] tile queue of k_classify, [4] segment counter
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    Staging stage[2];
    unsigned batch_no = 0;
    // per-read mode: segment table of long reads
    void *d_seg = nullptr;
    size_t seg_bytes = 0;
    // optional kernel timing (hast_classify_timing): ring of event triplets {before k_classify, between, after k_commit_votes}
    std::vector<hipEvent_t> t_ev;
    size_t t_slots = 0, t_next = 0, t_count = 0;
    // per-read votes of the last barcode-mode launch when the caller gave no buffer for them
    uint32_t *d_votes_scratch = nullptr;
    size_t votes_bytes = 0;
    // scratch of the partitioned commit (bins of compressed records + overflow list)
    void *d_part = nullptr;
    size_t part_bytes = 0;
    // stream_file_region: two pinned pieces, their device twins, "piece is on the device" events
    char *sf_h[2] = {nullptr, nullptr}, *sf_d[2] = {nullptr, nullptr};
    hipEvent_t sf_done[2] = {nullptr, nullptr};
    // fingerprint filter in front of the table (hast_common.h): rebuilt from the table's live keys before the first
    // classification after the table gained keys
    void *d_filter = nullptr;
    size_t filter_bytes = 0;
    FilterGeom fg{};
    bool filter_valid = false;
    bool use_filter = true;                 // HAST_CLASSIFY=exact: probe the exact table directly (the round-1 kernel)
    size_t filter_fallback_bytes = 0;       // != 0: the filter was wanted, this many bytes could not be had (also after the parked memory was
                                            // freed), and the context probes the table directly -- hast_ctx_options / --stats say so
    bool test_filter_oom = false;           // HAST_TEST_FILTER_OOM=1 (tests): the filter's allocation fails
    int filter_m = 0, filter_t = 0, filter_kp = 0;   // overrides (0 = by K and key count)
    int filter_exact = -1;                           // -1: exact entries where they fit (hast_common.h), 0: prints always
    int text_acgt_only = 0;                          // k-mer text lines must be upper-case A/C/G/T (hast_ctx_set_text_check)
    bool exact_env_off = false;                      // HAST_FILTER_EXACT=0 in the environment
    // measurement switches: read from the environment ONCE, when the context is created (hast_ctx_set_option changes them on a
    // live context); a variable that appears in a user's shell later cannot re-route a running job
    int commit_mode = 0;                             // HAST_COMMIT: 0 by batch size, 1 = one atomic per read, 2 = partitioned
    int kernel_geo = 1, kernel_rl = 1;               // HAST_F_GEO / HAST_F_RL = 0: the generic k_classify_f instantiations
    size_t tile_lds = 0;                             // HAST_TILE_LDS: LDS budget of a tile (0 = default)
    bool part_oom = false;                           // the partitioned commit's scratch did not fit once: atomics from then on
};

namespace {

TableGeom geom(const hast_ctx *c) { return TableGeom{c->nbuckets, c->k, c->m, c->k == 32}; }
size_t table_slots(const hast_ctx *c) { return (size_t)c->nbuckets * kSlotsPerBucket * (c->k == 32 ? 2 : 1); }

// default minimizer length: w = K-m+1 consecutive windows can share a bucket line; m stays >= 16 so that
// the minimizer space (4^m/2 = 2.1 G) is well above human-scale key counts (4e8) and buckets stay evenly
// loaded; measured sweep in DESIGN.md
int default_minimizer_plain(int k) { return k <= 16 ? k : std::max(16, k - 8); }
int default_minimizer(int k) {
    if (const char *e = getenv("HAST_MINIMIZER")) {
        int v = atoi(e);
        if (v >= 1 && v <= k) return v;
    }
    return default_minimizer_plain(k);
}

hast_status use(hast_ctx *c) {
    if (!c) return fail(HAST_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    return HAST_OK;
}

hast_status ensure_scratch(hast_ctx *c, size_t bytes) {
    if (c->scratch_bytes >= bytes) return HAST_OK;
    if (c->d_scratch) HIP_TRY(hipFree(c->d_scratch));
    c->d_scratch = nullptr;
    c->scratch_bytes = 0;
    HIP_TRY(dev_malloc(&c->d_scratch, bytes));
    c->scratch_bytes = bytes;
    return HAST_OK;
}

constexpr size_t kChunkBytes = 64u << 20;   // staging granule for table input

hast_status check_err_word(hast_ctx *c, hipStream_t s) {
    uint32_t e = 0;
    HIP_TRY(hipMemcpyAsync(&e, c->d_err, sizeof(e), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (e) HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(uint32_t), s));
    if (e & 2) return fail(HAST_ERR_FORMAT, "k-mer text: a line is not exactly K=%d bytes", c->k);
    if (e & 4) return fail(HAST_ERR_FORMAT, "k-mer text: a byte other than upper-case A/C/G/T");
    if (e & 1) return fail(HAST_ERR_TABLE_FULL, "k-mer table full: reserve more keys");
    return HAST_OK;
}

// d_err[1]: raised by kernels of the (asynchronous) classification path; looked at wherever the caller waits for results
// (read into a pinned word on the stream the caller has just waited for: a pageable hipMemcpy goes through the NULL stream, which
// synchronises with every blocking stream of the device -- ADVICE r4)
hast_status check_classify_err(hast_ctx *c, hipStream_t hs) {
    if (!hs) hs = c->stream;
    *c->h_errword = 0;
    HIP_TRY(hipMemcpyAsync(c->h_errword, c->d_err + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, hs));
    HIP_TRY(hipStreamSynchronize(hs));
    if (!*c->h_errword) return HAST_OK;
    HIP_TRY(hipMemsetAsync(c->d_err + 1, 0, sizeof(uint32_t), hs));
    HIP_TRY(hipStreamSynchronize(hs));
    return fail(HAST_ERR_INVALID, "read offsets overlap or are not monotonic (their lengths add up to more than bases_bytes): rows were dropped");
}

SynthParams resolve(const hast_synth_params *p) {
    SynthParams r;
    r.seed_k = p->seed_k ? p->seed_k : 0x4841535401ull;
    r.seed_r = p->seed_r ? p->seed_r : 0x4841535402ull;
    r.seed_b = p->seed_b ? p->seed_b : 0x4841535403ull;
    r.n_keys_per_hap = p->n_keys_per_hap;
    r.n_barcodes = p->n_barcodes ? p->n_barcodes : 1;
    r.read_len = p->read_len;
    r.k = p->k;
    r.reserved = p->reserved;      // 0 = random keys, 1 = clustered keys (runs of K around variant sites)
    return r;
}

hast_status check_synth(const hast_synth_params *p) {
    if (!p) return fail(HAST_ERR_INVALID, "null synth params");
    if (p->k < 1 || p->k > 32) return fail(HAST_ERR_INVALID, "synth k=%u out of [1,32]", p->k);
    return HAST_OK;
}

}  // namespace

// hooks for the other translation units of the library (hast_internal.h)
namespace hast {
hast_status set_error(hast_status st, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return st;
}
namespace {
struct Parked {
    std::mutex mu;
    std::vector<void *> device, pinned;
    size_t bytes = 0;
};
Parked &parked() {
    static Parked *p = new Parked();                       // (never destroyed: streams may be closed from threads that outlive main's statics)
    return *p;
}
size_t park_limit() {                                      // (the environment is read once, like every other switch: INTEGRATION.md)
    static const size_t limit = [] {
        const char *e = getenv("HAST_PARK_GB");
        return (size_t)((e ? atof(e) : 32.0) * 1073741824.0);
    }();
    return limit;
}
int park_sites() {
    static const int sites = [] {
        const char *e = getenv("HAST_PARK_SITES");
        return e ? atoi(e) : -1;
    }();
    return sites;
}
}  // namespace
size_t parked_bytes() {
    Parked &pk = parked();
    std::lock_guard<std::mutex> lk(pk.mu);
    return pk.bytes;
}
hipError_t dev_malloc(void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && parked_bytes()) {          // (what closed streams left parked may be exactly what is missing)
        (void)hipGetLastError();
        release_parked();
        e = hipMalloc(p, bytes);
    }
    return e;
}
hipError_t pinned_malloc(void **p, size_t bytes, unsigned flags) {
    hipError_t e = hipHostMalloc(p, bytes, flags);
    if (e == hipErrorOutOfMemory && parked_bytes()) {
        (void)hipGetLastError();
        release_parked();
        e = hipHostMalloc(p, bytes, flags);
    }
    return e;
}
void release_parked() {
    std::vector<void *> d, h;
    {
        Parked &pk = parked();
        std::lock_guard<std::mutex> lk(pk.mu);
        d.swap(pk.device);
        h.swap(pk.pinned);
        pk.bytes = 0;
    }
    for (void *p : d) (void)hipFree(p);
    for (void *p : h) (void)hipHostFree(p);
}
static void park(void *p, size_t bytes, bool pinned, int site) {
    if (!p) return;
    size_t limit = park_limit();
    if (!((park_sites() >> site) & 1)) limit = 0;              // (HAST_PARK_SITES, bisecting: only these sites park)
    if (limit == 0) {
        (void)(pinned ? hipHostFree(p) : hipFree(p));
        return;
    }
    bool over = false;
    {
        Parked &pk = parked();
        std::lock_guard<std::mutex> lk(pk.mu);
        (pinned ? pk.pinned : pk.device).push_back(p);
        pk.bytes += bytes;
        over = pk.bytes > limit;
    }
    if (over) release_parked();
}
void park_device(void *p, size_t bytes, int site) { park(p, bytes, false, site); }
void park_pinned(void *p, size_t bytes, int site) { park(p, bytes, true, site); }
int default_minimizer_for(int k) { return default_minimizer(k); }
}  // namespace hast

extern "C" {

const char *hast_version(void) { return "hast-mi355x 0.1 (gfx950)"; }
const char *hast_last_error(void) { return g_err; }


// ---------------------------------------------------------------------------------------------
hast_status hast_ctx_create(int device, int k, hast_ctx **out) {
    if (!out) return fail(HAST_ERR_INVALID, "out is null");
    *out = nullptr;
    if (k < 1 || k > 32) return fail(HAST_ERR_INVALID, "K=%d out of [1,32]", k);
    // HAST_TRACE_INIT=1: which call of the context's creation a box makes wait (0.07 - 0.26 s from run to run on one box)
    const bool trace = getenv("HAST_TRACE_INIT") != nullptr;
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_tr = now_s();
    auto tr = [&](const char *what) {
        if (trace) fprintf(stderr, "__trace_init__ device %d: %s %.4f s\n", device, what, now_s() - t_tr);
        t_tr = now_s();
    };
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    tr("hipGetDeviceCount (loads the runtime)");
    if (e != hipSuccess || n <= 0)
        return fail(HAST_ERR_NO_DEVICE, "no HIP device (%s); libhast has no CPU path", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(HAST_ERR_NO_DEVICE, "device %d not in [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    tr("hipSetDevice");
    hast_ctx *c = new (std::nothrow) hast_ctx();
    if (!c) return fail(HAST_ERR_OOM, "host allocation failed");
    c->device = device;
    c->k = k;
    c->m = default_minimizer(k);
    if (const char *e = getenv("HAST_CLASSIFY")) c->use_filter = strcmp(e, "exact") != 0;
    if (const char *e = getenv("HAST_TEST_FILTER_OOM")) c->test_filter_oom = atoi(e) != 0;
    if (const char *e = getenv("HAST_FILTER_M")) c->filter_m = atoi(e);
    if (const char *e = getenv("HAST_FILTER_T")) c->filter_t = atoi(e);
    if (const char *e = getenv("HAST_FILTER_KP")) c->filter_kp = atoi(e);
    if (const char *e = getenv("HAST_FILTER_EXACT")) c->exact_env_off = atoi(e) == 0;
    if (c->exact_env_off) c->filter_exact = 0;
    if (const char *e = getenv("HAST_COMMIT")) c->commit_mode = !strcmp(e, "atomic") ? 1 : !strcmp(e, "partition") ? 2 : 0;
    if (const char *e = getenv("HAST_F_GEO")) c->kernel_geo = e[0] != '0';
    if (const char *e = getenv("HAST_F_RL")) c->kernel_rl = e[0] != '0';
    if (const char *e = getenv("HAST_TILE_LDS")) {
        const long v = atol(e);
        if (v >= 4096 && v <= 160 * 1024) c->tile_lds = (size_t)v;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    tr("hipGetDeviceProperties");
    hast_status st = HAST_OK;
    auto bail = [&](hipError_t he, const char *what) {
        if (he != hipSuccess && st == HAST_OK) st = fail(HAST_ERR_HIP, "%s: %s", what, hipGetErrorString(he));
    };
    bail(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate");
    tr("hipStreamCreate");
    bail(dev_malloc(&c->d_err, 4 * sizeof(uint32_t)), "dev_malloc(err)");
    tr("first hipMalloc");
    bail(dev_malloc(&c->d_cnt, 8 * sizeof(unsigned long long)), "dev_malloc(cnt)");
    bail(pinned_malloc(reinterpret_cast<void **>(&c->h_errword), 64, hipHostMallocDefault), "pinned_malloc(err word)");
    if (st == HAST_OK) bail(hipMemsetAsync(c->d_err, 0, 4 * sizeof(uint32_t), c->stream), "hipMemset");
    if (st == HAST_OK) bail(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    tr("first fill + synchronize");
    for (auto &s : c->stage)
        if (st == HAST_OK) bail(hipEventCreateWithFlags(&s.done, hipEventDisableTiming), "hipEventCreate");
    tr("events");
    if (st != HAST_OK) {
        hast_ctx_destroy(c);
        return st;
    }
    *out = c;
    return HAST_OK;
}

void hast_release_parked(void) { release_parked(); }

void hast_ctx_destroy(hast_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    release_parked();                                      // (what closed streams left behind: a context that goes returns its memory)
    for (auto &s : c->stage) {
        if (s.h_bases) (void)hipHostFree(s.h_bases);
        if (s.h_off) (void)hipHostFree(s.h_off);
        if (s.h_ids) (void)hipHostFree(s.h_ids);
        if (s.d_bases) (void)hipFree(s.d_bases);
        if (s.d_off) (void)hipFree(s.d_off);
        if (s.d_ids) (void)hipFree(s.d_ids);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    if (c->d_slots) (void)hipFree(c->d_slots);
    if (c->counts_owned && c->d_counts) (void)hipFree(c->d_counts);
    if (c->d_pack) (void)hipFree(c->d_pack);
    if (c->h_errword) (void)hipHostFree(c->h_errword);
    if (c->d_err) (void)hipFree(c->d_err);
    if (c->d_cnt) (void)hipFree(c->d_cnt);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->d_seg) (void)hipFree(c->d_seg);
    if (c->d_votes_scratch) (void)hipFree(c->d_votes_scratch);
    if (c->d_part) (void)hipFree(c->d_part);
    for (int i = 0; i < 2; ++i) {
        if (c->sf_done[i]) (void)hipEventDestroy(c->sf_done[i]);
        if (c->sf_d[i]) (void)hipFree(c->sf_d[i]);
        if (c->sf_h[i]) (void)hipHostFree(c->sf_h[i]);
    }
    if (c->d_filter) (void)hipFree(c->d_filter);
    for (hipEvent_t e : c->t_ev) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int hast_ctx_k(const hast_ctx *c) { return c ? c->k : 0; }
int hast_ctx_minimizer(const hast_ctx *c) { return c ? c->m : 0; }

hast_status hast_ctx_set_minimizer(hast_ctx *c, int m) {
    if (!c) return fail(HAST_ERR_INVALID, "null context");
    if (m < 1 || m > c->k) return fail(HAST_ERR_INVALID, "minimizer length %d out of [1,%d]", m, c->k);
    if (c->d_slots) return fail(HAST_ERR_INVALID, "minimizer length is fixed once the table exists");
    c->m = m;
    return HAST_OK;
}
int hast_ctx_device(const hast_ctx *c) { return c ? c->device : -1; }
hast_stream hast_ctx_stream(const hast_ctx *c) { return c ? (hast_stream)c->stream : nullptr; }

hast_status hast_stream_sync(hast_ctx *c, hast_stream s) {
    if (hast_status st = use(c)) return st;
    HIP_TRY(hipStreamSynchronize(s ? (hipStream_t)s : c->stream));
    return check_classify_err(c, s ? (hipStream_t)s : c->stream);
}

// ---------------------------------------------------------------------------------------------
hast_status hast_dev_alloc(hast_ctx *c, size_t bytes, void **d_out) {
    if (hast_status st = use(c)) return st;
    if (!d_out) return fail(HAST_ERR_INVALID, "d_out is null");
    HIP_TRY(dev_malloc(d_out, bytes ? bytes : 1));
    return HAST_OK;
}
hast_status hast_dev_free(hast_ctx *c, void *p) {
    if (hast_status st = use(c)) return st;
    if (p) HIP_TRY(hipFree(p));
    return HAST_OK;
}
// free / total memory of the context's device as the runtime sees it, and what the library holds parked (hast_release_parked)
hast_status hast_dev_mem_info(hast_ctx *c, size_t *free_bytes, size_t *total_bytes, size_t *parked) {
    if (hast_status st = use(c)) return st;
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    if (parked) *parked = parked_bytes();
    return HAST_OK;
}
hast_status hast_memcpy_h2d(hast_ctx *c, void *d, const void *s, size_t n) {
    if (hast_status st = use(c)) return st;
    if (n) HIP_TRY(hipMemcpy(d, s, n, hipMemcpyHostToDevice));
    return HAST_OK;
}
hast_status hast_memcpy_d2h(hast_ctx *c, void *d, const void *s, size_t n) {
    if (hast_status st = use(c)) return st;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n) HIP_TRY(hipMemcpy(d, s, n, hipMemcpyDeviceToHost));
    return HAST_OK;
}
hast_status hast_memset_d(hast_ctx *c, void *d, int byte, size_t n, hast_stream s) {
    if (hast_status st = use(c)) return st;
    if (n) HIP_TRY(hipMemsetAsync(d, byte, n, s ? (hipStream_t)s : c->stream));
    return HAST_OK;
}

// ---------------------------------------------------------------------------------------------
hast_status hast_table_reserve(hast_ctx *c, uint64_t max_keys, double lf) {
    if (hast_status st = use(c)) return st;
    if (lf <= 0) lf = 0.2;
    if (lf > 0.9) return fail(HAST_ERR_INVALID, "load factor %.3f > 0.9", lf);
    double want = (double)(max_keys ? max_keys : 1) / lf / kSlotsPerBucket;
    uint64_t nb = (uint64_t)want + 1;
    if (nb < 64) nb = 64;
    nb += nb & 1;                                   // even: buckets may be used in pairs
    if (nb >= (1ull << 32)) return fail(HAST_ERR_INVALID, "table of %llu buckets exceeds 2^32", (unsigned long long)nb);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_slots) HIP_TRY(hipFree(c->d_slots));
    c->d_slots = nullptr;
    c->nbuckets = 0;
    c->filter_valid = false;
    // K == 32: two tag-less tables (one per haplotype) back to back, each sized for all the keys
    size_t bytes = (size_t)nb * kSlotsPerBucket * sizeof(uint64_t) * (c->k == 32 ? 2 : 1);
    HIP_TRY(dev_malloc(&c->d_slots, bytes));
    HIP_TRY(hipMemsetAsync(c->d_slots, 0xFF, bytes, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->nbuckets = (uint32_t)nb;
    return HAST_OK;
}

static hast_status ensure_filter(hast_ctx *c, hipStream_t hs);
// every entry point that adds keys calls this: the filter no longer covers the table
static void table_changed(hast_ctx *c) { c->filter_valid = false; }

static hast_status need_table(hast_ctx *c, int hap) {
    if (hast_status st = use(c)) return st;
    if (!c->d_slots) return fail(HAST_ERR_INVALID, "call hast_table_reserve first");
    if (hap != 0 && hap != 1) return fail(HAST_ERR_INVALID, "hap must be 0 or 1");
    return HAST_OK;
}

hast_status hast_ctx_set_text_check(hast_ctx *c, int acgt_only) {
    if (!c) return fail(HAST_ERR_INVALID, "null context");
    c->text_acgt_only = acgt_only ? 1 : 0;
    return HAST_OK;
}

hast_status hast_table_insert_keys_device(hast_ctx *c, int hap, const uint64_t *d_keys, size_t n, hast_stream s) {
    if (hast_status st = need_table(c, hap)) return st;
    hipStream_t hs = s ? (hipStream_t)s : c->stream;
    table_changed(c);
    HIP_TRY(launch_insert_keys(c->d_slots, geom(c), d_keys, n, (uint32_t)hap, c->d_err, hs));
    return check_err_word(c, hs);
}

hast_status hast_table_insert_keys(hast_ctx *c, int hap, const uint64_t *keys, size_t n) {
    if (hast_status st = need_table(c, hap)) return st;
    if (n && !keys) return fail(HAST_ERR_INVALID, "keys is null");
    table_changed(c);
    const size_t per = kChunkBytes / sizeof(uint64_t);
    if (hast_status st = ensure_scratch(c, std::min(n, per) * sizeof(uint64_t) + 16)) return st;
    for (size_t i = 0; i < n; i += per) {
        size_t m = std::min(per, n - i);
        HIP_TRY(hipMemcpyAsync(c->d_scratch, keys + i, m * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(launch_insert_keys(c->d_slots, geom(c), (const uint64_t *)c->d_scratch, m, (uint32_t)hap, c->d_err, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return check_err_word(c, c->stream);
}

hast_status hast_table_insert_text(hast_ctx *c, int hap, const char *text, size_t nbytes, uint64_t *lines_out) {
    if (hast_status st = need_table(c, hap)) return st;
    if (lines_out) *lines_out = 0;
    if (nbytes && !text) return fail(HAST_ERR_INVALID, "text is null");
    table_changed(c);
    const size_t stride = (size_t)c->k + 1;
    // classify.cpp:41: pieces are '\n'-terminated lines; a trailing piece without '\n' is dropped.
    // With fixed-width lines that is floor(nbytes/stride) lines, provided every line really is K bytes
    // (checked on the device); a ragged file fails the check (the reference asserts, kmer.h:154).
    size_t n_lines = nbytes / stride;
    size_t rem = nbytes - n_lines * stride;
    // the dropped tail must itself not contain a newline (else lines are ragged)
    if (rem && memchr(text + n_lines * stride, '\n', rem)) return fail(HAST_ERR_FORMAT, "k-mer text: ragged last line");
    const size_t per = (kChunkBytes / stride);
    if (hast_status st = ensure_scratch(c, std::min(n_lines, per) * stride + 16)) return st;
    for (size_t i = 0; i < n_lines; i += per) {
        size_t m = std::min(per, n_lines - i);
        HIP_TRY(hipMemcpyAsync(c->d_scratch, text + i * stride, m * stride, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(launch_insert_text(c->d_slots, geom(c), (const char *)c->d_scratch, m, (uint32_t)hap, c->text_acgt_only, c->d_err, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (hast_status st = check_err_word(c, c->stream)) return st;
    if (lines_out) *lines_out = n_lines;
    return HAST_OK;
}

// A region of a regular file through the device, piece by piece: eight threads pread the next piece into pinned memory while
// the previous one is copied to the device and consumed by `use(d_piece, bytes, first_unit)` (which enqueues on c->stream).
// Pieces hold whole units of `unit` bytes.  HAST_OK / HAST_ERR_IO (short file) / HAST_ERR_HIP.
static hast_status stream_file_region(hast_ctx *c, int fd, const char *path, uint64_t file_off, uint64_t n_units, size_t unit,
                                      const std::function<hipError_t(char *, size_t, uint64_t)> &use) {
    // the two pinned pieces (pinning costs ~0.7 ms per MB) and their device twins belong to the context: the second k-mer file
    // and a later --load-table reuse them
    constexpr size_t kPieceBytes = 16u << 20;
    const uint64_t per = std::max<uint64_t>(1, kPieceBytes / unit);             // units per piece
    char **h_buf = c->sf_h, **d_buf = c->sf_d;
    hipEvent_t *done = c->sf_done;
    hipError_t e = hipSuccess;
    bool short_read = false;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        if (!h_buf[i]) e = pinned_malloc((void **)&h_buf[i], kPieceBytes, hipHostMallocDefault);
        if (e == hipSuccess && !d_buf[i]) e = dev_malloc((void **)&d_buf[i], kPieceBytes + 16);
        if (e == hipSuccess && !done[i]) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) {
        const int nthreads = 8;
        WorkerPool pool(nthreads);
        size_t i = 0;
        for (uint64_t first = 0; first < n_units && e == hipSuccess; first += per, ++i) {
            const int b = (int)(i & 1);
            const size_t m = (size_t)std::min<uint64_t>(per, n_units - first), bytes = m * unit;
            if (i >= 2) e = hipEventSynchronize(done[b]);                       // the piece that used this buffer is on the device
            if (e != hipSuccess) break;
            const size_t share = ((bytes / nthreads) + 4095) & ~(size_t)4095;
            std::atomic<bool> bad{false};
            pool.run([&](int t) {
                const size_t from = std::min(bytes, share * (size_t)t), to = std::min(bytes, from + share);
                size_t got = 0;
                while (from + got < to) {
                    const ssize_t r = pread(fd, h_buf[b] + from + got, to - from - got, (off_t)(file_off + first * unit + from + got));
                    if (r <= 0) { bad = true; break; }
                    got += (size_t)r;
                }
            });
            if (bad) { short_read = true; break; }
            e = hipMemcpyAsync(d_buf[b], h_buf[b], bytes, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = use(d_buf[b], m, first);
            if (e == hipSuccess) e = hipEventRecord(done[b], c->stream);
        }
        const hipError_t e2 = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = e2;
    }
    if (e != hipSuccess) return fail(HAST_ERR_HIP, "%s: %s", path, hipGetErrorString(e));
    if (short_read) return fail(HAST_ERR_IO, "%s is shorter than its size says (truncated, or changed while it was read)", path);
    return HAST_OK;
}

// The same from a FILE, streamed: the reference reads a k-mer file line by line (classify.cpp:30-46; 4.4 GB per haplotype at
// the BASELINE sizes); here several threads pread the next piece into pinned memory while the previous one is copied to
// the device and inserted, so the load runs at the page cache's / the storage's rate instead of one thread's.
hast_status hast_table_insert_text_file(hast_ctx *c, int hap, const char *path, uint64_t *lines_out) {
    if (hast_status st = need_table(c, hap)) return st;
    if (lines_out) *lines_out = 0;
    if (!path) return fail(HAST_ERR_INVALID, "path is null");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(HAST_ERR_IO, "cannot read %s", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
        // not a regular file (a pipe): no pread -- the caller reads it into memory and uses hast_table_insert_text
        close(fd);
        return fail(HAST_ERR_IO, "%s is not a regular file", path);
    }
    const size_t nbytes = (size_t)sb.st_size, stride = (size_t)c->k + 1;
    const size_t n_lines = nbytes / stride, rem = nbytes - n_lines * stride;
    if (rem) {       // classify.cpp:41: a trailing piece without its own newline is dropped -- it must not hold one, though
        char tail[64];
        if (pread(fd, tail, rem, (off_t)(n_lines * stride)) != (ssize_t)rem) { close(fd); return fail(HAST_ERR_IO, "cannot read %s", path); }
        if (memchr(tail, '\n', rem)) { close(fd); return fail(HAST_ERR_FORMAT, "k-mer text: ragged last line"); }
    }
    table_changed(c);
    const hast_status st = stream_file_region(c, fd, path, 0, n_lines, stride, [&](char *d_piece, size_t m, uint64_t) {
        return launch_insert_text(c->d_slots, geom(c), d_piece, m, (uint32_t)hap, c->text_acgt_only, c->d_err, c->stream);
    });
    close(fd);
    if (st != HAST_OK) return st;
    if (hast_status s2 = check_err_word(c, c->stream)) return s2;
    if (lines_out) *lines_out = n_lines;
    return HAST_OK;
}

hast_status hast_table_erase(hast_ctx *c, const uint64_t *keys, size_t n, uint8_t *out_hit) {
    if (hast_status st = need_table(c, 0)) return st;
    if (n == 0) return HAST_OK;
    if (!keys) return fail(HAST_ERR_INVALID, "keys is null");
    size_t kb = n * sizeof(uint64_t);
    if (hast_status st = ensure_scratch(c, kb + n + 16)) return st;
    uint8_t *d_hit = (uint8_t *)c->d_scratch + kb;
    HIP_TRY(hipMemcpyAsync(c->d_scratch, keys, kb, hipMemcpyHostToDevice, c->stream));
    // one key at a time per launch order is not required: distinct keys touch distinct slots, and a
    // key listed twice reports its tags once (the second atomicAnd sees them already cleared) --
    // the same as the reference's find-then-erase loop (classify.cpp:318-337).
    HIP_TRY(launch_erase_keys(c->d_slots, geom(c), (const uint64_t *)c->d_scratch, n, d_hit, c->stream));
    table_changed(c);          // exact filter entries answer hits on their own: an erased key must leave the filter too
    std::vector<uint8_t> hit(n);
    HIP_TRY(hipMemcpyAsync(hit.data(), d_hit, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (out_hit) memcpy(out_hit, hit.data(), n);
    return HAST_OK;
}

hast_status hast_table_lookup(hast_ctx *c, const uint64_t *keys, size_t n, uint8_t *out_tags) {
    if (hast_status st = need_table(c, 0)) return st;
    if (n == 0) return HAST_OK;
    if (!keys || !out_tags) return fail(HAST_ERR_INVALID, "null argument");
    size_t kb = n * sizeof(uint64_t);
    if (hast_status st = ensure_scratch(c, kb + n + 16)) return st;
    uint8_t *d_tags = (uint8_t *)c->d_scratch + kb;
    HIP_TRY(hipMemcpyAsync(c->d_scratch, keys, kb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_lookup_keys(c->d_slots, geom(c), (const uint64_t *)c->d_scratch, n, d_tags, c->stream));
    HIP_TRY(hipMemcpyAsync(out_tags, d_tags, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HAST_OK;
}

hast_status hast_table_sizes(hast_ctx *c, uint64_t *n0, uint64_t *n1) {
    if (hast_status st = need_table(c, 0)) return st;
    unsigned long long h[2] = {0, 0};
    HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(h), c->stream));
    HIP_TRY(launch_count_tags(c->d_slots, geom(c), c->d_cnt, c->stream));
    HIP_TRY(hipMemcpyAsync(h, c->d_cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n0) *n0 = h[0];
    if (n1) *n1 = h[1];
    return HAST_OK;
}

// ---- binary key-set cache: file = 32-byte header + live slots (key<<2|tags), 8 B per distinct key
namespace {
struct CacheHeader {
    char magic[8];          // "HASTKEYS"
    uint32_t version, k;
    uint64_t n_slots;
    uint64_t reserved;
};
}  // namespace

hast_status hast_table_save(hast_ctx *c, const char *path) {
    if (hast_status st = need_table(c, 0)) return st;
    if (c->k == 32) return fail(HAST_ERR_INVALID, "the key-set cache stores tag bits next to the key: not available for K=32");
    if (!path) return fail(HAST_ERR_INVALID, "path is null");
    uint64_t n0 = 0, n1 = 0;
    if (hast_status st = hast_table_sizes(c, &n0, &n1)) return st;
    const size_t cap = (size_t)(n0 + n1) + 16;                    // distinct live keys <= n0 + n1
    uint64_t *d_out = nullptr;
    HIP_TRY(dev_malloc((void **)&d_out, cap * sizeof(uint64_t)));
    unsigned long long n = 0;
    hipError_t e = hipMemsetAsync(c->d_cnt, 0, sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = launch_export_slots(c->d_slots, (size_t)c->nbuckets * kSlotsPerBucket, d_out, cap, c->d_cnt, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&n, c->d_cnt, sizeof(n), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    std::vector<uint64_t> host;
    if (e == hipSuccess && n <= cap) {
        // sorted: a deterministic file for a given key set.  On the device (a library radix sort, as stage 00 uses for its
        // output order): one host thread needs ~30 s for the 400M slots of the BASELINE sets.
        uint64_t *d_sorted = nullptr;
        void *d_tmp = nullptr;
        size_t tmp_bytes = 0;
        if (n) e = kc_sort_keys(nullptr, &tmp_bytes, (unsigned long long *)d_out, nullptr, (size_t)n, 32, c->stream);
        if (e == hipSuccess && n) e = dev_malloc((void **)&d_sorted, (size_t)n * sizeof(uint64_t));
        if (e == hipSuccess && n) e = dev_malloc(&d_tmp, tmp_bytes ? tmp_bytes : 16);
        if (e == hipSuccess && n) e = kc_sort_keys(d_tmp, &tmp_bytes, (unsigned long long *)d_out, (unsigned long long *)d_sorted, (size_t)n, 32, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        host.resize((size_t)n);
        if (e == hipSuccess && n) e = hipMemcpy(host.data(), d_sorted, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost);
        (void)hipFree(d_sorted);
        (void)hipFree(d_tmp);
    }
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(HAST_ERR_HIP, "table export: %s", hipGetErrorString(e));
    if (n > cap) return fail(HAST_ERR_INVALID, "table export overflow");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(HAST_ERR_IO, "cannot write %s", path);
    CacheHeader h;
    memcpy(h.magic, "HASTKEYS", 8);
    h.version = 1;
    h.k = (uint32_t)c->k;
    h.n_slots = n;
    h.reserved = 0;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && (n == 0 || fwrite(host.data(), sizeof(uint64_t), (size_t)n, f) == (size_t)n);
    ok = (fclose(f) == 0) && ok;
    if (!ok) return fail(HAST_ERR_IO, "short write to %s", path);
    return HAST_OK;
}

hast_status hast_table_file_info(const char *path, int *k_out, uint64_t *n_keys_out) {
    if (!path) return fail(HAST_ERR_INVALID, "path is null");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(HAST_ERR_IO, "cannot read %s", path);
    CacheHeader h;
    const bool ok = fread(&h, sizeof(h), 1, f) == 1 && memcmp(h.magic, "HASTKEYS", 8) == 0 && h.version == 1;
    fclose(f);
    if (!ok) return fail(HAST_ERR_FORMAT, "%s is not a HASTKEYS v1 file", path);
    if (k_out) *k_out = (int)h.k;
    if (n_keys_out) *n_keys_out = h.n_slots;
    return HAST_OK;
}

hast_status hast_table_load(hast_ctx *c, const char *path, double load_factor) {
    if (hast_status st = use(c)) return st;
    if (!path) return fail(HAST_ERR_INVALID, "path is null");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(HAST_ERR_IO, "cannot read %s", path);
    CacheHeader h;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "HASTKEYS", 8) != 0 || h.version != 1) {
        fclose(f);
        return fail(HAST_ERR_FORMAT, "%s is not a HASTKEYS v1 file", path);
    }
    if (h.k == 32 || (int)h.k != c->k) {
        fclose(f);
        return fail(HAST_ERR_INVALID, "%s holds %u-mers, the context is K=%d", path, h.k, c->k);
    }
    hast_status st = hast_table_reserve(c, h.n_slots + 1, load_factor);
    struct stat sb;
    if (st == HAST_OK && fstat(fileno(f), &sb) == 0 && S_ISREG(sb.st_mode)) {
        // a regular file: streamed like the k-mer text (pread workers + pinned double buffer)
        st = stream_file_region(c, fileno(f), path, sizeof(CacheHeader), h.n_slots, sizeof(uint64_t), [&](char *d_piece, size_t m, uint64_t) {
            return launch_import_slots(c->d_slots, geom(c), (const uint64_t *)d_piece, m, c->d_err, c->stream);
        });
        if (st == HAST_ERR_IO) st = fail(HAST_ERR_IO, "%s is truncated", path);
    } else {
        const size_t per = kChunkBytes / sizeof(uint64_t);
        std::vector<uint64_t> buf(std::min<uint64_t>(per, h.n_slots ? h.n_slots : 1));
        if (st == HAST_OK) st = ensure_scratch(c, buf.size() * sizeof(uint64_t) + 16);
        for (uint64_t i = 0; st == HAST_OK && i < h.n_slots; i += per) {
            const size_t m = (size_t)std::min<uint64_t>(per, h.n_slots - i);
            if (fread(buf.data(), sizeof(uint64_t), m, f) != m) { st = fail(HAST_ERR_IO, "%s is truncated", path); break; }
            hipError_t e = hipMemcpyAsync(c->d_scratch, buf.data(), m * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = launch_import_slots(c->d_slots, geom(c), (const uint64_t *)c->d_scratch, m, c->d_err, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) st = fail(HAST_ERR_HIP, "table import: %s", hipGetErrorString(e));
        }
    }
    fclose(f);
    if (st != HAST_OK) return st;
    return check_err_word(c, c->stream);
}

// Replicate a finished table (and its filter) onto another context's device: one peer copy over xGMI instead of parsing
// and inserting the k-mer text once per GPU.
hast_status hast_table_clone(hast_ctx *dst, hast_ctx *src) {
    if (!dst || !src || dst == src) return fail(HAST_ERR_INVALID, "clone needs two different contexts");
    if (dst->k != src->k) return fail(HAST_ERR_INVALID, "clone: K differs (%d vs %d)", dst->k, src->k);
    if (hast_status st = need_table(src, 0)) return st;
    if (src->use_filter)
        if (hast_status st = ensure_filter(src, src->stream)) return st;
    HIP_TRY(hipStreamSynchronize(src->stream));
    if (hast_status st = use(dst)) return st;
    HIP_TRY(hipStreamSynchronize(dst->stream));
    if (dst->d_slots) HIP_TRY(hipFree(dst->d_slots));
    dst->d_slots = nullptr;
    dst->nbuckets = 0;
    dst->filter_valid = false;
    dst->m = src->m;
    const size_t bytes = table_slots(src) * sizeof(uint64_t);
    HIP_TRY(dev_malloc(&dst->d_slots, bytes));
    // ON dst's stream, and waited for below: hipMemcpyPeer between two contexts of ONE GPU is a device-to-device copy, which does not
    // wait for the host -- and dst's stream is a non-blocking one, so nothing would order the reads classified on it behind the copy
    // (seen with --devices 0,0,0 HAST_DEAL=files: a context's first blocks probed a table that was still arriving and lost hits; until
    // round 5 a hipHostFree elsewhere in the process happened to stop the device at the right moment)
    HIP_TRY(hipMemcpyPeerAsync(dst->d_slots, dst->device, src->d_slots, src->device, bytes, dst->stream));
    dst->nbuckets = src->nbuckets;
    if (src->filter_valid && dst->use_filter) {
        if (dst->filter_bytes != src->filter_bytes) {
            if (dst->d_filter) HIP_TRY(hipFree(dst->d_filter));
            dst->d_filter = nullptr;
            dst->filter_bytes = 0;
            if (dst->test_filter_oom || dev_malloc(&dst->d_filter, src->filter_bytes) != hipSuccess) {
                (void)hipGetLastError();
                dst->d_filter = nullptr;
                dst->use_filter = false;                   // no room: this device probes the table directly
                dst->filter_fallback_bytes = src->filter_bytes;
                HIP_TRY(hipStreamSynchronize(dst->stream));
                return HAST_OK;
            }
            dst->filter_bytes = src->filter_bytes;
        }
        HIP_TRY(hipMemcpyPeerAsync(dst->d_filter, dst->device, src->d_filter, src->device, src->filter_bytes, dst->stream));
        dst->fg = src->fg;
        dst->filter_m = src->filter_m;
        dst->filter_exact = src->filter_exact;
        dst->filter_t = src->filter_t;
        dst->filter_kp = src->filter_kp;
        dst->filter_valid = true;
    }
    HIP_TRY(hipStreamSynchronize(dst->stream));            // (src may change once this returns)
    return HAST_OK;
}

hast_status hast_table_info(const hast_ctx *c, uint64_t *n_buckets, uint64_t *bytes) {
    if (!c) return fail(HAST_ERR_INVALID, "null context");
    if (n_buckets) *n_buckets = c->nbuckets;
    if (bytes) *bytes = (uint64_t)table_slots(c) * sizeof(uint64_t);
    return HAST_OK;
}

// ---------------------------------------------------------------------------------------------
hast_status hast_counts_resize(hast_ctx *c, size_t n) {
    if (hast_status st = use(c)) return st;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->counts_owned && c->d_counts) HIP_TRY(hipFree(c->d_counts));
    c->d_counts = nullptr;
    c->n_barcodes = 0;
    c->counts_owned = false;
    size_t bytes = (n ? n : 1) * 4 * sizeof(unsigned long long);
    HIP_TRY(dev_malloc(reinterpret_cast<void **>(&c->d_counts), bytes));
    c->counts_owned = true;
    c->n_barcodes = n;
    HIP_TRY(hipMemsetAsync(c->d_counts, 0, bytes, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HAST_OK;
}

hast_status hast_counts_permute(hast_ctx *c, const uint32_t *perm, size_t n_perm, size_t n_new) {
    if (hast_status st = use(c)) return st;
    if (!c->d_counts || !c->counts_owned) return fail(HAST_ERR_INVALID, "hast_counts_permute: library-owned counters only (hast_counts_resize)");
    if (n_perm > c->n_barcodes || (n_perm && !perm)) return fail(HAST_ERR_INVALID, "hast_counts_permute: %zu ids > %zu counters", n_perm, c->n_barcodes);
    for (size_t i = 0; i < n_perm; i++)
        if (perm[i] >= n_new && perm[i] != 0xFFFFFFFFu) return fail(HAST_ERR_INVALID, "hast_counts_permute: perm[%zu] = %u is outside the %zu new records", i, perm[i], n_new);
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long *d_new = nullptr;
    uint32_t *d_perm = nullptr;
    const size_t bytes = (n_new ? n_new : 1) * 4 * sizeof(unsigned long long);
    HIP_TRY(dev_malloc(reinterpret_cast<void **>(&d_new), bytes));
    hipError_t e = dev_malloc(reinterpret_cast<void **>(&d_perm), (n_perm ? n_perm : 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(d_new, 0, bytes, c->stream);
    if (e == hipSuccess && n_perm) e = hipMemcpyAsync(d_perm, perm, n_perm * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_counts_permute(d_new, c->d_counts, d_perm, n_perm, c->n_barcodes, n_new, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (d_perm) (void)hipFree(d_perm);
    if (e != hipSuccess) {
        (void)hipFree(d_new);
        return fail(e == hipErrorOutOfMemory ? HAST_ERR_OOM : HAST_ERR_HIP, "hast_counts_permute: %s", hipGetErrorString(e));
    }
    HIP_TRY(hipFree(c->d_counts));
    c->d_counts = d_new;
    c->n_barcodes = n_new;
    return HAST_OK;
}

hast_status hast_counts_bind(hast_ctx *c, uint64_t *d_counts, size_t n) {
    if (hast_status st = use(c)) return st;
    if (!d_counts) return fail(HAST_ERR_INVALID, "d_counts is null");
    if ((uintptr_t)d_counts & 31) return fail(HAST_ERR_INVALID, "d_counts must be 32-byte aligned (one record = 4 x u64)");
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->counts_owned && c->d_counts) HIP_TRY(hipFree(c->d_counts));
    c->d_counts = reinterpret_cast<unsigned long long *>(d_counts);
    c->counts_owned = false;
    c->n_barcodes = n;
    return HAST_OK;
}

hast_status hast_counts_zero(hast_ctx *c, hast_stream s) {
    if (hast_status st = use(c)) return st;
    if (!c->d_counts) return fail(HAST_ERR_INVALID, "no counters: call hast_counts_resize/bind");
    HIP_TRY(hipMemsetAsync(c->d_counts, 0, c->n_barcodes * 4 * sizeof(unsigned long long), s ? (hipStream_t)s : c->stream));
    return HAST_OK;
}

static hast_status ensure_pack(hast_ctx *c, size_t n) {
    if (c->pack_words >= 3 * n) return HAST_OK;
    if (c->d_pack) HIP_TRY(hipFree(c->d_pack));
    c->d_pack = nullptr;
    c->pack_words = 0;
    HIP_TRY(dev_malloc((void **)&c->d_pack, std::max<size_t>(3 * n, 1) * sizeof(unsigned long long)));
    c->pack_words = 3 * n;
    return HAST_OK;
}

hast_status hast_counts_pack(hast_ctx *c, uint64_t *d_packed, size_t n, hast_stream s) {
    if (hast_status st = use(c)) return st;
    if (!c->d_counts || !d_packed) return fail(HAST_ERR_INVALID, "no counters");
    if (n > c->n_barcodes) return fail(HAST_ERR_INVALID, "n_barcodes %zu > %zu", n, c->n_barcodes);
    HIP_TRY(launch_counts_pack(c->d_counts, reinterpret_cast<unsigned long long *>(d_packed), n, s ? (hipStream_t)s : c->stream));
    return HAST_OK;
}
hast_status hast_counts_unpack(hast_ctx *c, const uint64_t *d_packed, size_t n, hast_stream s) {
    if (hast_status st = use(c)) return st;
    if (!c->d_counts || !d_packed) return fail(HAST_ERR_INVALID, "no counters");
    if (n > c->n_barcodes) return fail(HAST_ERR_INVALID, "n_barcodes %zu > %zu", n, c->n_barcodes);
    HIP_TRY(launch_counts_unpack(c->d_counts, reinterpret_cast<const unsigned long long *>(d_packed), n, s ? (hipStream_t)s : c->stream));
    return HAST_OK;
}

hast_status hast_counts_read_range(hast_ctx *c, size_t first, size_t n, uint64_t *c0, uint64_t *c1, uint64_t *neg) {
    if (hast_status st = use(c)) return st;
    if (!c->d_counts) return fail(HAST_ERR_INVALID, "no counters");
    if (first > c->n_barcodes || n > c->n_barcodes - first) return fail(HAST_ERR_INVALID, "records [%zu, %zu) of %zu counters", first, first + n, c->n_barcodes);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (hast_status st = check_classify_err(c, c->stream)) return st;
    if (!n) return HAST_OK;
    // the three live words of every record, as three arrays: 24 bytes per barcode over PCIe, each array straight into the caller's
    if (hast_status st = ensure_pack(c, n)) return st;
    HIP_TRY(launch_counts_pack(c->d_counts + 4 * first, c->d_pack, n, c->stream));
    uint64_t *dst[3] = {c0, c1, neg};
    for (int a = 0; a < 3; a++)
        if (dst[a]) HIP_TRY(hipMemcpyAsync(dst[a], c->d_pack + (size_t)a * n, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HAST_OK;
}
hast_status hast_counts_read(hast_ctx *c, uint64_t *c0, uint64_t *c1, uint64_t *neg, size_t n) { return hast_counts_read_range(c, 0, n, c0, c1, neg); }

// RCCL, resolved lazily so that single-GPU users never load it.
// Any device list: contexts that share a GPU (a logical split, e.g. classify --devices 0,0 on a single-GPU box, or 0,0,1,1) are summed
// by a kernel on that GPU into its first context (the leader), the leaders of DISTINCT GPUs then run the one RCCL all-reduce (a
// communicator has one rank per GPU), and every leader hands the totals back to the contexts behind it.
hast_status hast_counts_allreduce(hast_ctx *const *ctxs, int n) {
    if (!ctxs || n < 1) return fail(HAST_ERR_INVALID, "no contexts");
    for (int i = 0; i < n; i++)
        if (!ctxs[i] || !ctxs[i]->d_counts || ctxs[i]->n_barcodes != ctxs[0]->n_barcodes) return fail(HAST_ERR_INVALID, "contexts need equal-size counters");
    const size_t nbc = ctxs[0]->n_barcodes, nw = nbc * 4;
    // leaders: the first context of every device, in the order of the list; members[l] = the contexts behind leader l
    std::vector<hast_ctx *> lead;
    std::vector<std::vector<hast_ctx *>> members;
    for (int i = 0; i < n; i++) {
        size_t l = 0;
        while (l < lead.size() && lead[l]->device != ctxs[i]->device) ++l;
        if (l == lead.size()) {
            lead.push_back(ctxs[i]);
            members.emplace_back();
        } else {
            for (hast_ctx *m : members[l]) if (m == ctxs[i]) return fail(HAST_ERR_INVALID, "hast_counts_allreduce: a context is named twice");
            if (lead[l] == ctxs[i]) return fail(HAST_ERR_INVALID, "hast_counts_allreduce: a context is named twice");
            members[l].push_back(ctxs[i]);
        }
    }
    // 1. per device: the members' counters into the leader's
    for (size_t l = 0; l < lead.size(); l++) {
        if (members[l].empty()) continue;
        if (hast_status st = use(lead[l])) return st;
        HIP_TRY(hipStreamSynchronize(lead[l]->stream));
        for (hast_ctx *m : members[l]) HIP_TRY(hipStreamSynchronize(m->stream));
        for (hast_ctx *m : members[l]) HIP_TRY(launch_add_u64(lead[l]->d_counts, m->d_counts, nw, lead[l]->stream));
    }
    // 3. (after the exchange) per device: the totals back to the members
    auto hand_back = [&]() -> hast_status {
        for (size_t l = 0; l < lead.size(); l++) {
            if (members[l].empty()) continue;
            if (hast_status st = use(lead[l])) return st;
            for (hast_ctx *m : members[l])
                HIP_TRY(hipMemcpyAsync(m->d_counts, lead[l]->d_counts, nw * sizeof(unsigned long long), hipMemcpyDeviceToDevice, lead[l]->stream));
            HIP_TRY(hipStreamSynchronize(lead[l]->stream));
        }
        return HAST_OK;
    };
    // a single GPU needs no exchange; HAST_FORCE_RCCL=1 still runs the RCCL path (1-rank communicator), which
    // is how a 1-GPU box checks the library loading, symbols and enum values used for N > 1
    const int nl = (int)lead.size();
    if (nl == 1 && !getenv("HAST_FORCE_RCCL")) return hand_back();
    typedef void *comm_t;
    typedef int (*init_all_t)(comm_t *, int, const int *);
    typedef int (*allreduce_t)(const void *, void *, size_t, int, int, comm_t, hipStream_t);
    typedef int (*group_t)(void);
    typedef int (*destroy_t)(comm_t);
    static void *lib = nullptr;
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(HAST_ERR_RCCL, "cannot load librccl: %s", dlerror());
    auto init_all = (init_all_t)dlsym(lib, "ncclCommInitAll");
    auto allreduce = (allreduce_t)dlsym(lib, "ncclAllReduce");
    auto gstart = (group_t)dlsym(lib, "ncclGroupStart");
    auto gend = (group_t)dlsym(lib, "ncclGroupEnd");
    auto destroy = (destroy_t)dlsym(lib, "ncclCommDestroy");
    if (!init_all || !allreduce || !gstart || !gend || !destroy) return fail(HAST_ERR_RCCL, "librccl lacks symbols");
    std::vector<int> devs(nl);
    for (int i = 0; i < nl; i++) devs[i] = lead[(size_t)i]->device;
    // one communicator clique per device list, kept for the life of the process: ncclCommInitAll costs hundreds of ms on 8
    // GPUs, and the CLI merges at every counter regrowth (classify_main.cpp) as well as at the end
    static std::mutex comm_mu;
    static std::map<std::vector<int>, std::vector<comm_t>> comm_cache;
    std::lock_guard<std::mutex> lock(comm_mu);
    auto it = comm_cache.find(devs);
    if (it == comm_cache.end()) {
        std::vector<comm_t> fresh(nl);
        if (int rc = init_all(fresh.data(), nl, devs.data())) return fail(HAST_ERR_RCCL, "ncclCommInitAll failed (%d)", rc);
        it = comm_cache.emplace(devs, std::move(fresh)).first;
    }
    const std::vector<comm_t> &comms = it->second;
    const int kUint64 = 5, kSum = 0;   // ncclUint64, ncclSum (nccl.h: ncclInt64 = 4, ncclUint64 = 5)
    // 2. what crosses xGMI: the three live words of every record (c0[n] | c1[n] | neg[n], 24 bytes per barcode), packed and unpacked on
    // each GPU around the collective
    for (int i = 0; i < nl; i++) {
        hast_ctx *c = lead[(size_t)i];
        if (hast_status st = use(c)) return st;
        if (hast_status st = ensure_pack(c, nbc)) return st;
        HIP_TRY(launch_counts_pack(c->d_counts, c->d_pack, nbc, c->stream));
    }
    int rc = gstart();
    for (int i = 0; i < nl && !rc; i++) {
        hast_ctx *c = lead[(size_t)i];
        (void)hipSetDevice(c->device);
        rc = allreduce(c->d_pack, c->d_pack, nbc * 3, kUint64, kSum, comms[(size_t)i], c->stream);
    }
    int rc2 = gend();
    for (int i = 0; i < nl; i++) {
        hast_ctx *c = lead[(size_t)i];
        (void)hipSetDevice(c->device);
        if (!rc && !rc2 && launch_counts_unpack(c->d_counts, c->d_pack, nbc, c->stream) != hipSuccess) rc2 = -1;
        (void)hipStreamSynchronize(c->stream);
    }
    if (rc || rc2) {                   // a failed collective leaves the clique in an unknown state: drop it
        for (auto cm : comms) destroy(cm);
        comm_cache.erase(it);
    }
    if (rc || rc2) return fail(HAST_ERR_RCCL, "ncclAllReduce failed (%d/%d)", rc, rc2);
    return hand_back();
}

// ---------------------------------------------------------------------------------------------
// The fingerprint filter (hast_common.h): sized from the number of live keys, rebuilt from the table's live slots.
static hast_status ensure_filter(hast_ctx *c, hipStream_t hs) {
    if (c->filter_valid) return HAST_OK;
    unsigned long long h[2] = {0, 0};
    HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(h), hs));
    HIP_TRY(launch_count_tags(c->d_slots, geom(c), c->d_cnt, hs));
    HIP_TRY(hipMemcpyAsync(h, c->d_cnt, sizeof(h), hipMemcpyDeviceToHost, hs));
    HIP_TRY(hipStreamSynchronize(hs));
    FilterGeom fg = filter_geom_for(c->k, h[0] + h[1], c->filter_m, c->filter_t, c->filter_kp, c->k == 32 ? 0 : c->filter_exact);
    size_t bytes = (size_t)filter_nblocks(fg) * 128;
    if (bytes != c->filter_bytes) {
        if (c->d_filter) HIP_TRY(hipFree(c->d_filter));
        c->d_filter = nullptr;
        c->filter_bytes = 0;
        // (dev_malloc: what closed streams parked is freed and the allocation tried again before anything smaller or slower is taken)
        hipError_t got = c->test_filter_oom ? hipErrorOutOfMemory : dev_malloc(&c->d_filter, bytes);
        if (got != hipSuccess && !c->filter_m && fg.m == kFilterMaxM) {
            // no room for 4^15 blocks (137 GB, what 800M keys ask for): 4^14 with two choices per print, as up to round 2
            (void)hipGetLastError();
            fg = filter_geom_for(c->k, h[0] + h[1], kFilterMaxM - 1, c->filter_t, c->filter_kp, c->k == 32 ? 0 : c->filter_exact);
            bytes = (size_t)filter_nblocks(fg) * 128;
            got = dev_malloc(&c->d_filter, bytes);
        }
        if (got != hipSuccess) {
            // no room for the filter next to the table (34 GB at 14-mers): probe the table directly, as round 1 did -- the same
            // GPU path minus the front end, same results -- NOT silently: hast_filter_info says 0, hast_ctx_options (the CLI's
            // __stats_switches__ line and its WARN) name the fallback and the bytes that were missing
            (void)hipGetLastError();
            c->d_filter = nullptr;
            c->use_filter = false;
            c->filter_fallback_bytes = bytes;
            return HAST_OK;
        }
        c->filter_bytes = bytes;
    }
    HIP_TRY(hipMemsetAsync(c->d_filter, 0, bytes, hs));
    HIP_TRY(launch_filter_build(c->d_slots, geom(c), c->d_filter, fg, hs));
    c->fg = fg;
    c->filter_valid = true;
    return HAST_OK;
}

static constexpr uint32_t kSegWindows = 482;     // + K-1 <= 513 bases per segment row for K <= 32
static constexpr uint32_t kLongRead = 4096;      // longer reads go through the segmented path: a row's positions must fit the 12 bits tmer_order gives them

// Per-barcode bookkeeping of n_reads (vote0, vote1) pairs: one atomic per read, or -- large batches over many barcodes -- the pairs
// partitioned by barcode range and summed in LDS (hast_kernels.hip, "partitioned commit"; hast_ctx_set_option "commit" / HAST_COMMIT
// force either).  The partitioned path needs ~19 B of scratch per read: when HBM has no room for that the atomic path does the same
// sums (an optional speed-up must not turn a run that fits into an error), and the context does not ask again.
static hast_status commit_votes(hast_ctx *c, const uint32_t *d_votes, const uint32_t *d_ids, size_t n_reads, uint32_t max_votes,
                                bool atomic_only, hipStream_t hs) {
    const int mode = c->commit_mode;
    if (mode != 1 && !atomic_only && !c->part_oom && commit_partition_usable(n_reads, c->n_barcodes, max_votes, mode == 2)) {
        const size_t need = commit_partition_scratch_bytes(n_reads, c->n_barcodes, nullptr, nullptr);
        if (c->part_bytes < need) {
            HIP_TRY(hipStreamSynchronize(hs));
            if (c->d_part) HIP_TRY(hipFree(c->d_part));
            c->d_part = nullptr;
            c->part_bytes = 0;
            if (dev_malloc(&c->d_part, need + need / 8) == hipSuccess) c->part_bytes = need + need / 8;
            else {
                (void)hipGetLastError();
                c->d_part = nullptr;
                c->part_oom = true;
            }
        }
        if (c->d_part) {
            HIP_TRY(launch_commit_partitioned(d_votes, d_ids, c->d_counts, c->n_barcodes, n_reads, c->d_part, hs));
            return HAST_OK;
        }
    }
    HIP_TRY(launch_commit_votes(d_votes, d_ids, c->d_counts, nullptr, n_reads, hs));
    return HAST_OK;
}

static hast_status classify_rows(hast_ctx *c, const uint8_t *d_bases, size_t bases_bytes, const uint64_t *d_offsets,
                                 const uint32_t *d_lens, const uint32_t *d_seg_read, int strict, uint32_t read_len,
                                 const uint32_t *d_barcode_ids, uint32_t *d_votes, size_t n_reads, hast_stream s,
                                 const unsigned long long *d_n_rows = nullptr) {
    if (hast_status st = need_table(c, 0)) return st;
    if (n_reads == 0) return HAST_OK;
    if (!d_bases) return fail(HAST_ERR_INVALID, "d_bases is null");
    if (d_barcode_ids && !c->d_counts) return fail(HAST_ERR_INVALID, "barcode ids given but no counters bound");
    if (read_len == 0 || read_len > (1u << 24)) return fail(HAST_ERR_INVALID, "read_len %u out of range", read_len);
    // (tmer_order keeps a row position in 12 bits: longer rows are cut into segments by the callers)
    if (read_len > kLongRead && c->use_filter) return fail(HAST_ERR_INVALID, "internal: a row of %u bases reached the filter kernel", read_len);
    if (!d_offsets && (uint64_t)n_reads * read_len > bases_bytes)
        return fail(HAST_ERR_INVALID, "bases_bytes too small for %zu reads of %u", n_reads, read_len);
    ClassifyArgs a;
    a.bases = d_bases;
    a.bases_bytes = bases_bytes;
    a.offsets = d_offsets;
    a.lens = d_lens;
    a.seg_read = d_seg_read;
    a.strict = strict;
    // The kernel only writes per-read votes; the per-barcode bookkeeping runs as a kernel of its own afterwards (its random
    // read-modify-writes cost 4x as much when they are interleaved with the probes' reads).  Without a votes buffer from
    // the caller the votes go to library scratch.
    if (reinterpret_cast<uintptr_t>(d_votes) & 7) return fail(HAST_ERR_INVALID, "d_votes must be 8-byte aligned (a row is stored as one 64-bit word)");
    uint32_t *votes_buf = d_votes;
    if (d_barcode_ids && !votes_buf) {
        const size_t need = n_reads * 2 * sizeof(uint32_t);
        if (c->votes_bytes < need) {
            HIP_TRY(hipStreamSynchronize(s ? (hipStream_t)s : c->stream));
            if (c->d_votes_scratch) HIP_TRY(hipFree(c->d_votes_scratch));
            c->d_votes_scratch = nullptr;
            c->votes_bytes = 0;
            HIP_TRY(dev_malloc(reinterpret_cast<void **>(&c->d_votes_scratch), need + need / 8));
            c->votes_bytes = need + need / 8;
        }
        votes_buf = c->d_votes_scratch;
    }
    a.votes = votes_buf;
    a.slots = c->d_slots;
    a.n_reads = n_reads;
    a.n_rows_ptr = d_n_rows;                   // (n_reads is then the most rows there can be)
    a.nbuckets = c->nbuckets;
    a.read_len = read_len;
    a.k = c->k;
    a.wide = c->k == 32;
    a.m = c->m;
    a.max_pos = read_len >= (uint32_t)c->k ? read_len - c->k + 1 : 0;
    a.mh_stride = read_len >= (uint32_t)c->m ? read_len - c->m + 1 : 0;
    a.w64 = (read_len + 31) / 32;
    // Reads per tile: at most what fits the LDS budget of a workgroup and at most 64; among the
    // candidates take the one whose windows fill the waves' 64-window blocks best (a workgroup walks
    // 4 waves x 2 blocks per iteration: 150-bp reads => 31 reads = 4030 windows = 63 of 64 block slots).
    hipStream_t hs = s ? (hipStream_t)s : c->stream;
    if (c->use_filter)
        if (hast_status st = ensure_filter(c, hs)) return st;        // (may switch the filter off when HBM is short)
    const bool filt = c->use_filter;
    size_t per_read, pad;
    if (filt) {
        a.filter = c->d_filter;
        a.fg = c->fg;
        a.l1_stride = ((read_len >= (uint32_t)c->fg.t ? read_len - (uint32_t)c->fg.t + 1 : 0) + 1 + 3) & ~3u;
        per_read = (size_t)(a.w64 + 1) * 8 + 8 + 8 + 4 + 4 + (size_t)a.l1_stride * 4 + (strict ? (size_t)(2 * a.w64 + 1) * 4 : 0);
        pad = 16 + classify_f_queue_bytes() + 64 * 4 + 64;
    } else {
        // The m-mer hash array is padded by W entries so window-min reads past a read's last window stay in bounds.
        const uint32_t wlen = (uint32_t)(c->k - c->m + 1);
        a.filter = nullptr;
        a.fg = FilterGeom{};
        a.l1_stride = 0;
        per_read = (size_t)(a.w64 + 1) * 8 + 8 + 8 + 4 + 4 + (size_t)a.mh_stride * 4 + (strict ? (size_t)(2 * a.w64 + 1) * 4 : 0);
        pad = (size_t)wlen * 4 + 64 + 64 + 16;
    }
    const size_t lds_budget = c->tile_lds ? c->tile_lds : (filt ? (size_t)32000 : (size_t)19968);      // 5 / 8 workgroups per CU
    const uint32_t tr_max = (uint32_t)std::min<size_t>(64, std::max<size_t>(1, lds_budget > pad + per_read ? (lds_budget - pad) / per_read : 1));
    uint32_t tr = tr_max;
    if (a.max_pos > 0) {
        double best = -1;
        for (uint32_t t = tr_max; t >= 1 && t + 8 > tr_max; --t) {
            const uint64_t q = (uint64_t)t * a.max_pos, blocks = (q + 63) / 64, slots = filt ? (blocks + 3) / 4 * 4 : (blocks + 7) / 8 * 8;
            const double eff = (double)q / (double)(slots * 64);
            if (eff > best + 1e-9) { best = eff; tr = t; }
        }
    }
    a.tile_reads = tr;
    const size_t smem = per_read * tr + pad;
    if (smem > (160u << 10)) return fail(HAST_ERR_INVALID, "read_len %u needs %zu B of LDS", read_len, smem);
    // __umulhi(q, magic) == q / d for every q the kernel forms (exact while q*d < 2^32)
    auto magic = [](uint64_t qmax, uint32_t d) -> uint32_t {
        if (d <= 1 || qmax * d >= (1ull << 32)) return 0;
        return (uint32_t)((1ull << 32) / d) + 1;
    };
    a.div_magic = magic((uint64_t)tr * a.max_pos + 1024, a.max_pos);
    a.div_mh = magic((uint64_t)tr * a.mh_stride + 1024, a.mh_stride);
    a.div_hw = magic((uint64_t)tr * a.w64 * 2 + 1024, a.w64 * 2);
    a.div_l1g = magic((uint64_t)tr * (a.l1_stride / 4) + 1024, a.l1_stride / 4);
    const uint64_t n_tiles = (n_reads + tr - 1) / tr;
    const int grid = (int)std::min<uint64_t>(n_tiles, (uint64_t)c->n_cu * 8);
    a.tile_queue = c->d_cnt + 3;
    HIP_TRY(hipMemsetAsync(a.tile_queue, 0, sizeof(unsigned long long), hs));
    hipEvent_t *ev = c->t_slots ? &c->t_ev[3 * (c->t_next % c->t_slots)] : nullptr;
    if (ev) HIP_TRY(hipEventRecord(ev[0], hs));
    HIP_TRY(filt ? launch_classify_f(a, grid, smem, (c->kernel_geo ? 1 : 0) | (c->kernel_rl ? 2 : 0), hs) : launch_classify(a, grid, smem, hs));
    if (ev) HIP_TRY(hipEventRecord(ev[1], hs));
    if (d_barcode_ids) {
        // per-barcode bookkeeping: one atomic per read, or -- large batches over many barcodes -- the pairs partitioned by barcode
        // range and summed in LDS (hast_kernels.hip, "partitioned commit"; HAST_COMMIT=atomic / partition forces either)
        if (hast_status st = commit_votes(c, votes_buf, d_barcode_ids, n_reads, a.max_pos, d_seg_read != nullptr, hs)) return st;
    }
    if (ev) {
        HIP_TRY(hipEventRecord(ev[2], hs));
        c->t_next++;
        c->t_count = std::min(c->t_count + 1, c->t_slots);
    }
    return HAST_OK;
}

static hast_status classify_segmented(hast_ctx *c, const uint8_t *d_bases, size_t bases_bytes, const uint64_t *d_offsets, const uint32_t *d_lens,
                                      uint32_t fixed_len, size_t n_reads, int strict, const uint32_t *d_barcode_ids, uint32_t *d_votes_out,
                                      hipStream_t hs);

hast_status hast_ctx_set_filter(hast_ctx *c, int enable, int m, int t, int kp) {
    if (!c) return fail(HAST_ERR_INVALID, "null context");
    if (m < 0 || m > c->k || m > kFilterMaxM) return fail(HAST_ERR_INVALID, "filter m=%d out of [0,%d]", m, std::min(c->k, kFilterMaxM));
    if (t < 0 || (m && t > m)) return fail(HAST_ERR_INVALID, "filter t=%d out of [0,m]", t);
    if (kp < 0 || kp > c->k || (kp && m && (kp < m || kp - m >= 32))) return fail(HAST_ERR_INVALID, "filter kp=%d out of [m,K]", kp);
    c->use_filter = enable != 0;
    c->filter_fallback_bytes = 0;
    c->filter_exact = (enable == 2 || c->exact_env_off) ? 0 : -1;
    c->filter_m = m;
    c->filter_t = t;
    c->filter_kp = kp;
    c->filter_valid = false;
    return HAST_OK;
}

// Measurement switches of a live context (the environment is read once, in hast_ctx_create).
hast_status hast_ctx_set_option(hast_ctx *c, const char *name, long value) {
    if (!c || !name) return fail(HAST_ERR_INVALID, "null argument");
    if (!strcmp(name, "commit")) {
        if (value < 0 || value > 2) return fail(HAST_ERR_INVALID, "commit: 0 = by batch size, 1 = atomic, 2 = partitioned");
        c->commit_mode = (int)value;
    } else if (!strcmp(name, "kernel_geo")) c->kernel_geo = value != 0;
    else if (!strcmp(name, "kernel_rl")) c->kernel_rl = value != 0;
    else if (!strcmp(name, "tile_lds")) {
        if (value && (value < 4096 || value > 160 * 1024)) return fail(HAST_ERR_INVALID, "tile_lds %ld out of [4096, 163840]", value);
        c->tile_lds = (size_t)value;
    } else return fail(HAST_ERR_INVALID, "unknown option %s", name);
    return HAST_OK;
}
// the switches that differ from their defaults, as "name=value name=value" ("" when there are none): what --stats prints
hast_status hast_ctx_options(const hast_ctx *c, char *out, size_t cap) {
    if (!c || !out || !cap) return fail(HAST_ERR_INVALID, "null argument");
    std::string s;
    auto add = [&](const char *n, long v) { s += (s.empty() ? "" : " "); s += n; s += "=" + std::to_string(v); };
    if (!c->use_filter) add("filter", 0);
    if (c->filter_fallback_bytes) add("filter_fallback_table_only_no_room_for_bytes", (long)c->filter_fallback_bytes);
    if (c->filter_exact == 0) add("filter_exact", 0);
    if (c->filter_m) add("filter_m", c->filter_m);
    if (c->filter_t) add("filter_t", c->filter_t);
    if (c->filter_kp) add("filter_kp", c->filter_kp);
    if (c->commit_mode) add("commit", c->commit_mode);
    if (!c->kernel_geo) add("kernel_geo", 0);
    if (!c->kernel_rl) add("kernel_rl", 0);
    if (c->tile_lds) add("tile_lds", (long)c->tile_lds);
    if (c->m != default_minimizer_plain(c->k)) add("minimizer", c->m);
    snprintf(out, cap, "%s", s.c_str());
    return HAST_OK;
}

hast_status hast_filter_build(hast_ctx *c) {
    if (hast_status st = need_table(c, 0)) return st;
    if (!c->use_filter) return HAST_OK;
    if (hast_status st = ensure_filter(c, c->stream)) return st;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HAST_OK;
}

hast_status hast_filter_info(const hast_ctx *c, int *enabled, int *m, int *t, int *kp, uint64_t *bytes) {
    if (!c) return fail(HAST_ERR_INVALID, "null context");
    if (enabled) *enabled = !c->use_filter ? 0 : (c->filter_valid && c->fg.exact) ? 2 : 1;
    if (m) *m = c->filter_valid ? c->fg.m : 0;
    if (t) *t = c->filter_valid ? c->fg.t : 0;
    if (kp) *kp = c->filter_valid ? c->fg.kp : 0;
    if (bytes) *bytes = c->filter_valid ? c->filter_bytes : 0;
    return HAST_OK;
}

// Measurement: the rate at which this GPU serves uniformly random 128-B blocks of the context's own filter (the allocation
// k_classify_f probes), in requests per second -- the ceiling the probe kernel's request rate is priced against, measured in
// the same run on the same box over the same footprint (it differs by a few per cent from box to box and with the footprint:
// 47.5 G/s over 6 GB, 45.6 over 137 GB, profiles/round3_hbm_randread_footprint.jsonl).
hast_status hast_filter_request_ceiling(hast_ctx *c, double *requests_per_s) {
    if (hast_status st = need_table(c, 0)) return st;
    if (!requests_per_s) return fail(HAST_ERR_INVALID, "null argument");
    if (!c->use_filter) return fail(HAST_ERR_INVALID, "the filter is off");
    if (hast_status st = ensure_filter(c, c->stream)) return st;
    if (!c->use_filter || !c->d_filter) return fail(HAST_ERR_INVALID, "no filter (HBM too small for it)");
    const uint64_t nblocks = filter_nblocks(c->fg);
    const int grid = c->n_cu * 8;
    const uint32_t iters = 256;                                    // 2048 x 32 groups x 1024 blocks = 67M requests = 8.6 GB
    uint32_t *d_sink = reinterpret_cast<uint32_t *>(c->d_cnt + 2);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float best = 0;
    for (int rep = 0; rep < 3 && e == hipSuccess; ++rep) {         // the first launch warms the TLBs
        e = hipEventRecord(e0, c->stream);
        if (e == hipSuccess) e = launch_request_ceiling(c->d_filter, nblocks, iters, grid, d_sink, c->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && rep && (best == 0 || ms < best)) best = ms;
    }
    if (e0) (void)hipEventDestroy(e0);                             // (also on the error paths)
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) return fail(HAST_ERR_HIP, "request ceiling: %s", hipGetErrorString(e));
    *requests_per_s = (double)grid * 32.0 * 4.0 * (double)iters / ((double)best * 1e-3);
    return HAST_OK;
}

hast_status hast_classify_timing(hast_ctx *c, int n_slots) {
    if (hast_status st = use(c)) return st;
    for (hipEvent_t e : c->t_ev) (void)hipEventDestroy(e);
    c->t_ev.clear();
    c->t_slots = c->t_next = c->t_count = 0;
    if (n_slots <= 0) return HAST_OK;
    c->t_ev.resize(3 * (size_t)n_slots);
    for (hipEvent_t &e : c->t_ev) HIP_TRY(hipEventCreate(&e));
    c->t_slots = (size_t)n_slots;
    return HAST_OK;
}
hast_status hast_classify_times(hast_ctx *c, float *classify_ms, float *commit_ms, int max, int *n_out) {
    if (hast_status st = use(c)) return st;
    if (!n_out || (max > 0 && (!classify_ms || !commit_ms))) return fail(HAST_ERR_INVALID, "null argument");
    int n = 0;
    const size_t first = c->t_next - c->t_count;
    for (size_t i = 0; i < c->t_count && n < max; ++i, ++n) {
        hipEvent_t *ev = &c->t_ev[3 * ((first + i) % c->t_slots)];
        HIP_TRY(hipEventSynchronize(ev[2]));
        HIP_TRY(hipEventElapsedTime(&classify_ms[n], ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&commit_ms[n], ev[1], ev[2]));
    }
    *n_out = n;
    c->t_count = 0;
    return HAST_OK;
}

hast_status hast_classify_device(hast_ctx *c, const uint8_t *d_bases, size_t bases_bytes, const uint64_t *d_offsets,
                                 uint32_t read_len, const uint32_t *d_barcode_ids, uint32_t *d_votes, size_t n_reads,
                                 hast_stream s) {
    if (read_len > kLongRead && c && c->d_slots && n_reads) {
        // stage-01 semantics on long reads (with offsets or of one fixed length): whole-read N skip by a pre-pass, windows
        // through segments
        if (d_barcode_ids && !c->d_counts) return fail(HAST_ERR_INVALID, "barcode ids given but no counters bound");
        if (!d_bases) return fail(HAST_ERR_INVALID, "d_bases is null");
        if (!d_offsets && (uint64_t)n_reads * read_len > bases_bytes)
            return fail(HAST_ERR_INVALID, "bases_bytes too small for %zu reads of %u", n_reads, read_len);
        if (hast_status st = use(c)) return st;
        return classify_segmented(c, d_bases, bases_bytes, d_offsets, nullptr, read_len, n_reads, 0, d_barcode_ids, d_votes, s ? (hipStream_t)s : c->stream);
    }
    return classify_rows(c, d_bases, bases_bytes, d_offsets, nullptr, nullptr, 0, read_len, d_barcode_ids, d_votes, n_reads, s);
}

// process_reads' bookkeeping alone (classify.cpp:203-208) for votes the caller holds (e.g. per-read votes kept from an earlier
// classification, or a test's synthetic votes): barcode[0] += vote0, barcode[1] += vote1, or barcode[-1] += 1 when both are zero.
hast_status hast_counts_add_votes(hast_ctx *c, const uint32_t *d_votes, const uint32_t *d_barcode_ids, size_t n_reads, uint32_t max_votes,
                                  hast_stream s) {
    if (hast_status st = use(c)) return st;
    if (!c->d_counts) return fail(HAST_ERR_INVALID, "no counters: call hast_counts_resize/bind");
    if (n_reads == 0) return HAST_OK;
    if (!d_votes || !d_barcode_ids) return fail(HAST_ERR_INVALID, "null argument");
    if (reinterpret_cast<uintptr_t>(d_votes) & 7) return fail(HAST_ERR_INVALID, "d_votes must be 8-byte aligned");
    return commit_votes(c, d_votes, d_barcode_ids, n_reads, max_votes, false, s ? (hipStream_t)s : c->stream);
}

// Reads of any length: cut into segments on the device, classify the segments, add their votes per read.
//   strict = 1: stage-03 semantics (per-window validity), votes written to d_votes_out.
//   strict = 0: stage-01 semantics (a read holding 'N' is skipped as a whole: found by a pre-pass, such reads get no
//               windows), then per-read bookkeeping into the barcode counters and/or d_votes_out.
static hast_status classify_segmented(hast_ctx *c, const uint8_t *d_bases, size_t bases_bytes, const uint64_t *d_offsets, const uint32_t *d_lens,
                                      uint32_t fixed_len, size_t n_reads, int strict, const uint32_t *d_barcode_ids, uint32_t *d_votes_out,
                                      hipStream_t hs) {
    const size_t max_seg = n_reads + bases_bytes / kSegWindows + 1;
    const size_t votes_b = ((n_reads * 2 * sizeof(uint32_t) + 255) & ~(size_t)255), flags_b = ((n_reads + 255) & ~(size_t)255);
    const size_t need = max_seg * (sizeof(uint64_t) + 2 * sizeof(uint32_t)) + votes_b + flags_b + 256;
    if (c->seg_bytes < need) {
        HIP_TRY(hipStreamSynchronize(hs));
        if (c->d_seg) HIP_TRY(hipFree(c->d_seg));
        c->d_seg = nullptr;
        c->seg_bytes = 0;
        HIP_TRY(dev_malloc(&c->d_seg, need + need / 4));
        c->seg_bytes = need + need / 4;
    }
    uint32_t *acc = (uint32_t *)c->d_seg;                                  // per-read vote accumulator
    uint8_t *has_n = (uint8_t *)c->d_seg + votes_b;
    const size_t cap = (c->seg_bytes - votes_b - flags_b - 256) / (sizeof(uint64_t) + 2 * sizeof(uint32_t));
    uint64_t *seg_off = (uint64_t *)((uint8_t *)c->d_seg + votes_b + flags_b);
    uint32_t *seg_len = (uint32_t *)(seg_off + cap);
    uint32_t *seg_read = seg_len + cap;
    // strict callers own an output row per read; otherwise accumulate privately and commit afterwards
    uint32_t *target = strict ? d_votes_out : acc;
    unsigned long long *d_nseg = c->d_cnt + 4;                           // (its own word: ensure_filter counts tags in [0..1])
    HIP_TRY(hipMemsetAsync(d_nseg, 0, sizeof(unsigned long long), hs));
    HIP_TRY(hipMemsetAsync(target, 0, n_reads * 2 * sizeof(uint32_t), hs));
    if (!strict) HIP_TRY(launch_scan_n(d_bases, d_offsets, d_lens, fixed_len, n_reads, has_n, hs));
    HIP_TRY(launch_build_segments(d_offsets, d_lens, fixed_len, n_reads, c->k, kSegWindows, seg_off, seg_len, seg_read, d_nseg,
                                  strict ? nullptr : has_n, (uint64_t)max_seg, (uint64_t)bases_bytes, c->d_err + 1, hs));
    // the number of segments stays on the device (no host round trip between the two kernels): the classify kernel reads it
    // from d_nseg and is launched for the most segments there can be -- every read has at least one, every further one
    // covers kSegWindows more bases -- which the table was sized for above
    // (offsets that overlap or run backwards make the lengths add up to more: k_build_segments drops such rows and raises
    // c->d_err[1], which hast_stream_sync / hast_counts_read / the synchronous calls report)
    if (max_seg > cap) return fail(HAST_ERR_INVALID, "segment table too small (%zu > %zu)", max_seg, cap);
    if (hast_status st = classify_rows(c, d_bases, bases_bytes, seg_off, seg_len, seg_read, strict, kSegWindows + (uint32_t)c->k - 1,
                                       nullptr, target, max_seg, hs, d_nseg))
        return st;
    if (!strict) HIP_TRY(launch_commit_votes(acc, d_barcode_ids, c->d_counts, d_votes_out, n_reads, hs));
    return HAST_OK;
}

hast_status hast_classify_perread_device(hast_ctx *c, const uint8_t *d_bases, size_t bases_bytes, const uint64_t *d_offsets,
                                         size_t n_reads, uint32_t *d_votes, hast_stream s) {
    if (hast_status st = need_table(c, 0)) return st;
    if (n_reads == 0) return HAST_OK;
    if (!d_bases || !d_offsets || !d_votes) return fail(HAST_ERR_INVALID, "null argument");
    return classify_segmented(c, d_bases, bases_bytes, d_offsets, nullptr, 0, n_reads, 1, nullptr, d_votes, s ? (hipStream_t)s : c->stream);
}

hast_status hast_classify_perread(hast_ctx *c, const uint8_t *bases, const uint64_t *offsets, size_t n_reads,
                                  uint32_t *votes_out) {
    if (hast_status st = need_table(c, 0)) return st;
    if (n_reads == 0) return HAST_OK;
    if (!bases || !offsets || !votes_out) return fail(HAST_ERR_INVALID, "null argument");
    const size_t nbytes = offsets[n_reads] - offsets[0];
    const size_t ob = (n_reads + 1) * sizeof(uint64_t), vb = n_reads * 2 * sizeof(uint32_t);
    const size_t total = ((nbytes + 255) & ~(size_t)255) + ((ob + 255) & ~(size_t)255) + vb + 256;
    if (hast_status st = ensure_scratch(c, total)) return st;
    uint8_t *d_b = (uint8_t *)c->d_scratch;
    uint64_t *d_o = (uint64_t *)(d_b + ((nbytes + 255) & ~(size_t)255));
    uint32_t *d_v = (uint32_t *)((uint8_t *)d_o + ((ob + 255) & ~(size_t)255));
    std::vector<uint64_t> rel(n_reads + 1);
    for (size_t i = 0; i <= n_reads; i++) rel[i] = offsets[i] - offsets[0];
    if (nbytes) HIP_TRY(hipMemcpyAsync(d_b, bases + offsets[0], nbytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_o, rel.data(), ob, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));        // rel is a local
    if (hast_status st = hast_classify_perread_device(c, d_b, nbytes ? nbytes : 1, d_o, n_reads, d_v, c->stream)) return st;
    HIP_TRY(hipMemcpyAsync(votes_out, d_v, vb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return check_classify_err(c, c->stream);
}

static hast_status stage_reserve(hast_ctx *c, Staging &s, size_t nbytes, size_t nreads) {
    (void)c;
    if (s.cap_bases < nbytes) {
        if (s.h_bases) HIP_TRY(hipHostFree(s.h_bases));
        if (s.d_bases) HIP_TRY(hipFree(s.d_bases));
        s.h_bases = s.d_bases = nullptr;
        s.cap_bases = 0;
        size_t cap = nbytes + nbytes / 4 + 64;
        HIP_TRY(pinned_malloc((void **)&s.h_bases, cap, hipHostMallocDefault));
        HIP_TRY(dev_malloc((void **)&s.d_bases, cap));
        s.cap_bases = cap;
    }
    if (s.cap_reads < nreads) {
        if (s.h_off) HIP_TRY(hipHostFree(s.h_off));
        if (s.h_ids) HIP_TRY(hipHostFree(s.h_ids));
        if (s.d_off) HIP_TRY(hipFree(s.d_off));
        if (s.d_ids) HIP_TRY(hipFree(s.d_ids));
        s.h_off = s.d_off = nullptr;
        s.h_ids = s.d_ids = nullptr;
        s.cap_reads = 0;
        size_t cap = nreads + nreads / 4 + 16;
        HIP_TRY(pinned_malloc((void **)&s.h_off, (cap + 1) * sizeof(uint64_t), hipHostMallocDefault));
        HIP_TRY(pinned_malloc((void **)&s.h_ids, cap * sizeof(uint32_t), hipHostMallocDefault));
        HIP_TRY(dev_malloc((void **)&s.d_off, (cap + 1) * sizeof(uint64_t)));
        HIP_TRY(dev_malloc((void **)&s.d_ids, cap * sizeof(uint32_t)));
        s.cap_reads = cap;
    }
    return HAST_OK;
}

hast_status hast_batch_begin(hast_ctx *c, size_t bases_capacity, size_t reads_capacity, uint8_t **bases,
                             uint64_t **offsets, uint32_t **ids) {
    if (hast_status st = need_table(c, 0)) return st;
    if (!bases || !offsets || !ids) return fail(HAST_ERR_INVALID, "null argument");
    if (!c->d_counts) return fail(HAST_ERR_INVALID, "no counters bound");
    Staging &s = c->stage[c->batch_no & 1];
    if (s.in_flight) {
        HIP_TRY(hipEventSynchronize(s.done));
        s.in_flight = false;
    }
    if (hast_status st = stage_reserve(c, s, bases_capacity, reads_capacity)) return st;
    *bases = s.h_bases;
    *offsets = s.h_off;
    *ids = s.h_ids;
    return HAST_OK;
}

hast_status hast_batch_submit(hast_ctx *c, size_t n_reads, uint32_t max_read_len) {
    if (hast_status st = need_table(c, 0)) return st;
    Staging &s = c->stage[c->batch_no & 1];
    if (n_reads == 0) return HAST_OK;
    if (n_reads > s.cap_reads) return fail(HAST_ERR_INVALID, "batch of %zu reads exceeds the staged capacity", n_reads);
    const size_t nbytes = s.h_off[n_reads];
    if (nbytes > s.cap_bases || s.h_off[0] != 0) return fail(HAST_ERR_INVALID, "bad staged offsets");
    c->batch_no++;
    if (nbytes) HIP_TRY(hipMemcpyAsync(s.d_bases, s.h_bases, nbytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(s.d_off, s.h_off, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(s.d_ids, s.h_ids, n_reads * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    hast_status st = hast_classify_device(c, s.d_bases, nbytes ? nbytes : 1, s.d_off, max_read_len ? max_read_len : 1, s.d_ids,
                                          nullptr, n_reads, c->stream);
    HIP_TRY(hipEventRecord(s.done, c->stream));
    s.in_flight = true;
    return st;
}

hast_status hast_classify_batch(hast_ctx *c, const uint8_t *bases, const uint64_t *offsets, const uint32_t *ids,
                                size_t n_reads, uint32_t max_read_len) {
    if (n_reads == 0) return c ? HAST_OK : fail(HAST_ERR_INVALID, "null context");
    if (!bases || !offsets || !ids) return fail(HAST_ERR_INVALID, "null argument");
    const size_t nbytes = offsets[n_reads] - offsets[0];
    uint8_t *hb;
    uint64_t *ho;
    uint32_t *hi;
    if (hast_status st = hast_batch_begin(c, nbytes, n_reads, &hb, &ho, &hi)) return st;
    memcpy(hb, bases + offsets[0], nbytes);
    const uint64_t o0 = offsets[0];
    for (size_t i = 0; i <= n_reads; i++) ho[i] = offsets[i] - o0;
    memcpy(hi, ids, n_reads * sizeof(uint32_t));
    return hast_batch_submit(c, n_reads, max_read_len);
}

}  // extern "C"

// ---- hooks for fq_api.cpp (hast_internal.h): reads given as starts + lengths inside a raw FASTQ block -----------------
namespace hast {
hast_status classify_framed(hast_ctx *c, const uint8_t *d_buf, size_t buf_bytes, const uint64_t *d_off, const uint32_t *d_len, uint32_t max_len,
                            uint32_t *d_votes, size_t n_reads, hipStream_t hs) {
    if (hast_status st = need_table(c, 0)) return st;
    if (n_reads == 0) return HAST_OK;
    if (max_len > kLongRead) return classify_segmented(c, d_buf, buf_bytes, d_off, d_len, 0, n_reads, 0, nullptr, d_votes, hs);
    return classify_rows(c, d_buf, buf_bytes, d_off, d_len, nullptr, 0, max_len ? max_len : 1, nullptr, d_votes, n_reads, hs);
}
hast_status commit_framed(hast_ctx *c, const uint32_t *d_votes, const uint32_t *d_ids, size_t n_reads, hipStream_t hs) {
    if (hast_status st = use(c)) return st;
    if (!c->d_counts) return fail(HAST_ERR_INVALID, "no counters bound");
    HIP_TRY(launch_commit_votes(d_votes, d_ids, c->d_counts, nullptr, n_reads, hs));
    return HAST_OK;
}
hipStream_t ctx_stream_of(hast_ctx *c) { return c->stream; }
size_t ctx_n_barcodes(const hast_ctx *c) { return c->n_barcodes; }
}  // namespace hast

extern "C" {

// ---------------------------------------------------------------------------------------------
// host-side pieces
void hast_parse_barcode(const char *head, size_t len, size_t *start, size_t *n) {
    // classify.cpp:112-119: last '#', last '/'; substr(s+1, e-s-1) where a negative count means "to the end"
    ptrdiff_t s = -1, e = -1;
    for (size_t i = 0; i < len; i++) {
        if (head[i] == '#') s = (ptrdiff_t)i;
        else if (head[i] == '/') e = (ptrdiff_t)i;
    }
    size_t pos = (size_t)(s + 1);
    size_t avail = len - pos;
    ptrdiff_t cnt = e - s - 1;
    *start = pos;
    *n = (cnt < 0 || (size_t)cnt > avail) ? avail : (size_t)cnt;
}

int hast_get_hap(const char *bc, size_t blen, uint64_t c0, uint64_t c1, uint64_t n0, uint64_t n1, double w0, double w1) {
    // classify.cpp:66-86
    if ((blen == 5 && !memcmp(bc, "0_0_0", 5)) || (blen == 3 && !memcmp(bc, "0_0", 3)) || (blen == 1 && bc[0] == '0'))
        return -1;
    if (c0 > 0 && c1 > 0) {
        // the reference holds counts in `int` (classify.cpp:51) and converts int -> double: the same value up to INT_MAX; past it
        // the reference's counter has overflowed (undefined there), here the call goes by the true count
        double df0 = double(c0) / double(n0);
        double df1 = double(c1) / double(n1);
        df0 *= w0;
        df1 *= w1;
        if (df0 > df1) return 0;
        if (df1 > df0) return 1;
        return -1;
    }
    if (c0 > 0) return 0;
    if (c1 > 0) return 1;
    return -1;
}

uint64_t hast_canon_kmer(const char *s, int k) { return kmer_canon(kmer_pack(s, k), k); }

size_t hast_chop_read(const char *seq, size_t len, int k, uint64_t *out) {
    if (k < 1 || k > 32 || len < (size_t)k) return 0;
    const uint64_t mask = kmer_mask(k);
    uint64_t w = kmer_pack(seq, k);
    size_t n = 0;
    out[n++] = kmer_canon(w, k);
    for (size_t i = (size_t)k; i < len; i++) {
        w = ((w << 2) & mask) | base_code((uint8_t)seq[i]);
        out[n++] = kmer_canon(w, k);
    }
    return n;
}

void hast_kmer_to_str(uint64_t kmer, int k, char *out) {
    for (int i = 0; i < k; i++) {
        out[k - 1 - i] = "ACTG"[kmer & 3];   // kmer.h:12
        kmer >>= 2;
    }
    out[k] = 0;
}

// ---------------------------------------------------------------------------------------------
hast_status hast_synth_keys_host(const hast_synth_params *p, int hap, uint64_t first, size_t n, uint64_t *out) {
    if (hast_status st = check_synth(p)) return st;
    SynthParams sp = resolve(p);
    for (size_t i = 0; i < n; i++) out[i] = synth_key(sp, hap, first + i);
    return HAST_OK;
}

hast_status hast_synth_reads_host(const hast_synth_params *p, uint64_t first, size_t n, uint8_t *bases, uint32_t *ids) {
    if (hast_status st = check_synth(p)) return st;
    SynthParams sp = resolve(p);
    if (sp.read_len == 0) return fail(HAST_ERR_INVALID, "read_len is 0");
    for (size_t i = 0; i < n; i++) {
        uint32_t bc;
        synth_read(sp, first + i, bases + i * (size_t)sp.read_len, &bc);
        if (ids) ids[i] = bc;
    }
    return HAST_OK;
}

hast_status hast_synth_keys_device(hast_ctx *c, const hast_synth_params *p, int hap, uint64_t first, size_t n,
                                   uint64_t *d_out, hast_stream s) {
    if (hast_status st = use(c)) return st;
    if (hast_status st = check_synth(p)) return st;
    HIP_TRY(launch_synth_keys(resolve(p), hap, first, n, d_out, s ? (hipStream_t)s : c->stream));
    return HAST_OK;
}

hast_status hast_synth_reads_device(hast_ctx *c, const hast_synth_params *p, uint64_t first, size_t n, uint8_t *d_bases,
                                    uint32_t *d_ids, hast_stream s) {
    if (hast_status st = use(c)) return st;
    if (hast_status st = check_synth(p)) return st;
    if (p->read_len == 0) return fail(HAST_ERR_INVALID, "read_len is 0");
    HIP_TRY(launch_synth_reads(resolve(p), first, n, d_bases, d_ids, s ? (hipStream_t)s : c->stream));
    return HAST_OK;
}

hast_status hast_synth_table_build(hast_ctx *c, const hast_synth_params *p) {
    if (hast_status st = need_table(c, 0)) return st;
    if (hast_status st = check_synth(p)) return st;
    if ((int)p->k != c->k) return fail(HAST_ERR_INVALID, "synth k=%u != context K=%d", p->k, c->k);
    table_changed(c);
    SynthParams sp = resolve(p);
    const size_t per = kChunkBytes / sizeof(uint64_t);
    if (hast_status st = ensure_scratch(c, std::min<uint64_t>(sp.n_keys_per_hap, per) * sizeof(uint64_t) + 16)) return st;
    for (int hap = 0; hap < 2; hap++)
        for (uint64_t i = 0; i < sp.n_keys_per_hap; i += per) {
            size_t m = (size_t)std::min<uint64_t>(per, sp.n_keys_per_hap - i);
            HIP_TRY(launch_synth_keys(sp, hap, i, m, (uint64_t *)c->d_scratch, c->stream));
            HIP_TRY(launch_insert_keys(c->d_slots, geom(c), (const uint64_t *)c->d_scratch, m, (uint32_t)hap, c->d_err, c->stream));
        }
    return check_err_word(c, c->stream);
}

}  // extern "C"

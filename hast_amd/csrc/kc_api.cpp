// kc_api.cpp -- the stage-00 part of the C ABI (include/hast.h, hast_kc_*): k-mer count table of both parents in HBM,
// histograms, parent-unique selections.  Host C++ over the HIP runtime; device work is in kc_kernels.hip.
// No CPU path: a context needs a GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "hast_internal.h"
#include "kc_device.h"

using namespace hast;

#define KC_TRY(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return set_error(e_ == hipErrorOutOfMemory ? HAST_ERR_OOM : HAST_ERR_HIP, "%s: %s", #expr, \
                             hipGetErrorString(e_));                                                  \
    } while (0)

namespace {
constexpr size_t kStageBytes = 64u << 20;      // one pinned/device staging buffer of hast_kc_count
struct KcStage {
    uint8_t *h = nullptr, *d = nullptr;
    hipEvent_t done = nullptr;
    bool busy = false;
};
// d_small layout (unsigned long long words)
enum { kQueue = 0, kTotal = 1 /* ,2 */, kCursor = 3, kStats = 4 /* ,5,6 */, kRecCursor = 7, kSpillN = 8, kSmallWords = 10 };
}  // namespace

struct hast_kc {
    int device = 0, k = 0, m = 0, n_cu = 256;
    hipStream_t stream = nullptr;
    unsigned long long *d_table = nullptr;
    uint32_t nbuckets = 0;
    uint32_t slice = 0, n_slices = 1;
    uint64_t expected_windows = 0;                 // what the caller expects to count (0 = unknown): bounds the record buffers
    uint32_t tile_bases = 4096;
    unsigned long long *d_small = nullptr;
    uint32_t *d_err = nullptr;
    unsigned long long *d_histo = nullptr;
    KcStage stage[2];
    unsigned turn = 0;
    // early word on a full table: the error word is copied to pinned memory every few chunks and looked at without waiting
    uint32_t *h_err = nullptr;
    hipEvent_t err_ev = nullptr;
    bool err_pending = false;
    // partitioned counting (kc_kernels.hip, "partitioned counting"): windows are written out as records and applied to the table in
    // flushes -- at hast_kc_sync, before the table is read, and when the record buffer is nearly full
    bool part_on = false;
    uint32_t fine_shift = 9, n_fine = 0, n_l1 = 0, f2 = 0;
    bool fresh = false;                        // the table holds nothing and has NOT been cleared: the first flush writes every slice (KcFlushArgs.fresh)
    uint32_t l1_split = 8;                     // a level-1 bin's records lie in l1_split regions (KcFlushArgs)
    int small_flush = 0;                       // HAST_KC_FLUSH=sweep|atomic pins how a flush is applied (1 | 2); default: by size
    unsigned long long *d_rec = nullptr, *d_l1 = nullptr, *d_spill = nullptr;
    uint32_t *d_fills = nullptr;               // [n_l1 fill | n_l1 valid] (kKcL1FillWords apart) [n_fine fill | n_fine valid]
    uint64_t rec_cap = 0, a_cap = 0, b_cap = 0, spill_cap = 0;
    uint64_t est_records = 0;                  // upper bound of the records written since the last flush
    // ... kept tight by the device's own cursor: a copy of it travels back behind a count launch and replaces the bound of everything
    // launched up to there (a bound of one record per two windows is twice what real reads write: the bench flushed twice a step)
    unsigned long long *h_cursor = nullptr;    // pinned
    hipEvent_t cur_ev = nullptr;
    bool cur_pending = false;
    uint64_t est_after = 0;                    // bound of the launches behind the copy in flight
    uint64_t n_flushes = 0, n_flushed_records = 0, n_spilled = 0;
    std::vector<uint64_t> sel[2];              // print keys selected so far (host side, unsorted)
    unsigned long long *d_sorted[2] = {nullptr, nullptr};
    size_t n_sorted[2] = {0, 0};
};

namespace {
hast_status use(hast_kc *c) {
    if (!c) return set_error(HAST_ERR_INVALID, "null k-mer count context");
    KC_TRY(hipSetDevice(c->device));
    return HAST_OK;
}
hast_status need_table(hast_kc *c) {
    if (hast_status st = use(c)) return st;
    if (!c->d_table) return set_error(HAST_ERR_INVALID, "the count table has been released");
    return HAST_OK;
}
hast_status check_parent(int parent) {
    return (parent == 0 || parent == 1) ? HAST_OK : set_error(HAST_ERR_INVALID, "parent %d is not 0 (paternal) or 1 (maternal)", parent);
}
KcSynth resolve(const hast_kc_synth *p) {
    KcSynth g;
    g.seed = p->seed ? p->seed : 0x4841535400ull;
    g.genome_len = p->genome_len;
    g.read_len = p->read_len;
    g.snp_per_1024 = p->snp_per_1024;
    g.err_per_4096 = p->err_per_4096;
    g.n_per_4096 = p->n_per_4096;
    return g;
}
hast_status check_synth(const hast_kc_synth *p) {
    if (!p) return set_error(HAST_ERR_INVALID, "null synth params");
    if (p->read_len < 1 || p->genome_len < p->read_len) return set_error(HAST_ERR_INVALID, "synth: genome shorter than a read");
    return HAST_OK;
}
}  // namespace

// Partitioned counting (kc_kernels.hip) is worth its buffers when the table is far larger than the caches (>= 2^20 buckets = 128 MB)
// and a record has room for a window (K <= 27 with the default minimizer length); HAST_KC_COUNT=atomic|partition in the environment
// (read here, once) forces either.  Round 4, bench.py --workload s00 (10.4 G windows, 60-GB table): 0.151 s a step against 0.271 s
// of the direct kernel.  The buffers take what is left of the device memory next to the table: ~22.5 B per record of capacity.
// HAST_KC_FLUSH=sweep|atomic pins how a flush is applied (default: by the number of records, launch_kc_flush).
static void part_setup(hast_kc *c) {
    const char *e = getenv("HAST_KC_COUNT");
    const bool forced = e && !strcmp(e, "partition"), off = e && !strcmp(e, "atomic");
    if (const char *fl = getenv("HAST_KC_FLUSH")) c->small_flush = !strcmp(fl, "sweep") ? 1 : !strcmp(fl, "atomic") ? 2 : 0;
    if (off || kc_run_max(c->k, c->m) == 0 || (!forced && c->nbuckets < (1u << 20))) return;
    c->fine_shift = ((uint64_t)c->nbuckets >> 9) > (1u << 20) ? 10 : 9;
    const uint64_t n_fine = ((uint64_t)c->nbuckets + (1u << c->fine_shift) - 1) >> c->fine_shift;
    if (n_fine > (1u << 20)) return;                                   // a table of more than 2^30 buckets (128 GB): two levels of 1024 do not reach
    c->n_fine = (uint32_t)n_fine;
    c->f2 = 1;                                                         // fine bins per level-1 bin: a power of two (a shift in the kernels)
    while ((n_fine + c->f2 - 1) / c->f2 > 1024) c->f2 <<= 1;
    c->n_l1 = (uint32_t)((n_fine + c->f2 - 1) / c->f2);
    if (const char *sp = getenv("HAST_KC_L1_SPLIT")) c->l1_split = (uint32_t)std::min(32l, std::max(1l, atol(sp)));
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
    const uint64_t fixed = 8ull * (64ull * c->n_fine + 4096ull * c->n_l1 * c->l1_split + (1u << 20)) + 8ull * ((uint64_t)c->n_l1 * c->l1_split * kKcL1FillWords + c->n_fine) + (64u << 20);
    const uint64_t budget = (uint64_t)((double)free_b * 0.8);
    if (budget <= fixed + (16u << 20)) return;
    uint64_t R = (uint64_t)((double)(budget - fixed) / 22.6);
    R = std::min<uint64_t>(R, 6ull << 30);
    // ... and no more records than the table has slots: a caller sizes the table for its input, and device memory that another
    // process has just given back is slow to get (seconds per 100 GB: the stage-00 program run back to back)
    R = std::min<uint64_t>(R, std::max<uint64_t>((uint64_t)c->nbuckets * kKcSlots, 64ull << 20));
    // ... nor more than the input can write (a record holds 3.3 windows on average, never fewer than one per two here: the bound the
    // flush logic works with): a 20-Mbp trio then takes 16 GB of record buffers, not the 60 GB that happened to be free
    if (c->expected_windows) R = std::min<uint64_t>(R, std::max<uint64_t>(c->expected_windows / 2 + (1u << 20), 16ull << 20));
    if (const char *m = getenv("HAST_KC_RECORD_MB")) R = std::min<uint64_t>(R, ((uint64_t)atol(m) << 20) / 8);     // (the record buffer itself)
    if (R < (forced ? 4096u : (16u << 20))) return;
    c->rec_cap = R;
    c->a_cap = R + R / 2 + 64ull * c->n_fine + 1024;
    c->b_cap = R + R / 4 + 4096ull * c->n_l1 * c->l1_split + 1024;
    c->spill_cap = R / 16 + (1u << 20);
    hipError_t h = dev_malloc(reinterpret_cast<void **>(&c->d_rec), c->a_cap * 8);
    if (h == hipSuccess) h = dev_malloc(reinterpret_cast<void **>(&c->d_l1), c->b_cap * 8);
    if (h == hipSuccess) h = dev_malloc(reinterpret_cast<void **>(&c->d_spill), c->spill_cap * 8);
    if (h == hipSuccess) h = dev_malloc(reinterpret_cast<void **>(&c->d_fills), 8ull * ((uint64_t)c->n_l1 * c->l1_split * kKcL1FillWords + c->n_fine));
    if (h != hipSuccess) {                                             // no room: count with atomics, as before
        (void)hipGetLastError();
        for (void *p : {(void *)c->d_rec, (void *)c->d_l1, (void *)c->d_spill, (void *)c->d_fills})
            if (p) (void)hipFree(p);
        c->d_rec = c->d_l1 = c->d_spill = nullptr;
        c->d_fills = nullptr;
        return;
    }
    c->part_on = true;
}

// a table that is empty by declaration only (c->fresh) is cleared for real
static hast_status table_real(hast_kc *c) {
    if (!c->fresh) return HAST_OK;
    KC_TRY(launch_kc_clear(c->d_table, c->nbuckets, c->stream));
    c->fresh = false;
    return HAST_OK;
}
// an empty table: the partitioned path's first flush writes every slice anyway, so it is only DECLARED empty (the clear is 9 ms of a
// 110-ms step on the bench's 64-GB table; HAST_KC_FRESH=0: always cleared)
static hast_status table_empty(hast_kc *c) {
    const char *e = getenv("HAST_KC_FRESH");
    const bool lazy = !(e && !strcmp(e, "0"));
    c->fresh = c->part_on && lazy;
    if (c->fresh) return HAST_OK;
    KC_TRY(launch_kc_clear(c->d_table, c->nbuckets, c->stream));
    return HAST_OK;
}

// apply what has been written out since the last flush (no-op when there is nothing)
static hast_status part_flush(hast_kc *c) {
    if (!c->part_on || c->est_records == 0) return table_real(c);
    unsigned long long cur = 0;
    KC_TRY(hipMemcpyAsync(&cur, c->d_small + kRecCursor, sizeof(cur), hipMemcpyDeviceToHost, c->stream));
    KC_TRY(hipStreamSynchronize(c->stream));
    const uint64_t n = std::min<uint64_t>(cur, c->rec_cap);
    KcFlushArgs a;
    a.table = c->d_table;
    a.nbuckets = c->nbuckets;
    a.k = c->k;
    a.m = c->m;
    a.fine_shift = c->fine_shift;
    a.n_fine = c->n_fine;
    a.n_l1 = c->n_l1;
    a.f2 = c->f2;
    a.records = c->d_rec;
    a.n_records = n;
    a.rec_cursor = c->d_small + kRecCursor;
    a.small_flush = c->small_flush;
    a.l1_recs = c->d_l1;
    a.l1_split = c->l1_split;
    const uint64_t n_reg = (uint64_t)c->n_l1 * c->l1_split;
    a.l1_cap = (uint32_t)std::min<uint64_t>(0x7FFFFFFFu, std::min<uint64_t>(c->b_cap / n_reg, n / n_reg + n / n_reg / 4 + 4096));
    a.fine_cap = (uint32_t)std::min<uint64_t>(0x7FFFFFFFu, std::min<uint64_t>(c->a_cap / c->n_fine, n / c->n_fine + n / c->n_fine / 2 + 64));
    a.l1_fill = c->d_fills;
    a.l1_valid = c->d_fills + (size_t)n_reg * kKcL1FillWords;
    a.fine_fill = c->d_fills + 2 * (size_t)n_reg * kKcL1FillWords;
    a.fine_valid = a.fine_fill + c->n_fine;
    a.spill = c->d_spill;
    a.spill_cap = c->spill_cap;
    a.spill_n = c->d_small + kSpillN;
    a.err = c->d_err;
    a.fresh = 0;
    if (c->fresh) {
        if (n == 0 || kc_flush_is_small(a)) {          // (no sweep: the records are counted where they lie, into a real table)
            if (hast_status st = table_real(c)) return st;
        } else a.fresh = 1;
    }
    KC_TRY(launch_kc_flush(a, c->stream));
    c->fresh = false;
    unsigned long long sp = 0;
    KC_TRY(hipMemcpyAsync(&sp, c->d_small + kSpillN, sizeof(sp), hipMemcpyDeviceToHost, c->stream));
    KC_TRY(hipMemsetAsync(c->d_small + kRecCursor, 0, 2 * sizeof(unsigned long long), c->stream));      // cursor and spill count
    KC_TRY(hipStreamSynchronize(c->stream));
    c->est_records = c->est_after = 0;
    c->cur_pending = false;                        // (the stream has been waited for: a copy of the old cursor is history)
    c->n_flushes++;
    c->n_flushed_records += n;
    c->n_spilled += sp;
    return HAST_OK;
}

hast_status hast_kc_create(int device, int k, size_t table_bytes, hast_kc **out) { return hast_kc_create_ex(device, k, table_bytes, 0, out); }
hast_status hast_kc_create_ex(int device, int k, size_t table_bytes, uint64_t expected_windows, hast_kc **out) {
    if (!out) return set_error(HAST_ERR_INVALID, "out is null");
    *out = nullptr;
    if (k < 1 || k > 32) return set_error(HAST_ERR_INVALID, "K=%d out of [1,32]", k);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_error(HAST_ERR_NO_DEVICE, "no HIP device (%s); libhast has no CPU path", hipGetErrorString(e));
    if (device < 0 || device >= n) return set_error(HAST_ERR_NO_DEVICE, "device %d not in [0,%d)", device, n);
    KC_TRY(hipSetDevice(device));
    hast_kc *c = new (std::nothrow) hast_kc();
    if (!c) return set_error(HAST_ERR_OOM, "host allocation failed");
    c->device = device;
    c->k = k;
    c->m = default_minimizer_for(k);
    c->expected_windows = expected_windows;
    if (const char *t = getenv("HAST_KC_TILE")) {
        const long v = atol(t);
        if (v >= 256 && v <= 16384) c->tile_bases = (uint32_t)(v & ~31l);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    hast_status st = HAST_OK;
    auto bail = [&](hipError_t he, const char *what) {
        if (he != hipSuccess && st == HAST_OK)
            st = set_error(he == hipErrorOutOfMemory ? HAST_ERR_OOM : HAST_ERR_HIP, "%s: %s", what, hipGetErrorString(he));
    };
    bail(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate");
    bail(dev_malloc(&c->d_small, kSmallWords * sizeof(unsigned long long)), "dev_malloc(small)");
    bail(dev_malloc(&c->d_err, 4 * sizeof(uint32_t)), "dev_malloc(err)");
    bail(dev_malloc(&c->d_histo, (HAST_KC_HISTO_HIGH + 2) * sizeof(unsigned long long)), "dev_malloc(histo)");
    for (auto &s : c->stage) bail(hipEventCreateWithFlags(&s.done, hipEventDisableTiming), "hipEventCreate");
    bail(hipEventCreateWithFlags(&c->err_ev, hipEventDisableTiming), "hipEventCreate");
    bail(pinned_malloc(reinterpret_cast<void **>(&c->h_err), sizeof(uint32_t), hipHostMallocDefault), "pinned_malloc(err)");
    bail(hipEventCreateWithFlags(&c->cur_ev, hipEventDisableTiming), "hipEventCreate");
    bail(pinned_malloc(reinterpret_cast<void **>(&c->h_cursor), sizeof(unsigned long long), hipHostMallocDefault), "pinned_malloc(cursor)");
    if (c->h_err) *c->h_err = 0;
    if (st == HAST_OK) {
        size_t free_b = 0, total_b = 0;
        bail(hipMemGetInfo(&free_b, &total_b), "hipMemGetInfo");
        const size_t most = (size_t)((double)free_b * 0.85);
        if (table_bytes == 0 || table_bytes > most) table_bytes = most;      // never more than 85 % of what is free
        size_t nb = table_bytes / (kKcBucketWords * sizeof(unsigned long long));
        nb = std::min<size_t>(std::max<size_t>(nb, 64), 0xFFFFFFF0u);
        // whole slices of the partitioned path (512 or 1024 buckets, kc_common.h kc_fine_of_hash): a minimizer names every slice with the
        // same probability, so a last slice of a few buckets would take a whole slice's keys
        if (nb > 1024) nb &= ~(size_t)1023;
        else if (nb > 512) nb = 512;
        c->nbuckets = (uint32_t)nb;
        bail(dev_malloc(&c->d_table, nb * kKcBucketWords * sizeof(unsigned long long)), "dev_malloc(count table)");
    }
    if (st == HAST_OK) part_setup(c);
    if (st == HAST_OK) bail(hipMemsetAsync(c->d_small, 0, kSmallWords * sizeof(unsigned long long), c->stream), "hipMemset");
    if (st == HAST_OK) bail(hipMemsetAsync(c->d_err, 0, 4 * sizeof(uint32_t), c->stream), "hipMemset");
    if (st == HAST_OK && table_empty(c) != HAST_OK) st = HAST_ERR_HIP;
    if (st == HAST_OK) bail(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    if (st != HAST_OK) {
        hast_kc_destroy(c);
        return st;
    }
    *out = c;
    return HAST_OK;
}

void hast_kc_destroy(hast_kc *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &s : c->stage) {
        if (s.h) (void)hipHostFree(s.h);
        if (s.d) (void)hipFree(s.d);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    if (c->err_ev) (void)hipEventDestroy(c->err_ev);
    if (c->cur_ev) (void)hipEventDestroy(c->cur_ev);
    if (c->h_cursor) (void)hipHostFree(c->h_cursor);
    if (c->h_err) (void)hipHostFree(c->h_err);
    for (auto *p : c->d_sorted)
        if (p) (void)hipFree(p);
    if (c->d_table) (void)hipFree(c->d_table);
    for (void *p : {(void *)c->d_rec, (void *)c->d_l1, (void *)c->d_spill, (void *)c->d_fills})
        if (p) (void)hipFree(p);
    if (c->d_small) (void)hipFree(c->d_small);
    if (c->d_err) (void)hipFree(c->d_err);
    if (c->d_histo) (void)hipFree(c->d_histo);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

hast_stream hast_kc_stream(hast_kc *c) { return c ? (hast_stream)c->stream : nullptr; }

hast_status hast_kc_set_slice(hast_kc *c, uint32_t slice, uint32_t n_slices) {
    if (hast_status st = need_table(c)) return st;
    if (n_slices < 1 || slice >= n_slices) return set_error(HAST_ERR_INVALID, "slice %u of %u", slice, n_slices);
    c->slice = slice;
    c->n_slices = n_slices;
    if (hast_status st = table_empty(c)) return st;
    KC_TRY(hipMemsetAsync(c->d_small, 0, kSmallWords * sizeof(unsigned long long), c->stream));
    KC_TRY(hipMemsetAsync(c->d_err, 0, 4 * sizeof(uint32_t), c->stream));
    if (c->err_pending) {                          // a copy of the old word may still be in flight
        KC_TRY(hipEventSynchronize(c->err_ev));
        c->err_pending = false;
    }
    *c->h_err = 0;
    c->est_records = c->est_after = 0;             // (records of the old slice are void with its table; the cursor words were zeroed above)
    if (c->cur_pending) {
        KC_TRY(hipEventSynchronize(c->cur_ev));
        c->cur_pending = false;
    }
    return HAST_OK;
}

static hast_status count_launch(hast_kc *c, int parent, const uint8_t *d_bytes, size_t n_bytes, size_t n_starts) {
    if (n_starts == 0) return HAST_OK;
    KcCountArgs a;
    a.bytes = d_bytes;
    a.n_bytes = n_bytes;
    a.n_starts = n_starts;
    a.table = c->d_table;
    a.nbuckets = c->nbuckets;
    a.k = c->k;
    a.m = c->m;
    a.parent = (uint32_t)parent;
    a.slice = c->slice;
    a.n_slices = c->n_slices;
    a.tile_bases = c->tile_bases;
    a.tile_queue = c->d_small + kQueue;
    a.total = c->d_small + kTotal;
    a.err = c->d_err;
    a.rec_out = nullptr;
    a.rec_cap = 0;
    a.rec_cursor = nullptr;
    a.rec_chunk = 0;
    a.rec_run_max = a.rec_off_bits = a.fine_shift = 0;
    a.fresh = 0;
    a.spill = nullptr;
    a.spill_cap = 0;
    a.spill_n = nullptr;
    if (!c->part_on)
        if (hast_status st = table_real(c)) return st;
    if (c->part_on) {
        // (an upper bound of one record per two windows; a minimizer run holds ~3.5.  What does not fit the buffer after all is
        // counted on the spot, by the kernel itself)
        // ... plus the chunk a workgroup has in hand when it ends (filled with null records)
        a.rec_chunk = 2 * a.tile_bases;
        a.rec_run_max = kc_run_max(c->k, c->m);
        a.rec_off_bits = kc_rec_off_bits(c->k, c->m);
        a.fine_shift = c->fine_shift;
        const uint64_t worst = n_starts / 2 + 1 + (uint64_t)c->n_cu * 4 * a.rec_chunk;
        if (c->cur_pending && hipEventQuery(c->cur_ev) == hipSuccess) {
            c->est_records = std::min<uint64_t>(c->est_records, (uint64_t)*c->h_cursor + c->est_after);
            c->cur_pending = false;
        }
        if (c->est_records + worst > c->rec_cap) {     // before the table is swept: what does the device say, now?  (a flush waits for
            // everything launched so far as well)
            KC_TRY(hipMemcpyAsync(c->h_cursor, c->d_small + kRecCursor, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            KC_TRY(hipStreamSynchronize(c->stream));
            c->est_records = std::min<uint64_t>(c->est_records, (uint64_t)*c->h_cursor);
            c->cur_pending = false;
            c->est_after = 0;
        }
        if (c->est_records + worst > c->rec_cap)
            if (hast_status st = part_flush(c)) return st;
        a.rec_out = c->d_rec;
        a.rec_cap = c->rec_cap;
        a.rec_cursor = c->d_small + kRecCursor;
        a.fresh = c->fresh ? 1u : 0u;                  // (after a flush above: no longer)
        a.spill = c->d_spill;
        a.spill_cap = c->spill_cap;
        a.spill_n = c->d_small + kSpillN;
        c->est_records += worst;
    }
    const size_t n_tiles = (n_starts + a.tile_bases - 1) / a.tile_bases;
    const unsigned grid = (unsigned)std::min<size_t>(n_tiles, (size_t)c->n_cu * (c->part_on ? 4 : 8));
    KC_TRY(hipMemsetAsync(c->d_small + kQueue, 0, sizeof(unsigned long long), c->stream));
    KC_TRY(launch_kc_count(a, grid, c->stream));
    if (c->part_on) {
        if (!c->cur_pending) {
            KC_TRY(hipMemcpyAsync(c->h_cursor, c->d_small + kRecCursor, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            KC_TRY(hipEventRecord(c->cur_ev, c->stream));
            c->cur_pending = true;
            c->est_after = 0;
        } else c->est_after += n_starts / 2 + 1 + (uint64_t)c->n_cu * 4 * a.rec_chunk;
    }
    return HAST_OK;
}

hast_status hast_kc_count_device(hast_kc *c, int parent, const uint8_t *d_bytes, size_t n_bytes) {
    if (hast_status st = need_table(c)) return st;
    if (hast_status st = check_parent(parent)) return st;
    if (n_bytes && !d_bytes) return set_error(HAST_ERR_INVALID, "d_bytes is null");
    return count_launch(c, parent, d_bytes, n_bytes, n_bytes);
}

hast_status hast_kc_count(hast_kc *c, int parent, const uint8_t *bytes, size_t n_bytes) {
    if (hast_status st = need_table(c)) return st;
    if (hast_status st = check_parent(parent)) return st;
    if (n_bytes && !bytes) return set_error(HAST_ERR_INVALID, "bytes is null");
    const size_t overlap = (size_t)c->k - 1, chunk = kStageBytes - 64;
    // a table that is already full makes the rest of the pass pointless: say so as soon as the device's word has arrived
    if (c->err_pending && hipEventQuery(c->err_ev) == hipSuccess) {
        c->err_pending = false;
        if (*c->h_err & 1) return set_error(HAST_ERR_TABLE_FULL, "k-mer count table full (%u buckets of %d keys, slice %u of %u): use more slices",
                                            c->nbuckets, kKcSlots, c->slice, c->n_slices);
    }
    for (size_t at = 0; at < n_bytes; at += chunk) {
        KcStage &s = c->stage[c->turn++ & 1];
        if (!s.h) {
            KC_TRY(pinned_malloc(reinterpret_cast<void **>(&s.h), kStageBytes, hipHostMallocDefault));
            KC_TRY(dev_malloc(reinterpret_cast<void **>(&s.d), kStageBytes));
        }
        if (s.busy) {
            KC_TRY(hipEventSynchronize(s.done));
            s.busy = false;
        }
        const size_t starts = std::min(chunk, n_bytes - at);
        const size_t avail = std::min(starts + overlap, n_bytes - at);       // windows near the cut need the next K-1 bytes
        memcpy(s.h, bytes + at, avail);
        KC_TRY(hipMemcpyAsync(s.d, s.h, avail, hipMemcpyHostToDevice, c->stream));
        if (hast_status st = count_launch(c, parent, s.d, avail, starts)) return st;
        KC_TRY(hipEventRecord(s.done, c->stream));
        s.busy = true;
    }
    if (!c->err_pending && (c->turn & 7) == 0) {
        KC_TRY(hipMemcpyAsync(c->h_err, c->d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        KC_TRY(hipEventRecord(c->err_ev, c->stream));
        c->err_pending = true;
    }
    return HAST_OK;
}

hast_status hast_kc_sync(hast_kc *c) {
    if (hast_status st = use(c)) return st;
    if (c->d_table)
        if (hast_status st = part_flush(c)) return st;
    uint32_t e = 0;
    KC_TRY(hipMemcpyAsync(&e, c->d_err, sizeof(e), hipMemcpyDeviceToHost, c->stream));
    KC_TRY(hipStreamSynchronize(c->stream));
    for (auto &s : c->stage) s.busy = false;
    c->err_pending = false;
    if (e & 1) return set_error(HAST_ERR_TABLE_FULL, "k-mer count table full (%u buckets of %d keys, slice %u of %u): use more slices",
                                c->nbuckets, kKcSlots, c->slice, c->n_slices);
    return HAST_OK;
}

hast_status hast_kc_stats(hast_kc *c, uint64_t out[6]) {
    if (hast_status st = need_table(c)) return st;
    if (!out) return set_error(HAST_ERR_INVALID, "out is null");
    if (hast_status st = part_flush(c)) return st;
    unsigned long long h[kSmallWords];
    KC_TRY(hipMemsetAsync(c->d_small + kStats, 0, 3 * sizeof(unsigned long long), c->stream));
    KC_TRY(launch_kc_stats(c->d_table, c->nbuckets, c->d_small + kStats, c->stream));
    KC_TRY(hipMemcpyAsync(h, c->d_small, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    KC_TRY(hipStreamSynchronize(c->stream));
    out[0] = h[kStats];
    out[1] = h[kStats + 1];
    out[2] = h[kStats + 2];
    out[3] = (uint64_t)c->nbuckets * kKcSlots;
    out[4] = h[kTotal];
    out[5] = h[kTotal + 1];
    return HAST_OK;
}

hast_status hast_kc_histo(hast_kc *c, int parent, uint64_t *histo) {
    if (hast_status st = need_table(c)) return st;
    if (hast_status st = check_parent(parent)) return st;
    if (!histo) return set_error(HAST_ERR_INVALID, "histo is null");
    if (hast_status st = part_flush(c)) return st;
    const size_t n = HAST_KC_HISTO_HIGH + 2;
    std::vector<unsigned long long> h(n);
    KC_TRY(hipMemsetAsync(c->d_histo, 0, n * sizeof(unsigned long long), c->stream));
    KC_TRY(launch_kc_histo(c->d_table, c->nbuckets, (uint32_t)parent, c->d_histo, c->stream));
    KC_TRY(hipMemcpyAsync(h.data(), c->d_histo, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    KC_TRY(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; ++i) histo[i] += h[i];
    return HAST_OK;
}

void hast_kc_find_bounds(const uint64_t *histo, long out[4]) {
    // find_bounds.awk:8-25 over the rows `jellyfish histo` prints (ascending count, empty rows absent).  The awk program
    // tests a variable it never assigns (so: 0) until the descent ends; the row that ends the descent is not a
    // candidate for the maximum.  awk compares numbers as doubles.
    double lowest = 0, highest = 0;
    long min_index = 0, max_index = 0;
    bool descending = true;
    for (long i = 1; i <= (long)HAST_KC_HISTO_HIGH + 1; ++i) {
        if (!histo[i]) continue;
        const double c = (double)histo[i];
        if (descending) {
            if (lowest == 0 || c < lowest) {
                lowest = c;
                min_index = i;
            } else descending = false;
        } else if (highest == 0 || c > highest) {
            highest = c;
            max_index = i;
        }
    }
    out[0] = min_index;
    out[1] = max_index;
    out[2] = min_index + 1;                              // LOWER_INDEX (find_bounds.awk:29)
    out[3] = 3 * max_index - 2 * min_index - 1;          // UPPER_INDEX (find_bounds.awk:28,30)
}

hast_status hast_kc_select(hast_kc *c, int parent, uint32_t lower, uint32_t upper, size_t *n_added) {
    if (hast_status st = need_table(c)) return st;
    if (hast_status st = check_parent(parent)) return st;
    if (n_added) *n_added = 0;
    if (hast_status st = part_flush(c)) return st;
    unsigned long long n = 0;
    unsigned long long *cur = c->d_small + kCursor;
    KC_TRY(hipMemsetAsync(cur, 0, sizeof(unsigned long long), c->stream));
    KC_TRY(launch_kc_select(c->d_table, c->nbuckets, (uint32_t)parent, lower, upper, c->k, nullptr, 0, cur, c->stream));
    KC_TRY(hipMemcpyAsync(&n, cur, sizeof(n), hipMemcpyDeviceToHost, c->stream));
    KC_TRY(hipStreamSynchronize(c->stream));
    if (n == 0) return HAST_OK;
    unsigned long long *d_out = nullptr;
    KC_TRY(dev_malloc(reinterpret_cast<void **>(&d_out), n * sizeof(unsigned long long)));
    hast_status st = HAST_OK;
    auto step = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && st == HAST_OK) st = set_error(HAST_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    };
    step(hipMemsetAsync(cur, 0, sizeof(unsigned long long), c->stream), "hipMemset");
    step(launch_kc_select(c->d_table, c->nbuckets, (uint32_t)parent, lower, upper, c->k, d_out, (size_t)n, cur, c->stream), "select");
    std::vector<uint64_t> &v = c->sel[parent];
    const size_t old = v.size();
    if (st == HAST_OK) {
        v.resize(old + n);
        step(hipMemcpyAsync(v.data() + old, d_out, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream), "D2H");
        step(hipStreamSynchronize(c->stream), "sync");
    }
    (void)hipFree(d_out);
    if (st != HAST_OK) {
        v.resize(old);
        return st;
    }
    if (n_added) *n_added = (size_t)n;
    return HAST_OK;
}

hast_status hast_kc_selection_clear(hast_kc *c) {
    if (hast_status st = use(c)) return st;
    for (int p = 0; p < 2; ++p) {
        std::vector<uint64_t>().swap(c->sel[p]);
        if (c->d_sorted[p]) KC_TRY(hipFree(c->d_sorted[p]));
        c->d_sorted[p] = nullptr;
        c->n_sorted[p] = 0;
    }
    return HAST_OK;
}

hast_status hast_kc_selection_adopt(hast_kc *dst, hast_kc *src) {
    if (!dst || !src || dst == src) return set_error(HAST_ERR_INVALID, "selection_adopt needs two different contexts");
    if (dst->k != src->k) return set_error(HAST_ERR_INVALID, "selection_adopt: K differs (%d vs %d)", dst->k, src->k);
    for (int p = 0; p < 2; ++p) {
        dst->sel[p].insert(dst->sel[p].end(), src->sel[p].begin(), src->sel[p].end());
        std::vector<uint64_t>().swap(src->sel[p]);
    }
    return HAST_OK;
}

// out[0] = 1 when windows go through the partitioned path, out[1] flushes so far, out[2] records applied, out[3] windows that took
// the atomic path after all (full buckets beyond a slice), out[4] capacity of the record buffer
hast_status hast_kc_partition_info(hast_kc *c, uint64_t out[5]) {
    if (!c || !out) return set_error(HAST_ERR_INVALID, "null argument");
    out[0] = c->part_on ? 1 : 0;
    out[1] = c->n_flushes;
    out[2] = c->n_flushed_records;
    out[3] = c->n_spilled;
    out[4] = c->rec_cap;
    return HAST_OK;
}

hast_status hast_kc_release_table(hast_kc *c) {
    if (hast_status st = use(c)) return st;
    KC_TRY(hipStreamSynchronize(c->stream));
    if (c->d_table) KC_TRY(hipFree(c->d_table));
    c->d_table = nullptr;
    for (void *p : {(void *)c->d_rec, (void *)c->d_l1, (void *)c->d_spill, (void *)c->d_fills})
        if (p) KC_TRY(hipFree(p));
    c->d_rec = c->d_l1 = c->d_spill = nullptr;
    c->d_fills = nullptr;
    c->part_on = false;
    for (auto &s : c->stage) {
        if (s.d) KC_TRY(hipFree(s.d));
        s.d = nullptr;
        if (s.h) KC_TRY(hipHostFree(s.h));
        s.h = nullptr;
    }
    return HAST_OK;
}

hast_status hast_kc_selection_sort(hast_kc *c, int parent, size_t *n_out) {
    if (hast_status st = use(c)) return st;
    if (hast_status st = check_parent(parent)) return st;
    std::vector<uint64_t> &v = c->sel[parent];
    if (c->d_sorted[parent]) {
        KC_TRY(hipFree(c->d_sorted[parent]));
        c->d_sorted[parent] = nullptr;
        c->n_sorted[parent] = 0;
    }
    const size_t n = v.size();
    if (n_out) *n_out = n;
    if (n == 0) return HAST_OK;
    unsigned long long *d_in = nullptr, *d_out = nullptr;
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    hast_status st = HAST_OK;
    auto step = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && st == HAST_OK)
            st = set_error(e == hipErrorOutOfMemory ? HAST_ERR_OOM : HAST_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    };
    step(dev_malloc(reinterpret_cast<void **>(&d_in), n * sizeof(unsigned long long)), "dev_malloc(sort in)");
    if (st == HAST_OK) step(dev_malloc(reinterpret_cast<void **>(&d_out), n * sizeof(unsigned long long)), "dev_malloc(sort out)");
    if (st == HAST_OK) step(hipMemcpyAsync(d_in, v.data(), n * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream), "H2D");
    if (st == HAST_OK) step(kc_sort_keys(nullptr, &tmp_bytes, d_in, d_out, n, c->k, c->stream), "sort (size query)");
    if (st == HAST_OK) step(dev_malloc(&d_tmp, tmp_bytes ? tmp_bytes : 16), "dev_malloc(sort temp)");
    if (st == HAST_OK) step(kc_sort_keys(d_tmp, &tmp_bytes, d_in, d_out, n, c->k, c->stream), "sort");
    if (st == HAST_OK) step(hipStreamSynchronize(c->stream), "sync");
    if (d_tmp) (void)hipFree(d_tmp);
    if (d_in) (void)hipFree(d_in);
    if (st != HAST_OK) {
        if (d_out) (void)hipFree(d_out);
        return st;
    }
    c->d_sorted[parent] = d_out;
    c->n_sorted[parent] = n;
    std::vector<uint64_t>().swap(v);
    return HAST_OK;
}

static hast_status selection_rows(hast_kc *c, int parent, size_t first, size_t count) {
    if (hast_status st = use(c)) return st;
    if (hast_status st = check_parent(parent)) return st;
    if (first + count > c->n_sorted[parent] || first + count < first)
        return set_error(HAST_ERR_INVALID, "rows [%zu, %zu) outside the sorted selection of %zu", first, first + count, c->n_sorted[parent]);
    return HAST_OK;
}

hast_status hast_kc_selection_text(hast_kc *c, int parent, size_t first, size_t count, char *out) {
    if (hast_status st = selection_rows(c, parent, first, count)) return st;
    if (count == 0) return HAST_OK;
    if (!out) return set_error(HAST_ERR_INVALID, "out is null");
    const size_t bytes = count * (size_t)(c->k + 1);
    char *d_text = nullptr;
    KC_TRY(dev_malloc(reinterpret_cast<void **>(&d_text), bytes));
    hipError_t e = launch_kc_format(c->d_sorted[parent] + first, count, c->k, d_text, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_text, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_text);
    if (e != hipSuccess) return set_error(HAST_ERR_HIP, "formatting the selection: %s", hipGetErrorString(e));
    return HAST_OK;
}

hast_status hast_kc_selection_keys(hast_kc *c, int parent, size_t first, size_t count, uint64_t *out) {
    if (hast_status st = selection_rows(c, parent, first, count)) return st;
    if (count == 0) return HAST_OK;
    if (!out) return set_error(HAST_ERR_INVALID, "out is null");
    unsigned long long *d_keys = nullptr;
    KC_TRY(dev_malloc(reinterpret_cast<void **>(&d_keys), count * sizeof(unsigned long long)));
    hipError_t e = launch_kc_to_table_keys(c->d_sorted[parent] + first, count, c->k, d_keys, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_keys, count * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_keys);
    if (e != hipSuccess) return set_error(HAST_ERR_HIP, "converting the selection: %s", hipGetErrorString(e));
    return HAST_OK;
}

hast_status hast_kc_synth_host(const hast_kc_synth *p, int parent, uint64_t first_read, size_t n_reads, uint8_t *out) {
    if (hast_status st = check_synth(p)) return st;
    if (hast_status st = check_parent(parent)) return st;
    if (n_reads && !out) return set_error(HAST_ERR_INVALID, "out is null");
    const KcSynth g = resolve(p);
    const size_t rec = (size_t)g.read_len + 1;
    for (size_t i = 0; i < n_reads; ++i)
        for (uint32_t j = 0; j < rec; ++j) out[i * rec + j] = kc_synth_byte(g, parent, first_read + i, j);
    return HAST_OK;
}

hast_status hast_kc_synth_device(hast_kc *c, const hast_kc_synth *p, int parent, uint64_t first_read, size_t n_reads, uint8_t *d_out) {
    if (hast_status st = use(c)) return st;
    if (hast_status st = check_synth(p)) return st;
    if (hast_status st = check_parent(parent)) return st;
    if (n_reads && !d_out) return set_error(HAST_ERR_INVALID, "d_out is null");
    KC_TRY(launch_kc_synth(resolve(p), parent, first_read, n_reads * ((size_t)p->read_len + 1), d_out, c->stream));
    return HAST_OK;
}

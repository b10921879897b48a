// fq_api.cpp -- hast_fq_*: a FASTQ stream framed on the GPU (include/hast.h, "FASTQ framing").  The caller puts raw file bytes
// into pinned buffers; everything the reference's producer thread does per record (classify.cpp:238-278: four getlines,
// parseName) happens in fq_kernels.hip, the reads are classified where they lie in the raw block, and the host only has to
// name the barcodes (text -> dense id, one dictionary per job) before the per-barcode bookkeeping runs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/hast.h"
#include "fq_device.h"
#include "hast_internal.h"

using namespace hast;

namespace {
#define FQ_TRY(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return set_error(e_ == hipErrorOutOfMemory ? HAST_ERR_OOM : HAST_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct Slot {
    uint8_t *h_buf = nullptr, *d_buf = nullptr;        // pad + block bytes (+ slack)
    size_t h_buf_bytes = 0;
    FqState *d_st = nullptr, *h_st = nullptr;          // h_st pinned
    uint32_t *d_tile = nullptr, *d_nl = nullptr;
    uint64_t *d_off = nullptr;
    uint32_t *d_len = nullptr, *d_bcpos = nullptr, *d_bclen = nullptr, *d_ids = nullptr, *d_votes = nullptr;
    uint32_t *d_text = nullptr;                        // [max_rec][4]: the barcode text record of EVERY record of the block, for the naming kernel
    uint32_t *h_nids = nullptr;                        // pinned: ids the device dictionary had handed out when this block was named
    uint32_t *h_unknown = nullptr;                     // pinned [1 + h_cap]: count, then the records the cache did not know
    NamePub *h_pubs = nullptr;                         // pinned [h_cap]: what this block teaches the cache
    hipEvent_t named = nullptr;
    bool use_cache = false;                            // this block's ids came from the cache (unknown list valid)
    bool host_texts = false;                           // ... and the pinned copy of its text records is complete (not a block that outgrew the arrays)
    bool named_early = false;                          // a dictionary named this block right behind its framing, on the framing's stream (`named` is there)
    uint32_t *h_bc = nullptr, *h_ids = nullptr;        // pinned: h_bc = [pos x h_cap | len x h_cap], written by the records kernel itself
    size_t h_cap = 0;                                  // records the pinned arrays hold (a block with more: grown, copied)
    size_t k_cap = 0;                                  // ... the capacity the kernel of the submitted block was given
    size_t n_bytes = 0;
    int last = 0;
    hipEvent_t copied = nullptr, parsed = nullptr, done = nullptr;
    enum { FREE, ACQUIRED, SUBMITTED, OPEN } state = FREE;
    bool done_pending = false;
    // striped streams
    int lane = 0;                                      // which context / GPU this slot's buffers live on
    FqState *h_cnt = nullptr;                          // pinned: the newline count of the block's own bytes lands here
    hipEvent_t counted = nullptr, over_copied = nullptr;
    size_t n_over = 0;                                 // bytes of the next block behind this block's own bytes
    bool over_ready = false, eof_view = false, first_of_file = false;
    uint8_t last_byte = 0;
    uint8_t *h_last = nullptr;                         // pinned: the last byte of a block that was filled on the device (striped streams)
    hipEvent_t over_read = nullptr;                    // ... and, behind the copy of its first bytes into the view of the block in front of it (on that
    bool over_read_pending = false;                    //     block's GPU), the event the next filling of this buffer waits for
    bool dev_src = false;                              // the block's bytes were written on the device (hast_fq_submit_device): h_buf holds nothing
    bool host_view = false;                            // ... until hast_fq_block_host_bytes has fetched them
    // routing streams (hast_fq_set_route): class + extent of every record, the tiles' prefix sums, the routed bytes and their host copy
    uint32_t *d_rstart = nullptr, *d_rlen = nullptr, *d_rtile = nullptr;
    uint8_t *d_rcls = nullptr, *d_out = nullptr, *h_out = nullptr;
    RouteState *d_rs = nullptr, *h_rs = nullptr;       // h_rs pinned
    size_t route_rec = 0;                              // record slots the routing arrays hold
    std::vector<uint32_t> v_rstart, v_rlen;            // host copies for a block the caller routes itself (rare)
    std::vector<uint8_t> v_rcls, v_tail;
};
}  // namespace

struct hast_names {            // device-side cache barcode text -> id of one GPU, shared by the FASTQ streams of that GPU
    hast_ctx *ctx = nullptr;
    int device = 0;
    NameEntry *d_tab = nullptr;
    uint32_t mask = 0;
    size_t count = 0, limit = 0;                       // entries published / published at most (half the slots)
    // dictionary mode (hast_names_create_dict): the table hands out the ids itself -- ids 0 .. limit-1, one counter, the texts by id
    bool dict = false;
    bool name_early = false;                           // HAST_NAME_EARLY=1 (measurements): the naming kernel behind the framing kernels, on their stream (below)
    uint32_t *d_n_ids = nullptr;
    void *d_text_of_id = nullptr;                      // [limit][16]
};

struct FqLane {                // what a striped stream keeps per context: its GPU's streams and name cache
    hast_ctx *ctx = nullptr;
    hast_names *names = nullptr;
    int device = 0;
    hipStream_t copy_stream = nullptr, parse_stream = nullptr;
};

struct hast_fq {
    hast_ctx *ctx = nullptr;
    hast_names *names = nullptr;
    int device = 0, k = 0;
    // striped: block i of the stream lives on lane i % lanes.size(); framed from the newline count of the blocks in front of it
    bool striped = false;
    std::vector<FqLane> lanes;
    size_t over_cap = 0;                                       // bytes of the next block a block's view may reach into
    size_t n_framed = 0;                                       // blocks whose framing has been launched
    size_t n_device_blocks = 0;                                // hast_fq_device_block calls (device-side blocks)
    uint64_t nl_before = 0;                                    // newlines of the current file in front of block n_framed
    int source = 0;                                            // 0 = not decided, 1 = host blocks (hast_fq_submit), 2 = device blocks (hast_fq_submit_device)
    hast_status failed = HAST_OK;                              // sticky failure of the striped framing path (advance_striped)
    std::string fail_msg;
    std::vector<uint64_t> records_per_lane;
    size_t block = 0, pad = 0, max_rec = 0;
    std::vector<Slot> slots;
    size_t n_acquired = 0, n_submitted = 0, n_opened = 0;      // blocks handed out / submitted / returned by hast_fq_next
    int prev_submitted = -1;                                   // slot of the previous block of this file (device-side tail)
    // raw bytes + framing run on a stream of their own, so that the copies of the blocks submitted ahead do not queue up in
    // front of the (short) per-block work hast_fq_next / hast_fq_commit put on the context's stream
    hipStream_t copy_stream = nullptr, parse_stream = nullptr;     // H2D of block k+1 runs while block k is framed
    std::vector<uint8_t> carry;                                // host copy of the previous block's tail (barcodes may lie in it)
    // routing (hast_fq_set_route): the blocks are not classified; their records leave the GPU as four runs by barcode class
    bool route = false;
    std::vector<hast_names *> route_tab;                       // per lane (one for a plain stream): barcode text -> class on that GPU
};

// Every per-record array of a slot has ONE capacity, h_cap, and is only ever resized here: h_bc = [pos | len | 4 words of
// text] x cap, h_ids, h_unknown (1 + cap), h_pubs.  The records kernel is handed h_cap and writes text records at
// h_bc + 2*h_cap (the first h_cap records) and d_text[4*i] (every record: the device array holds the whole record table); the naming
// kernel writes up to n ids and n + 1 words of h_unknown, so a block with more records than h_cap grows the arrays BEFORE it is named;
// hast_fq_commit fills up to n h_pubs.
static hast_status grow_records(Slot &s, size_t cap) {
    // (parked, not freed: hipFree / hipHostFree wait for every stream of the device, hast_internal.h)
    park_pinned(s.h_bc, 6 * s.h_cap * sizeof(uint32_t), 3);
    park_pinned(s.h_ids, s.h_cap * sizeof(uint32_t), 3);
    park_pinned(s.h_unknown, (1 + s.h_cap) * sizeof(uint32_t), 3);
    park_pinned(s.h_pubs, s.h_cap * sizeof(NamePub), 3);
    s.h_bc = s.h_ids = s.h_unknown = nullptr;
    s.h_pubs = nullptr;
    s.h_cap = 0;
    FQ_TRY(pinned_malloc((void **)&s.h_bc, (2 + 4) * cap * sizeof(uint32_t), hipHostMallocDefault));
    FQ_TRY(pinned_malloc((void **)&s.h_ids, cap * sizeof(uint32_t), hipHostMallocDefault));
    FQ_TRY(pinned_malloc((void **)&s.h_unknown, (1 + cap) * sizeof(uint32_t), hipHostMallocDefault));
    FQ_TRY(pinned_malloc((void **)&s.h_pubs, cap * sizeof(NamePub), hipHostMallocDefault));
    s.h_cap = cap;
    return HAST_OK;
}

static void free_slot(Slot &s) {
    park_pinned(s.h_cnt, 64, 3);
    park_pinned(s.h_last, 64, 3);
    if (s.over_read) (void)hipEventDestroy(s.over_read);
    if (s.counted) (void)hipEventDestroy(s.counted);
    if (s.over_copied) (void)hipEventDestroy(s.over_copied);
    park_pinned(s.h_buf, s.h_buf_bytes, 3);
    park_pinned(s.h_st, 64, 3);
    park_pinned(s.h_nids, 64, 3);
    park_pinned(s.h_bc, 6 * s.h_cap * sizeof(uint32_t), 3);
    park_pinned(s.h_ids, s.h_cap * sizeof(uint32_t), 3);
    park_pinned(s.h_unknown, (1 + s.h_cap) * sizeof(uint32_t), 3);
    park_pinned(s.h_pubs, s.h_cap * sizeof(NamePub), 3);
    park_device(s.d_text, 0, 3);
    if (s.named) (void)hipEventDestroy(s.named);
    for (void *p : {(void *)s.d_buf, (void *)s.d_st, (void *)s.d_tile, (void *)s.d_nl, (void *)s.d_off, (void *)s.d_len, (void *)s.d_bcpos,
                    (void *)s.d_bclen, (void *)s.d_ids, (void *)s.d_votes, (void *)s.d_rstart, (void *)s.d_rlen, (void *)s.d_rtile, (void *)s.d_rcls,
                    (void *)s.d_out, (void *)s.d_rs})
        park_device(p, (p == (void *)s.d_buf || p == (void *)s.d_out) ? s.h_buf_bytes : 0, 3);
    park_pinned(s.h_out, s.h_buf_bytes, 3);
    park_pinned(s.h_rs, 64, 3);
    if (s.copied) (void)hipEventDestroy(s.copied);
    if (s.parsed) (void)hipEventDestroy(s.parsed);
    if (s.done) (void)hipEventDestroy(s.done);
    s = Slot();
}

static int dev_of(const hast_fq *f, const Slot &s) { return f->striped ? f->lanes[(size_t)s.lane].device : f->device; }
static hast_ctx *ctx_of(const hast_fq *f, const Slot &s) { return f->striped ? f->lanes[(size_t)s.lane].ctx : f->ctx; }
static hast_names *names_of(const hast_fq *f, const Slot &s) { return f->striped ? f->lanes[(size_t)s.lane].names : f->names; }

// Tried in round 6, OFF by default (HAST_NAME_EARLY=1): a dictionary naming a block right behind its framing, on the framing's stream --
// the naming kernel needs nothing from the host, and on the context's stream it queues behind the classification of the blocks in front,
// so that hast_fq_next waits ~0.8 ms a block (0.8 s of a 1.7-s read phase at BASELINE config 2's size).  With it that wait is gone
// (0.02 s) -- and the read phase is no shorter: the time reappears as waiting for the .gz stream's bytes (1.94-1.96 s against 1.63-1.70 s,
// plain files 1.49-1.59 against 1.52-1.75 s, alternating: profiles/round6_cli_c2_ab_naming.txt).  The host's wait was not what bounded
// the pipeline; the kernel on the framing stream delays the next block's framing instead.
static hast_status enqueue_names_early(Slot &s, hast_names *nm, hipStream_t hs) {
    s.named_early = false;
    if (!nm || !nm->dict || !nm->name_early || !s.h_cap) return HAST_OK;
    s.h_unknown[0] = 0;
    FQ_TRY(launch_fq_name_claim_framed(s.d_text, s.d_st, (uint32_t)s.h_cap, nm->d_tab, nm->mask, nm->d_n_ids, (uint32_t)nm->limit, nm->d_text_of_id, s.h_ids, s.h_unknown,
                                       s.d_ids, hs));
    FQ_TRY(hipMemcpyAsync(s.h_nids, nm->d_n_ids, sizeof(uint32_t), hipMemcpyDeviceToHost, hs));
    FQ_TRY(hipEventRecord(s.named, hs));
    s.named_early = true;
    return HAST_OK;
}

// Routing streams: behind the framing of a block, on the same stream -- class and extent of every record, prefix sums, the records copied
// into four runs, the runs and their sizes to pinned host memory (fq_kernels.hip, "routing").  view_bytes bounds what the runs can hold.
static hast_status enqueue_route(hast_fq *f, Slot &s, hast_names *tab, bool striped, int last, size_t view_bytes, hipStream_t hs) {
    (void)f;
    FQ_TRY(launch_fq_route(s.d_buf, s.d_st, s.d_nl, striped ? 1 : 0, last, tab ? tab->d_tab : nullptr, tab ? tab->mask : 0, s.d_rstart, s.d_rlen, s.d_rcls, s.d_rtile,
                           (uint32_t)s.route_rec, s.d_rs, s.d_out, hs));
    FQ_TRY(hipMemcpyAsync(s.h_rs, s.d_rs, sizeof(RouteState), hipMemcpyDeviceToHost, hs));
    if (view_bytes) FQ_TRY(hipMemcpyAsync(s.h_out, s.d_out, std::min(view_bytes, s.h_buf_bytes), hipMemcpyDeviceToHost, hs));
    return HAST_OK;
}

// Striped streams: launch the framing of every block that can be framed now, in order.  Block j needs (a) the bytes of block
// j + 1 that its records may reach into (there once j + 1 has been submitted, or j ends its file) and (b) the number of
// newlines in front of it: the count of block j - 1's own bytes (a kernel of its own, queued right behind that block's
// upload) added to the running sum of the file.  wait: block on (b) instead of giving up (hast_fq_next needs its block).
static hast_status advance_striped_once(hast_fq *f, bool wait, size_t upto);
// A failure on this path (an event query, a launch, a device switch) is STICKY: the stream stays failed, hast_fq_poll then says
// "go on" so that the caller's next hast_fq_next returns the error instead of waiting for a block that will never be framed.
static hast_status advance_striped(hast_fq *f, bool wait, size_t upto) {
    if (f->failed != HAST_OK) return set_error(f->failed, "%s", f->fail_msg.c_str());
    const hast_status st = advance_striped_once(f, wait, upto);
    if (st != HAST_OK) {
        f->failed = st;
        f->fail_msg = hast_last_error();
    }
    return st;
}
static hast_status advance_striped_once(hast_fq *f, bool wait, size_t upto) {
    while (f->n_framed < f->n_submitted && f->n_framed < upto) {
        const size_t j = f->n_framed;
        Slot &s = f->slots[j % f->slots.size()];
        if (!s.over_ready) break;
        uint32_t phase = 0;
        int bol = 1;
        uint64_t nl_before = 0;                       // (f->nl_before moves on only once the framing launch has succeeded)
        if (!s.first_of_file) {
            Slot &pv = f->slots[(j - 1) % f->slots.size()];
            FQ_TRY(hipSetDevice(dev_of(f, pv)));
            if (wait) FQ_TRY(hipEventSynchronize(pv.counted));
            else {
                const hipError_t q = hipEventQuery(pv.counted);
                if (q == hipErrorNotReady) break;
                if (q != hipSuccess) return set_error(HAST_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(q));
            }
            nl_before = f->nl_before + pv.h_cnt->n_nl;
            bol = (pv.dev_src ? (pv.n_bytes ? pv.h_last[0] : (uint8_t)'\n') : pv.last_byte) == '\n';      // (h_last landed in front of `counted`)
        }
        phase = (uint32_t)(nl_before & 3);
        FqLane &ln = f->lanes[(size_t)s.lane];
        FQ_TRY(hipSetDevice(ln.device));
        if (s.n_over) FQ_TRY(hipStreamWaitEvent(ln.parse_stream, s.over_copied, 0));
        FQ_TRY(launch_fq_block_striped(s.d_buf, s.d_st, f->pad, s.n_bytes, s.n_over, phase, bol, s.eof_view ? 1 : 0, s.d_tile, s.d_nl, s.d_off, s.d_len,
                                       s.d_bcpos, s.d_bclen, s.h_bc, s.d_text, (uint32_t)s.h_cap, (uint32_t)f->k, ln.parse_stream));
        s.k_cap = s.h_cap;
        FQ_TRY(hipMemcpyAsync(s.h_st, s.d_st, sizeof(FqState), hipMemcpyDeviceToHost, ln.parse_stream));
        if (f->route) {
            if (hast_status st = enqueue_route(f, s, f->route_tab[(size_t)s.lane], true, s.eof_view ? 1 : 0, f->pad + s.n_bytes + s.n_over, ln.parse_stream)) return st;
        } else if (hast_status st = enqueue_names_early(s, ln.names, ln.parse_stream)) return st;
        FQ_TRY(hipEventRecord(s.parsed, ln.parse_stream));
        f->nl_before = nl_before;
        f->n_framed++;
    }
    return HAST_OK;
}

static hast_status submit_striped(hast_fq *f, size_t n_bytes, int last, bool dev_src) {
    const size_t i = f->n_submitted, S = f->slots.size();
    if (!last && n_bytes != f->block)
        return set_error(HAST_ERR_INVALID, "a striped stream takes full blocks (%zu bytes) except for the last one of a file, got %zu", f->block, n_bytes);
    // this slot held block i - S; the framing of block i - S + 1 read that block's newline count and last byte: it must have been
    // launched before they are overwritten
    if (i >= S)
        if (hast_status st = advance_striped(f, true, i - S + 2)) return st;
    Slot &s = f->slots[i % S];
    FqLane &ln = f->lanes[(size_t)s.lane];
    FQ_TRY(hipSetDevice(ln.device));
    s.n_bytes = n_bytes;
    s.last = last;
    s.n_over = 0;
    s.over_ready = last != 0;                      // the last block of a file has nothing behind it
    s.eof_view = last != 0;
    s.first_of_file = i == 0 || f->slots[(i - 1) % S].last != 0;
    s.dev_src = dev_src;
    s.host_view = !dev_src;
    if (!dev_src) {
        s.last_byte = n_bytes ? s.h_buf[f->pad + n_bytes - 1] : (uint8_t)'\n';
        if (n_bytes) FQ_TRY(hipMemcpyAsync(s.d_buf + f->pad, s.h_buf + f->pad, n_bytes, hipMemcpyHostToDevice, ln.copy_stream));
    }
    FQ_TRY(hipEventRecord(s.copied, ln.copy_stream));      // (a device block: behind the caller's writes, which it put on this stream)
    FQ_TRY(hipStreamWaitEvent(ln.parse_stream, s.copied, 0));
    // newlines of the block's own bytes, for the blocks behind it
    FQ_TRY(launch_fq_count_own(s.d_buf, s.d_st, f->pad, n_bytes, s.d_tile, ln.parse_stream));
    FQ_TRY(hipMemcpyAsync(s.h_cnt, s.d_st, sizeof(FqState), hipMemcpyDeviceToHost, ln.parse_stream));
    if (dev_src && n_bytes) FQ_TRY(hipMemcpyAsync(s.h_last, s.d_buf + f->pad + n_bytes - 1, 1, hipMemcpyDeviceToHost, ln.parse_stream));
    FQ_TRY(hipEventRecord(s.counted, ln.parse_stream));
    if (!s.first_of_file) {
        // the block in front gets the first bytes of this one behind its own: host copy (the barcode extents of its records may
        // point there) + upload on ITS GPU -- or, for blocks filled on the device, a copy from this block's GPU to that one
        Slot &pv = f->slots[(i - 1) % S];
        FqLane &pl = f->lanes[(size_t)pv.lane];
        // ... as far as its last record can reach: that record started in the block in front, so its header and base lines end at
        // the latest with the 4th newline of this block (a FASTQ record is four lines).  Copying the whole 1 MB a long read may
        // need cost ~0.15 ms of host memcpy per 16-MB block -- what made a file striped over several contexts slower than on one.
        // (Fewer than 4 newlines in the first over_cap bytes: long lines; then all of it, as before.  Too short a view would be a
        // loud HAST_ERR_FORMAT from the framer, never a cut read.  Device blocks: the host does not see the bytes, and a
        // device-to-device megabyte costs microseconds: all of it.)
        size_t ov = std::min(f->over_cap, n_bytes);
        if (!dev_src) {
            const uint8_t *p0 = s.h_buf + f->pad, *p = p0, *const pe = p0 + ov;
            int nl = 0;
            while (nl < 4 && p < pe && (p = static_cast<const uint8_t *>(memchr(p, '\n', (size_t)(pe - p))))) { ++nl; ++p; }
            if (nl == 4) ov = (size_t)(p - p0);
        }
        pv.n_over = ov;
        pv.eof_view = last != 0 && ov == n_bytes;  // the whole rest of the file is in its view
        if (ov && !dev_src) {
            memcpy(pv.h_buf + f->pad + pv.n_bytes, s.h_buf + f->pad, ov);
            FQ_TRY(hipSetDevice(pl.device));
            FQ_TRY(hipMemcpyAsync(pv.d_buf + f->pad + pv.n_bytes, pv.h_buf + f->pad + pv.n_bytes, ov, hipMemcpyHostToDevice, pl.copy_stream));
            FQ_TRY(hipEventRecord(pv.over_copied, pl.copy_stream));
        } else if (ov) {
            FQ_TRY(hipSetDevice(pl.device));
            FQ_TRY(hipStreamWaitEvent(pl.copy_stream, s.copied, 0));
            if (pl.device == ln.device) FQ_TRY(hipMemcpyAsync(pv.d_buf + f->pad + pv.n_bytes, s.d_buf + f->pad, ov, hipMemcpyDeviceToDevice, pl.copy_stream));
            else FQ_TRY(hipMemcpyPeerAsync(pv.d_buf + f->pad + pv.n_bytes, pl.device, s.d_buf + f->pad, ln.device, ov, pl.copy_stream));
            FQ_TRY(hipEventRecord(pv.over_copied, pl.copy_stream));
            FQ_TRY(hipEventRecord(s.over_read, pl.copy_stream));       // (this buffer's next filling waits for the copy out of it)
            s.over_read_pending = true;
        }
        pv.over_ready = true;
    }
    s.state = Slot::SUBMITTED;
    f->n_submitted++;
    return advance_striped(f, false, f->n_submitted);
}

extern "C" {

hast_status hast_names_create(hast_ctx *ctx, size_t max_barcodes, hast_names **out) {
    if (!ctx || !out) return set_error(HAST_ERR_INVALID, "null argument");
    *out = nullptr;
    FQ_TRY(hipSetDevice(hast_ctx_device(ctx)));
    size_t slots = 64;
    while (slots < 2 * std::max<size_t>(max_barcodes, 1) && slots < (1ull << 31)) slots <<= 1;
    hast_names *nm = new (std::nothrow) hast_names();
    if (!nm) return set_error(HAST_ERR_OOM, "host allocation failed");
    nm->ctx = ctx;
    nm->device = hast_ctx_device(ctx);
    hipError_t e = dev_malloc((void **)&nm->d_tab, slots * sizeof(NameEntry));
    if (e == hipSuccess) e = hipMemsetAsync(nm->d_tab, 0, slots * sizeof(NameEntry), ctx_stream_of(ctx));
    if (e != hipSuccess) {
        if (nm->d_tab) (void)hipFree(nm->d_tab);
        delete nm;
        return set_error(e == hipErrorOutOfMemory ? HAST_ERR_OOM : HAST_ERR_HIP, "name cache: %s", hipGetErrorString(e));
    }
    nm->mask = (uint32_t)(slots - 1);
    nm->limit = slots / 2;
    *out = nm;
    return HAST_OK;
}

// The table as the job's DICTIONARY on this GPU: it hands out the dense ids itself (0 .. hast_names_limit - 1, in the order in which its
// kernels meet new texts), so that no first sighting of a barcode goes through the host (classify.cpp:52-56: the map insert of
// BarcodeCache::IncrBarcodeHaps).  The host reads the texts by id once, at the end (hast_names_texts), for printing.
hast_status hast_names_create_dict(hast_ctx *ctx, size_t max_barcodes, hast_names **out) {
    if (hast_status st = hast_names_create(ctx, max_barcodes, out)) return st;
    hast_names *nm = *out;
    hipError_t e = dev_malloc((void **)&nm->d_n_ids, 64);
    if (e == hipSuccess) e = dev_malloc(&nm->d_text_of_id, nm->limit * 16);
    if (e == hipSuccess) e = hipMemsetAsync(nm->d_n_ids, 0, 64, ctx_stream_of(ctx));
    if (e != hipSuccess) {
        hast_names_destroy(nm);
        *out = nullptr;
        return set_error(e == hipErrorOutOfMemory ? HAST_ERR_OOM : HAST_ERR_HIP, "name dictionary: %s", hipGetErrorString(e));
    }
    nm->dict = true;
    if (const char *e = getenv("HAST_NAME_EARLY")) nm->name_early = atoi(e) != 0;
    return HAST_OK;
}
size_t hast_names_limit(const hast_names *nm) { return nm ? nm->limit : 0; }
// ids handed out so far (waits for the naming kernels queued on the context's stream)
hast_status hast_names_count(hast_names *nm, size_t *n_ids) {
    if (!nm || !n_ids) return set_error(HAST_ERR_INVALID, "null argument");
    *n_ids = 0;
    if (!nm->dict) { *n_ids = nm->count; return HAST_OK; }
    FQ_TRY(hipSetDevice(nm->device));
    uint32_t v = 0;
    FQ_TRY(hipMemcpy(&v, nm->d_n_ids, sizeof(v), hipMemcpyDeviceToHost));
    *n_ids = std::min<size_t>(v, nm->limit);
    return HAST_OK;
}
// the text records (16 bytes each: length byte + text) of ids [first, first + n)
hast_status hast_names_texts(hast_names *nm, size_t first, size_t n, uint8_t *out16) {
    if (!nm || (n && !out16)) return set_error(HAST_ERR_INVALID, "null argument");
    if (!nm->dict) return set_error(HAST_ERR_INVALID, "hast_names_texts: not a dictionary (hast_names_create_dict)");
    if (first + n > nm->limit) return set_error(HAST_ERR_INVALID, "hast_names_texts: ids [%zu, %zu) beyond the %zu the dictionary can hand out", first, first + n, nm->limit);
    FQ_TRY(hipSetDevice(nm->device));
    if (n) FQ_TRY(hipMemcpy(out16, static_cast<const uint8_t *>(nm->d_text_of_id) + 16 * first, 16 * n, hipMemcpyDeviceToHost));
    return HAST_OK;
}

// src's texts [first, first + n) through dst's naming kernel: their ids in dst's numbering, new texts claimed there.
hast_status hast_names_merge(hast_names *dst, hast_names *src, size_t first, size_t n, uint32_t *ids_out) {
    if (!dst || !src || (n && !ids_out)) return set_error(HAST_ERR_INVALID, "null argument");
    if (!dst->dict || !src->dict) return set_error(HAST_ERR_INVALID, "hast_names_merge: both must be dictionaries (hast_names_create_dict)");
    if (first + n > src->limit) return set_error(HAST_ERR_INVALID, "hast_names_merge: ids [%zu, %zu) beyond the source's %zu", first, first + n, src->limit);
    if (!n) return HAST_OK;
    FQ_TRY(hipSetDevice(src->device));
    FQ_TRY(hipStreamSynchronize(ctx_stream_of(src->ctx)));                     // (its naming kernels have filed the texts)
    FQ_TRY(hipSetDevice(dst->device));
    hipStream_t hs = ctx_stream_of(dst->ctx);
    uint8_t *d_text = nullptr;
    uint32_t *d_ids = nullptr, *d_unknown = nullptr;
    hipError_t e = dev_malloc(&d_text, 16 * n);
    if (e == hipSuccess) e = dev_malloc(&d_ids, n * sizeof(uint32_t));
    if (e == hipSuccess) e = dev_malloc(&d_unknown, (n + 1) * sizeof(uint32_t));
    const uint8_t *from = static_cast<const uint8_t *>(src->d_text_of_id) + 16 * first;
    if (e == hipSuccess) e = src->device == dst->device ? hipMemcpyAsync(d_text, from, 16 * n, hipMemcpyDeviceToDevice, hs)
                                                        : hipMemcpyPeerAsync(d_text, dst->device, from, src->device, 16 * n, hs);
    if (e == hipSuccess) e = hipMemsetAsync(d_unknown, 0, sizeof(uint32_t), hs);
    for (size_t at = 0; e == hipSuccess && at < n; at += 1u << 24) {           // (the kernel counts records in 32 bits: 16M a launch)
        const uint32_t m = (uint32_t)std::min<size_t>(n - at, 1u << 24);
        e = launch_fq_name_claim(reinterpret_cast<const uint32_t *>(d_text + 16 * at), m, dst->d_tab, dst->mask, dst->d_n_ids, (uint32_t)dst->limit, dst->d_text_of_id,
                                 d_ids + at, d_unknown, nullptr, hs);
    }
    uint32_t n_unknown = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(ids_out, d_ids, n * sizeof(uint32_t), hipMemcpyDeviceToHost, hs);
    if (e == hipSuccess) e = hipMemcpyAsync(&n_unknown, d_unknown, sizeof(uint32_t), hipMemcpyDeviceToHost, hs);
    if (e == hipSuccess) e = hipStreamSynchronize(hs);
    for (void *p : {(void *)d_text, (void *)d_ids, (void *)d_unknown})
        if (p) (void)hipFree(p);
    if (e != hipSuccess) return set_error(e == hipErrorOutOfMemory ? HAST_ERR_OOM : HAST_ERR_HIP, "hast_names_merge: %s", hipGetErrorString(e));
    if (n_unknown) return set_error(HAST_ERR_TABLE_FULL, "hast_names_merge: the dictionary has no id left for %u of the other one's texts (its limit is %zu)", n_unknown, dst->limit);
    return HAST_OK;
}

void hast_names_destroy(hast_names *nm) {
    if (!nm) return;
    (void)hipSetDevice(nm->device);
    (void)hipStreamSynchronize(ctx_stream_of(nm->ctx));
    if (nm->d_tab) (void)hipFree(nm->d_tab);
    if (nm->d_n_ids) (void)hipFree(nm->d_n_ids);
    if (nm->d_text_of_id) (void)hipFree(nm->d_text_of_id);
    delete nm;
}

// Entries the CALLER knows (text record -> id), e.g. a table barcode -> class for a routing stream (hast_fq_set_route): copied to the
// device and filed by k_names_insert on the context's stream; returns when they are in.  Texts longer than 15 bytes have no record.
hast_status hast_names_insert(hast_names *nm, const uint8_t *text16, const uint32_t *ids, size_t n) {
    if (!nm || (n && (!text16 || !ids))) return set_error(HAST_ERR_INVALID, "null argument");
    if (nm->count + n > nm->limit) return set_error(HAST_ERR_INVALID, "hast_names_insert: %zu entries + %zu do not fit a table made for %zu", nm->count, n, nm->limit);
    FQ_TRY(hipSetDevice(nm->device));
    hipStream_t hs = ctx_stream_of(nm->ctx);
    const size_t kPiece = 1u << 20;
    NamePub *h = nullptr, *d = nullptr;
    FQ_TRY(pinned_malloc((void **)&h, std::min(n, kPiece) * sizeof(NamePub) + 64));
    if (dev_malloc((void **)&d, std::min(n, kPiece) * sizeof(NamePub) + 64) != hipSuccess) {
        (void)hipHostFree(h);
        return set_error(HAST_ERR_OOM, "hast_names_insert: device allocation failed");
    }
    hast_status st = HAST_OK;
    for (size_t at = 0; at < n && st == HAST_OK; at += kPiece) {
        const size_t m = std::min(kPiece, n - at);
        for (size_t i = 0; i < m; ++i) {
            memcpy(h[i].key, text16 + 16 * (at + i), 16);
            h[i].id = ids[at + i];
        }
        hipError_t e = hipMemcpyAsync(d, h, m * sizeof(NamePub), hipMemcpyHostToDevice, hs);
        if (e == hipSuccess) e = launch_names_insert(d, (uint32_t)m, nm->d_tab, nm->mask, hs);
        if (e == hipSuccess) e = hipStreamSynchronize(hs);
        if (e != hipSuccess) st = set_error(HAST_ERR_HIP, "hast_names_insert: %s", hipGetErrorString(e));
    }
    (void)hipHostFree(h);
    (void)hipFree(d);
    if (st == HAST_OK) nm->count += n;
    return st;
}

// buffers + events of one slot, on the current device
static hast_status alloc_slot(Slot &s, size_t buf, size_t max_rec, size_t pad, size_t block, bool striped, bool device_blocks = false) {
    // (a stream of device-side blocks pins no host copy of its blocks: ~0.7 ms per MB, 6 x 17 MB per stream -- the copy is made
    // if and when hast_fq_block_host_bytes asks for one)
    s.h_buf_bytes = buf;
    if (!device_blocks) FQ_TRY(pinned_malloc((void **)&s.h_buf, buf, hipHostMallocDefault));
    FQ_TRY(pinned_malloc((void **)&s.h_st, sizeof(FqState), hipHostMallocDefault));
    FQ_TRY(pinned_malloc((void **)&s.h_nids, 64, hipHostMallocDefault));
    s.h_nids[0] = 0;
    FQ_TRY(dev_malloc((void **)&s.d_buf, buf));
    FQ_TRY(dev_malloc((void **)&s.d_st, sizeof(FqState)));
    FQ_TRY(dev_malloc((void **)&s.d_tile, (buf / 4096 + 2) * sizeof(uint32_t)));
    FQ_TRY(dev_malloc((void **)&s.d_nl, (buf + 16) * sizeof(uint32_t)));
    FQ_TRY(dev_malloc((void **)&s.d_off, max_rec * sizeof(uint64_t)));
    for (uint32_t **p : {&s.d_len, &s.d_bcpos, &s.d_bclen, &s.d_ids}) FQ_TRY(dev_malloc((void **)p, max_rec * sizeof(uint32_t)));
    FQ_TRY(dev_malloc((void **)&s.d_votes, max_rec * 2 * sizeof(uint32_t)));
    FQ_TRY(dev_malloc((void **)&s.d_text, max_rec * 4 * sizeof(uint32_t)));
    // records the pinned per-record arrays hold (34 B each: page pinning is ~0.7 ms per MB, six slots a stream): a record of 100-bp
    // reads is ~240 bytes, of 150-bp reads ~340; a block of shorter ones takes the copy path and grows the arrays (hast_fq_next)
    size_t cap = block / 224 + 4096;
    if (const char *e = getenv("HAST_FQ_HOST_RECORDS")) cap = (size_t)std::max(1L, atol(e));           // (tests: force the copy path)
    if (hast_status st = grow_records(s, cap)) return st;
    for (hipEvent_t *e : {&s.named, &s.copied, &s.parsed, &s.done}) FQ_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    if (striped) {
        FQ_TRY(pinned_malloc((void **)&s.h_cnt, sizeof(FqState), hipHostMallocDefault));
        FQ_TRY(pinned_malloc((void **)&s.h_last, 16, hipHostMallocDefault));
        for (hipEvent_t *e : {&s.counted, &s.over_copied, &s.over_read}) FQ_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    (void)pad;
    return HAST_OK;
}

hast_status hast_fq_create(hast_ctx *ctx, size_t block_bytes, int n_buffers, hast_names *names, hast_fq **out) {
    return hast_fq_create_ex(ctx, block_bytes, n_buffers, names, 0, out);
}
hast_status hast_fq_create_ex(hast_ctx *ctx, size_t block_bytes, int n_buffers, hast_names *names, int device_blocks, hast_fq **out) {
    if (!ctx || !out) return set_error(HAST_ERR_INVALID, "null argument");
    *out = nullptr;
    // (contexts of ONE device may share a cache: an insert of one context's stream that is in flight while another's stream
    // probes can at worst hide a barcode from that probe -- the host then names the record, as for any new barcode)
    if (names && names->device != hast_ctx_device(ctx)) return set_error(HAST_ERR_INVALID, "the name cache belongs to another device");
    if (block_bytes < 4096 || block_bytes > (1ull << 30)) return set_error(HAST_ERR_INVALID, "block_bytes %zu out of [4 KB, 1 GB]", block_bytes);
    if (n_buffers < 2 || n_buffers > 16) return set_error(HAST_ERR_INVALID, "n_buffers %d out of [2,16]", n_buffers);
    FQ_TRY(hipSetDevice(hast_ctx_device(ctx)));
    hast_fq *f = new (std::nothrow) hast_fq();
    if (!f) return set_error(HAST_ERR_OOM, "host allocation failed");
    f->ctx = ctx;
    f->names = names;
    f->device = hast_ctx_device(ctx);
    f->k = hast_ctx_k(ctx);
    f->block = (block_bytes + 4095) & ~(size_t)4095;
    f->pad = 1u << 20;                                     // room for the unfinished record(s) carried from block to block: a record may be 1 MB
    const size_t buf = f->pad + f->block + 4096;
    f->max_rec = (f->pad + f->block) / 4 + 2;                                           // a record holds at least four newlines
    f->slots.resize((size_t)n_buffers);
    if (hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&f->parse_stream, hipStreamNonBlocking) != hipSuccess) {
        if (f->copy_stream) (void)hipStreamDestroy(f->copy_stream);
        delete f;
        return set_error(HAST_ERR_HIP, "hipStreamCreate failed");
    }
    if (device_blocks) f->source = 2;
    for (Slot &s : f->slots)
        if (hast_status st = alloc_slot(s, buf, f->max_rec, f->pad, f->block, false, device_blocks != 0)) {
            hast_fq_destroy(f);
            return st;
        }
    *out = f;
    return HAST_OK;
}

// A stream whose blocks go to n_ctx contexts in turn (block i -> context i % n_ctx): the blocks of ONE file on several GPUs, like
// the reference spreads the reads of one file over all its workers (classify.cpp:211-219).  A block is framed on its own GPU
// from the number of newlines in front of it in the file (fq_kernels.hip, "striped streams").
hast_status hast_fq_create_striped(hast_ctx *const *ctxs, int n_ctx, size_t block_bytes, int n_buffers_per_ctx, hast_names *const *names, hast_fq **out) {
    return hast_fq_create_striped_ex(ctxs, n_ctx, block_bytes, n_buffers_per_ctx, names, 0, out);
}
hast_status hast_fq_create_striped_ex(hast_ctx *const *ctxs, int n_ctx, size_t block_bytes, int n_buffers_per_ctx, hast_names *const *names, int device_blocks, hast_fq **out) {
    if (!ctxs || !out || n_ctx < 1) return set_error(HAST_ERR_INVALID, "null argument");
    *out = nullptr;
    if (block_bytes < 4096 || block_bytes > (1ull << 30)) return set_error(HAST_ERR_INVALID, "block_bytes %zu out of [4 KB, 1 GB]", block_bytes);
    // at least three buffers in all: a buffer is reused once the block BEHIND its old block has been framed (that framing reads
    // the old block's newline count and last byte), and a block is framed once the block behind IT has been submitted
    if (n_buffers_per_ctx < 2 || n_buffers_per_ctx * n_ctx < 3 || n_buffers_per_ctx * n_ctx > 64)
        return set_error(HAST_ERR_INVALID, "n_buffers_per_ctx %d out of range for %d contexts (3 to 64 buffers in all)", n_buffers_per_ctx, n_ctx);
    for (int i = 0; i < n_ctx; i++) {
        if (!ctxs[i] || hast_ctx_k(ctxs[i]) != hast_ctx_k(ctxs[0])) return set_error(HAST_ERR_INVALID, "contexts need one K");
        if (names && names[i] && names[i]->device != hast_ctx_device(ctxs[i])) return set_error(HAST_ERR_INVALID, "name cache %d belongs to another device", i);
    }
    hast_fq *f = new (std::nothrow) hast_fq();
    if (!f) return set_error(HAST_ERR_OOM, "host allocation failed");
    f->striped = true;
    f->ctx = ctxs[0];
    f->device = hast_ctx_device(ctxs[0]);
    f->k = hast_ctx_k(ctxs[0]);
    f->block = (block_bytes + 4095) & ~(size_t)4095;
    f->pad = 1u << 20;
    f->over_cap = std::min(f->pad, f->block);              // a record reaches into the next block only: at most this many bytes
    const size_t buf = f->pad + f->block + f->over_cap + 4096;
    f->max_rec = (f->block + f->over_cap) / 4 + 4;
    f->lanes.resize((size_t)n_ctx);
    f->records_per_lane.assign((size_t)n_ctx, 0);
    f->slots.resize((size_t)n_buffers_per_ctx * (size_t)n_ctx);
    hast_status st = HAST_OK;
    for (int i = 0; i < n_ctx && st == HAST_OK; i++) {
        FqLane &ln = f->lanes[(size_t)i];
        ln.ctx = ctxs[i];
        ln.names = names ? names[i] : nullptr;
        ln.device = hast_ctx_device(ctxs[i]);
        if (hipSetDevice(ln.device) != hipSuccess || hipStreamCreateWithFlags(&ln.copy_stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&ln.parse_stream, hipStreamNonBlocking) != hipSuccess)
            st = set_error(HAST_ERR_HIP, "hipStreamCreate failed on device %d", ln.device);
    }
    for (size_t si = 0; si < f->slots.size() && st == HAST_OK; si++) {
        Slot &s = f->slots[si];
        s.lane = (int)(si % (size_t)n_ctx);
        if (hipSetDevice(f->lanes[(size_t)s.lane].device) != hipSuccess) st = set_error(HAST_ERR_HIP, "hipSetDevice failed");
        else st = alloc_slot(s, buf, f->max_rec, f->pad, f->block, true, device_blocks != 0);
    }
    if (device_blocks) f->source = 2;
    if (st != HAST_OK) {
        hast_fq_destroy(f);
        return st;
    }
    *out = f;
    return HAST_OK;
}

void hast_fq_destroy(hast_fq *f) {
    if (!f) return;
    if (f->striped) {
        for (FqLane &ln : f->lanes) {
            (void)hipSetDevice(ln.device);
            if (ln.ctx) (void)hipStreamSynchronize(ctx_stream_of(ln.ctx));
            for (hipStream_t st : {ln.copy_stream, ln.parse_stream})
                if (st) {
                    (void)hipStreamSynchronize(st);
                    (void)hipStreamDestroy(st);
                }
        }
        for (Slot &s : f->slots) {
            (void)hipSetDevice(f->lanes[(size_t)s.lane].device);
            free_slot(s);
        }
        delete f;
        return;
    }
    (void)hipSetDevice(f->device);
    (void)hipStreamSynchronize(ctx_stream_of(f->ctx));
    for (hipStream_t st : {f->copy_stream, f->parse_stream})
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    for (Slot &s : f->slots) free_slot(s);
    delete f;
}

int hast_fq_lanes(const hast_fq *f) { return !f ? 0 : f->striped ? (int)f->lanes.size() : 1; }
uint64_t hast_fq_lane_records(const hast_fq *f, int lane) {
    if (!f || lane < 0) return 0;
    if (!f->striped) return 0;
    return (size_t)lane < f->records_per_lane.size() ? f->records_per_lane[(size_t)lane] : 0;
}

size_t hast_fq_block_bytes(const hast_fq *f) { return f ? f->block : 0; }

hast_status hast_fq_acquire(hast_fq *f, uint8_t **host_buf) {
    if (!f || !host_buf) return set_error(HAST_ERR_INVALID, "null argument");
    Slot &s = f->slots[f->n_acquired % f->slots.size()];
    if (s.state != Slot::FREE) return set_error(HAST_ERR_INVALID, "hast_fq_acquire: all %zu buffers are in use (commit the oldest block first)", f->slots.size());
    FQ_TRY(hipSetDevice(dev_of(f, s)));
    if (s.done_pending) {
        FQ_TRY(hipEventSynchronize(s.done));
        s.done_pending = false;
    }
    s.state = Slot::ACQUIRED;
    f->n_acquired++;
    *host_buf = s.h_buf ? s.h_buf + f->pad : nullptr;          // (NULL on a stream created for device-side blocks)
    return HAST_OK;
}

// The bytes of the next block are written ON THE DEVICE (e.g. by hast_gz_read_device): after hast_fq_acquire, *d_block is where
// they belong (room for hast_fq_block_bytes) and *fill_stream the stream the writes must be enqueued on -- the wait for the
// kernels that still read this buffer's previous block is already on it.  Then hast_fq_submit_device instead of hast_fq_submit.
hast_status hast_fq_device_block(hast_fq *f, uint8_t **d_block, hast_stream *fill_stream) {
    if (!f || !d_block || !fill_stream) return set_error(HAST_ERR_INVALID, "null argument");
    if (f->n_device_blocks >= f->n_acquired) return set_error(HAST_ERR_INVALID, "hast_fq_device_block without hast_fq_acquire");
    if (f->striped) {
        // block i of a striped stream lives on lane i % n: the address is on THAT GPU, the stream is that lane's copy stream.  What
        // still reads this buffer's previous block: its framing and classification (waited for by hast_fq_acquire) and the copy of
        // its first bytes into the view of the block in front of it, on that block's GPU (over_read).
        if (f->source == 1) return set_error(HAST_ERR_INVALID, "a stream takes host blocks or device blocks, not both");
        const size_t i = f->n_device_blocks, S = f->slots.size();
        if (i >= S && f->n_submitted + S <= i + 1)
            return set_error(HAST_ERR_INVALID, "hast_fq_device_block: %zu blocks are in hand and not submitted; a stream of %zu buffers allows %zu", i - f->n_submitted, S, S - 1);
        Slot &s = f->slots[i % S];
        FqLane &ln = f->lanes[(size_t)s.lane];
        FQ_TRY(hipSetDevice(ln.device));
        if (s.over_read_pending) {
            FQ_TRY(hipStreamWaitEvent(ln.copy_stream, s.over_read, 0));
            s.over_read_pending = false;
        }
        f->source = 2;
        f->n_device_blocks++;
        *d_block = s.d_buf + f->pad;
        *fill_stream = (hast_stream)ln.copy_stream;
        return HAST_OK;
    }
    if (f->source == 1) return set_error(HAST_ERR_INVALID, "a stream takes host blocks or device blocks, not both");
    f->source = 2;
    const size_t i = f->n_device_blocks;
    // This buffer held block i - n_buffers, whose unfinished tail the framing of block i - n_buffers + 1 reads: that framing must
    // have been LAUNCHED before anyone may be told to write here (the wait below is on its event).  A host block is uploaded by
    // hast_fq_submit, by when every earlier block has been submitted; a device block is filled by the caller, who could get here
    // with blocks in hand that it has not submitted yet -- seen: a record that straddles two blocks cut short, one run in two.
    if (i >= f->slots.size() && f->n_submitted + f->slots.size() <= i + 1)
        return set_error(HAST_ERR_INVALID, "hast_fq_device_block: %zu blocks are in hand and not submitted; a stream of %zu buffers allows %zu",
                         i - f->n_submitted, f->slots.size(), f->slots.size() - 1);
    const int si = (int)(i % f->slots.size());
    Slot &s = f->slots[(size_t)si];
    FQ_TRY(hipSetDevice(f->device));
    // (as in hast_fq_submit: the successor slot's framing reads this buffer's old tail)
    if (i >= f->slots.size()) FQ_TRY(hipStreamWaitEvent(f->copy_stream, f->slots[(size_t)(si + 1) % f->slots.size()].parsed, 0));
    f->n_device_blocks++;
    *d_block = s.d_buf + f->pad;
    *fill_stream = (hast_stream)f->copy_stream;
    return HAST_OK;
}

static hast_status submit_block(hast_fq *f, size_t n_bytes, int last, bool dev_src) {
    if (!f) return set_error(HAST_ERR_INVALID, "null argument");
    if (f->n_submitted >= f->n_acquired) return set_error(HAST_ERR_INVALID, "hast_fq_submit without hast_fq_acquire");
    if (n_bytes > f->block) return set_error(HAST_ERR_INVALID, "block of %zu bytes exceeds the capacity %zu", n_bytes, f->block);
    if (dev_src ? (f->source != 2 || f->n_device_blocks <= f->n_submitted) : (f->source == 2 || !f->slots[f->n_submitted % f->slots.size()].h_buf))
        return set_error(HAST_ERR_INVALID, dev_src ? "hast_fq_submit_device without hast_fq_device_block" : "a stream takes host blocks or device blocks, not both");
    if (!dev_src) f->source = 1;
    if (f->striped) return submit_striped(f, n_bytes, last, dev_src);
    const int si = (int)(f->n_submitted % f->slots.size());
    Slot &s = f->slots[(size_t)si];
    FQ_TRY(hipSetDevice(f->device));
    hipStream_t hs = f->parse_stream;
    s.n_bytes = n_bytes;
    s.last = last;
    s.dev_src = dev_src;
    s.host_view = !dev_src;
    // this slot's device buffer held block i - n_buffers, whose unfinished tail the NEXT block's k_fq_begin reads on the parse
    // stream: the upload that overwrites it waits for that kernel (the successor slot's `parsed` event, recorded behind it)
    if (!dev_src) {
        if (f->n_submitted >= f->slots.size()) FQ_TRY(hipStreamWaitEvent(f->copy_stream, f->slots[(size_t)(si + 1) % f->slots.size()].parsed, 0));
        if (n_bytes) FQ_TRY(hipMemcpyAsync(s.d_buf + f->pad, s.h_buf + f->pad, n_bytes, hipMemcpyHostToDevice, f->copy_stream));
    }
    FQ_TRY(hipEventRecord(s.copied, f->copy_stream));
    FQ_TRY(hipStreamWaitEvent(hs, s.copied, 0));
    const Slot *prev = f->prev_submitted >= 0 ? &f->slots[(size_t)f->prev_submitted] : nullptr;
    FQ_TRY(launch_fq_block(s.d_buf, s.d_st, prev ? prev->d_buf : nullptr, prev ? prev->d_st : nullptr, f->pad, n_bytes, s.d_tile, s.d_nl, s.d_off,
                           s.d_len, s.d_bcpos, s.d_bclen, s.h_bc, s.d_text, (uint32_t)s.h_cap, (uint32_t)f->k, last, hs));
    s.k_cap = s.h_cap;
    FQ_TRY(hipMemcpyAsync(s.h_st, s.d_st, sizeof(FqState), hipMemcpyDeviceToHost, hs));
    if (f->route) {
        if (hast_status st = enqueue_route(f, s, f->route_tab[0], false, last, f->pad + n_bytes, hs)) return st;
    } else if (hast_status st = enqueue_names_early(s, f->names, hs)) return st;
    FQ_TRY(hipEventRecord(s.parsed, hs));
    s.state = Slot::SUBMITTED;
    f->n_submitted++;
    f->prev_submitted = last ? -1 : si;                       // the next block starts a new file after `last`
    return HAST_OK;
}
hast_status hast_fq_submit(hast_fq *f, size_t n_bytes, int last) { return submit_block(f, n_bytes, last, false); }
hast_status hast_fq_submit_device(hast_fq *f, size_t n_bytes, int last) { return submit_block(f, n_bytes, last, true); }

// Host copy of the open block's bytes (the view bc_pos / bc_len point into): always there for host blocks; a device block is
// fetched on the first call (16 MB over PCIe: for the rare record whose barcode text the framer could not hand over in its 16-byte
// copy -- longer than 15 bytes -- or for a block with more records than the pinned arrays hold).
hast_status hast_fq_block_host_bytes(hast_fq *f, const uint8_t **bytes) {
    if (!f || !bytes) return set_error(HAST_ERR_INVALID, "null argument");
    *bytes = nullptr;
    if (f->n_opened == 0) return set_error(HAST_ERR_INVALID, "no open block");
    Slot &s = f->slots[(f->n_opened - 1) % f->slots.size()];
    if (s.state != Slot::OPEN) return set_error(HAST_ERR_INVALID, "no open block");
    if (!s.host_view) {
        FQ_TRY(hipSetDevice(dev_of(f, s)));
        if (!s.h_buf) FQ_TRY(pinned_malloc((void **)&s.h_buf, s.h_buf_bytes, hipHostMallocDefault));
        hipStream_t hs = ctx_stream_of(ctx_of(f, s));
        FQ_TRY(hipMemcpyAsync(s.h_buf, s.d_buf, f->pad + s.n_bytes + s.n_over, hipMemcpyDeviceToHost, hs));
        FQ_TRY(hipStreamSynchronize(hs));
        s.host_view = true;
    }
    *bytes = s.h_buf;
    return HAST_OK;
}

int hast_fq_poll(hast_fq *f) {
    if (!f || f->n_opened >= f->n_submitted) return 0;
    Slot &s = f->slots[f->n_opened % f->slots.size()];
    if (s.state != Slot::SUBMITTED) return 0;
    if (f->striped) {
        if (advance_striped(f, false, f->n_submitted) != HAST_OK) return 1;      // failed for good: hast_fq_next reports it
        if (f->n_framed <= f->n_opened) return 0;
    }
    (void)hipSetDevice(dev_of(f, s));
    return hipEventQuery(s.parsed) == hipSuccess ? 1 : 0;
}

hast_status hast_fq_next(hast_fq *f, hast_fq_block *out) {
    if (!f || !out) return set_error(HAST_ERR_INVALID, "null argument");
    memset(out, 0, sizeof(*out));
    if (f->route) return set_error(HAST_ERR_INVALID, "hast_fq_next: the stream routes (hast_fq_next_routed)");
    if (f->n_opened >= f->n_submitted) return set_error(HAST_ERR_INVALID, "hast_fq_next: no submitted block");
    Slot &s = f->slots[f->n_opened % f->slots.size()];
    if (s.state != Slot::SUBMITTED || (f->n_opened && f->slots[(f->n_opened - 1) % f->slots.size()].state == Slot::OPEN))
        return set_error(HAST_ERR_INVALID, "hast_fq_next: commit the previous block first");
    if (f->striped) {
        if (hast_status a = advance_striped(f, true, f->n_opened + 1)) return a;
        if (f->n_framed <= f->n_opened)
            return set_error(HAST_ERR_INVALID, "hast_fq_next: a block of a striped stream is framed once the block behind it has been submitted (or it ends the file)");
    }
    hast_ctx *const sctx = ctx_of(f, s);
    hast_names *const snames = names_of(f, s);
    FQ_TRY(hipSetDevice(dev_of(f, s)));
    hipStream_t hs = ctx_stream_of(sctx);
    FQ_TRY(hipEventSynchronize(s.parsed));
    const FqState st = *s.h_st;
    if (st.flags & 2) return set_error(HAST_ERR_FORMAT, "a FASTQ record is larger than %zu bytes", f->striped ? f->over_cap : f->pad);
    if (st.n_rec > f->max_rec) return set_error(HAST_ERR_INVALID, "record table overflow");
    const size_t n = st.n_rec;
    const bool by_copy = n > s.k_cap;          // more (short) records than the kernel could write to the host itself
    if (by_copy) {
        // every per-record array grows together (the slot's next block hands the new h_cap to the records kernel)
        if (hast_status g = grow_records(s, n + n / 4 + 1024)) return g;
        FQ_TRY(hipMemcpyAsync(s.h_bc, s.d_bcpos, n * sizeof(uint32_t), hipMemcpyDeviceToHost, hs));
        FQ_TRY(hipMemcpyAsync(s.h_bc + s.h_cap, s.d_bclen, n * sizeof(uint32_t), hipMemcpyDeviceToHost, hs));
        FQ_TRY(hipEventRecord(s.parsed, hs));
    }
    // ids from the device-side name cache (after a few blocks nearly every barcode of a block has been seen before)
    s.host_texts = !by_copy;
    s.use_cache = snames && n;                 // (also a block that outgrew the pinned arrays: they have just been regrown, and the text records
                                               // of all its records are on the device -- every record goes through ONE dictionary)
    if (s.use_cache && s.named_early && !by_copy) {
        // (named behind its framing: `named` lies in front of `parsed` on that stream, the ids are there)
    } else if (s.use_cache) {
        s.h_unknown[0] = 0;
        if (snames->dict) {
            FQ_TRY(launch_fq_name_claim(s.d_text, (uint32_t)n, snames->d_tab, snames->mask, snames->d_n_ids, (uint32_t)snames->limit, snames->d_text_of_id, s.h_ids,
                                        s.h_unknown, s.d_ids, hs));
            FQ_TRY(hipMemcpyAsync(s.h_nids, snames->d_n_ids, sizeof(uint32_t), hipMemcpyDeviceToHost, hs));
        } else FQ_TRY(launch_fq_name(s.d_text, (uint32_t)n, snames->d_tab, snames->mask, s.h_ids, s.h_unknown, hs));
        FQ_TRY(hipEventRecord(s.named, hs));
    }
    // the reads are classified where they lie in the raw block WHILE the host names the barcodes
    if (n && !(st.flags & 1))
        if (hast_status c = classify_framed(sctx, s.d_buf, f->pad + s.n_bytes + s.n_over, s.d_off, s.d_len, st.max_len, s.d_votes, n, hs)) return c;
    if (by_copy) FQ_TRY(hipEventSynchronize(s.parsed));        // (the event sits in front of the kernels)
    if (s.use_cache) FQ_TRY(hipEventSynchronize(s.named));
    if (f->striped) {
        // (the host view already holds the first bytes of the next block behind this block's own: submit_striped put them there)
        f->records_per_lane[(size_t)s.lane] += n;
    } else if (!s.dev_src) {
    // host view of the bytes the barcode extents point into: this block's bytes, preceded by the previous block's tail
    if (st.tail_in != f->carry.size()) return set_error(HAST_ERR_INVALID, "tail bookkeeping out of step (%llu vs %zu)", (unsigned long long)st.tail_in, f->carry.size());
    if (!f->carry.empty()) memcpy(s.h_buf + f->pad - f->carry.size(), f->carry.data(), f->carry.size());
    const size_t tail = s.last ? 0 : (size_t)(st.parse_hi - st.tail_lo);
    f->carry.assign(s.h_buf + st.parse_hi - tail, s.h_buf + st.parse_hi);
    }
    out->n_records = n;
    out->n_bases = st.bases;
    out->max_read_len = st.max_len;
    out->short_read = (st.flags & 1) ? 1 : 0;
    out->bytes = s.dev_src ? nullptr : s.h_buf;              // (device blocks: hast_fq_block_host_bytes fetches them when asked)
    out->bc_pos = s.h_bc;
    out->bc_len = s.h_bc + s.h_cap;
    out->bc_text = by_copy ? nullptr : reinterpret_cast<const uint8_t *>(s.h_bc + 2 * s.h_cap);
    out->ids = s.h_ids;
    out->unknown = s.use_cache ? s.h_unknown + 1 : nullptr;
    out->n_unknown = s.use_cache ? s.h_unknown[0] : n;
    out->dict_ids = (s.use_cache && snames->dict) ? std::min<uint64_t>(s.h_nids[0], snames->limit) : 0;
    s.state = Slot::OPEN;
    f->n_opened++;
    return HAST_OK;
}

// From here on the stream ROUTES: its blocks are framed as before but not classified; behind the framing the records of a block are
// sorted into four runs by the class of their barcode (fq_kernels.hip "routing": steps 10-11 of the wrapper, quartering_fastq.awk) and
// the runs are copied to pinned host memory; hast_fq_next_routed hands them out.  tables: barcode text -> class (1 paternal, 2 maternal,
// 3 homozygous), one per lane of a striped stream (contexts of one GPU may share one), made with hast_names_create / hast_names_insert.
// Only between files: no block may be acquired and not committed.  tables == NULL: back to classifying.
hast_status hast_fq_set_route(hast_fq *f, hast_names *const *tables, int n_tables) {
    if (!f) return set_error(HAST_ERR_INVALID, "null argument");
    for (const Slot &s : f->slots)
        if (s.state != Slot::FREE) return set_error(HAST_ERR_INVALID, "hast_fq_set_route: a block is in hand (commit it first)");
    if (!tables) {
        f->route = false;
        f->route_tab.clear();
        return HAST_OK;
    }
    const int lanes = f->striped ? (int)f->lanes.size() : 1;
    if (n_tables != lanes) return set_error(HAST_ERR_INVALID, "hast_fq_set_route: %d tables for a stream of %d lanes", n_tables, lanes);
    for (int i = 0; i < lanes; ++i)
        if (!tables[i] || tables[i]->device != (f->striped ? f->lanes[(size_t)i].device : f->device))
            return set_error(HAST_ERR_INVALID, "hast_fq_set_route: table %d is missing or lives on another device", i);
    for (Slot &s : f->slots) {
        if (s.d_out) continue;
        FQ_TRY(hipSetDevice(dev_of(f, s)));
        s.route_rec = f->max_rec;
        FQ_TRY(dev_malloc(&s.d_rstart, s.route_rec * sizeof(uint32_t)));
        FQ_TRY(dev_malloc(&s.d_rlen, s.route_rec * sizeof(uint32_t)));
        FQ_TRY(dev_malloc(&s.d_rcls, s.route_rec));
        FQ_TRY(dev_malloc(&s.d_rtile, (s.route_rec / kRouteTile + 2) * 8 * sizeof(uint32_t)));
        FQ_TRY(dev_malloc(&s.d_rs, sizeof(RouteState)));
        FQ_TRY(dev_malloc(&s.d_out, s.h_buf_bytes));
        FQ_TRY(pinned_malloc(&s.h_out, s.h_buf_bytes));
        FQ_TRY(pinned_malloc(&s.h_rs, sizeof(RouteState)));
    }
    f->route_tab.assign(tables, tables + lanes);
    f->route = true;
    // (the next block starts a file: a routing pass reads its inputs from their first byte)
    f->prev_submitted = -1;
    f->carry.clear();
    return HAST_OK;
}

hast_status hast_fq_next_routed(hast_fq *f, hast_fq_routed *out) {
    if (!f || !out) return set_error(HAST_ERR_INVALID, "null argument");
    memset(out, 0, sizeof(*out));
    if (!f->route) return set_error(HAST_ERR_INVALID, "hast_fq_next_routed: the stream does not route (hast_fq_set_route)");
    if (f->n_opened >= f->n_submitted) return set_error(HAST_ERR_INVALID, "hast_fq_next_routed: no submitted block");
    Slot &s = f->slots[f->n_opened % f->slots.size()];
    if (s.state != Slot::SUBMITTED || (f->n_opened && f->slots[(f->n_opened - 1) % f->slots.size()].state == Slot::OPEN))
        return set_error(HAST_ERR_INVALID, "hast_fq_next_routed: commit the previous block first");
    if (f->striped) {
        if (hast_status a = advance_striped(f, true, f->n_opened + 1)) return a;
        if (f->n_framed <= f->n_opened)
            return set_error(HAST_ERR_INVALID, "hast_fq_next_routed: a block of a striped stream is framed once the block behind it has been submitted (or it ends the file)");
    }
    FQ_TRY(hipSetDevice(dev_of(f, s)));
    FQ_TRY(hipEventSynchronize(s.parsed));
    const FqState st = *s.h_st;
    const RouteState rs = *s.h_rs;
    if ((st.flags & 2) || (rs.flags & 2)) return set_error(HAST_ERR_FORMAT, "a FASTQ record is larger than %zu bytes", f->striped ? f->over_cap : f->pad);
    hipStream_t hs = ctx_stream_of(ctx_of(f, s));
    uint64_t at = 0;
    for (int c = 0; c < 4; ++c) {
        out->count[c] = rs.count[c];
        out->run[c] = s.h_out + at;
        out->run_bytes[c] = rs.bytes[c];
        out->n_records += rs.count[c];
        at += rs.bytes[c];
    }
    if (at > s.h_buf_bytes) return set_error(HAST_ERR_INVALID, "routed runs overflow their buffer");
    if (rs.flags & 1) {
        // a record the device could not route: the caller gets the view and every record's extent + class and routes the block itself
        if (!s.h_buf) FQ_TRY(pinned_malloc(&s.h_buf, s.h_buf_bytes));
        const size_t view = std::min<size_t>(st.parse_hi, s.h_buf_bytes);
        s.v_rstart.resize(rs.n_cand);
        s.v_rlen.resize(rs.n_cand);
        s.v_rcls.resize(rs.n_cand);
        FQ_TRY(hipMemcpyAsync(s.h_buf, s.d_buf, view, hipMemcpyDeviceToHost, hs));
        if (rs.n_cand) {
            FQ_TRY(hipMemcpyAsync(s.v_rstart.data(), s.d_rstart, rs.n_cand * sizeof(uint32_t), hipMemcpyDeviceToHost, hs));
            FQ_TRY(hipMemcpyAsync(s.v_rlen.data(), s.d_rlen, rs.n_cand * sizeof(uint32_t), hipMemcpyDeviceToHost, hs));
            FQ_TRY(hipMemcpyAsync(s.v_rcls.data(), s.d_rcls, rs.n_cand, hipMemcpyDeviceToHost, hs));
        }
        FQ_TRY(hipStreamSynchronize(hs));
        s.host_view = true;
        out->host_block = 1;
        out->bytes = s.h_buf;
        out->rec_start = s.v_rstart.data();
        out->rec_len = s.v_rlen.data();
        out->rec_class = s.v_rcls.data();
        out->n_slots = rs.n_cand;
    }
    if (rs.tail_hi > rs.tail_lo) {                         // the partial record at the end of the file: the caller's, by awk's rules
        s.v_tail.resize(rs.tail_hi - rs.tail_lo);
        FQ_TRY(hipMemcpyAsync(s.v_tail.data(), s.d_buf + rs.tail_lo, s.v_tail.size(), hipMemcpyDeviceToHost, hs));
        FQ_TRY(hipStreamSynchronize(hs));
        out->tail = s.v_tail.data();
        out->tail_bytes = s.v_tail.size();
    }
    if (f->striped) f->records_per_lane[(size_t)s.lane] += out->n_records;
    s.state = Slot::OPEN;
    f->n_opened++;
    return HAST_OK;
}

hast_status hast_fq_commit(hast_fq *f) {
    if (!f) return set_error(HAST_ERR_INVALID, "null argument");
    if (f->n_opened == 0) return set_error(HAST_ERR_INVALID, "hast_fq_commit without hast_fq_next");
    Slot &s = f->slots[(f->n_opened - 1) % f->slots.size()];
    if (s.state != Slot::OPEN) return set_error(HAST_ERR_INVALID, "hast_fq_commit: no open block");
    hast_ctx *const sctx = ctx_of(f, s);
    FQ_TRY(hipSetDevice(dev_of(f, s)));
    hipStream_t hs = ctx_stream_of(sctx);
    const size_t n = s.h_st->n_rec;
    if (f->route) {                                        // a routed block: the caller is through with its runs, nothing to book
        s.done_pending = false;
        s.state = Slot::FREE;
        return HAST_OK;
    }
    if (n && !(s.h_st->flags & 1)) {
        const size_t nbc = ctx_n_barcodes(sctx);
        if (s.use_cache) {
            // the caller named the records the cache did not know: check those, teach the cache, and let the bookkeeping kernel
            // read the ids where they are (pinned host memory: a copy would queue up behind the next block's upload)
            const uint32_t nu = s.h_unknown[0];
            hast_names *nm = names_of(f, s);
            uint32_t np = 0;
            for (uint32_t j = 0; j < nu; ++j) {
                const uint32_t i = s.h_unknown[1 + j];
                if (s.h_ids[i] >= nbc) return set_error(HAST_ERR_INVALID, "barcode id %u of record %u is outside the %zu counters", s.h_ids[i], i, nbc);
                const uint32_t *t = s.h_bc + 2 * s.h_cap + 4 * (size_t)i;
                if (nm->dict || !s.host_texts) continue;                                      // (a dictionary names by itself: what it left is the host's for good)
                if ((t[0] & 0xFFu) == 0xFFu || nm->count + np >= nm->limit) continue;          // long barcodes stay with the host
                NamePub &p = s.h_pubs[np++];
                p.key[0] = t[0]; p.key[1] = t[1]; p.key[2] = t[2]; p.key[3] = t[3];
                p.id = s.h_ids[i];
            }
            if (np) {
                FQ_TRY(launch_names_insert(s.h_pubs, np, nm->d_tab, nm->mask, hs));
                nm->count += np;                               // (an upper bound: a barcode met twice in one block is counted twice)
            }
            // (a block the dictionary named completely: its ids lie in device memory as well -- the bookkeeping kernel reads them there)
            static const bool ids_from_host = [] { const char *e = getenv("HAST_COMMIT_IDS"); return e && !strcmp(e, "host"); }();
            const bool from_device = nm && nm->dict && nu == 0 && !ids_from_host;
            if (hast_status c = commit_framed(sctx, s.d_votes, from_device ? s.d_ids : s.h_ids, n, hs)) return c;
        } else {
            for (size_t i = 0; i < n; ++i)
                if (s.h_ids[i] >= nbc) return set_error(HAST_ERR_INVALID, "barcode id %u of record %zu is outside the %zu counters", s.h_ids[i], i, nbc);
            FQ_TRY(hipMemcpyAsync(s.d_ids, s.h_ids, n * sizeof(uint32_t), hipMemcpyHostToDevice, hs));
            if (hast_status c = commit_framed(sctx, s.d_votes, s.d_ids, n, hs)) return c;
        }
    }
    FQ_TRY(hipEventRecord(s.done, hs));
    s.done_pending = true;
    s.state = Slot::FREE;
    return HAST_OK;
}

}  // extern "C"

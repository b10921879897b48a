// gz_api.cpp -- hast_gz_*: one ordinary .gz file inflated ON THE GPU (include/hast.h, "gzip input decoded on the device").
// Host side of gz_kernels.hip: an uploader thread moves the COMPRESSED bytes to HBM (pread by several threads into pinned
// pieces), a producer thread runs the passes segment by segment -- search + decode of every 32-KB chunk, the chain of accepted
// chunks with its follow-up jobs (gz_chain.h), windows, CRC-32 -- and checks every member's CRC-32 / ISIZE; the caller's thread
// (hast_gz_read_device) only launches the translate kernel that writes the bytes it asks for where it wants them.
// Replaces, for this path, gzstream.h:47 + classify.cpp:245-254 (one zlib stream per file on one host thread).
// Several GPUs (hast_gz_open_multi; the reference deals the reads of ONE file to all its workers, classify.cpp:211-219): the PASSES
// of the one deflate stream go to the GPUs in turn -- a chunk's marker-symbol decode needs nothing from its neighbours -- every GPU
// ("unit") with its own copy of the compressed bytes, job arrays, symbol arenas and streams; the chain of accepted chunks stays one
// (host), the 32 KB a pass hands to the next travel through pinned host memory, and the reader's bytes are translated on the unit
// that decoded them and, when the reader's buffer lives on another GPU, copied there peer to peer.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hast.h"
#include "gz_chain.h"
#include "gz_core.h"
#include "gz_device.h"
#include "hast_internal.h"
#include "worker_pool.h"

using namespace hast;
using namespace hast::gz;

namespace {

constexpr size_t kPiece = 16u << 20;         // upload granule (HAST_GZ_PIECE_BYTES: tests)
constexpr size_t kInPad = 256;               // zero bytes behind the file's last byte on the device

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct DevBuf {                               // a device allocation that only ever grows
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        park_device(p, bytes, 2);                              // (not hipFree: it waits for every stream of the device, hast_internal.h)
        p = nullptr;
        bytes = 0;
        hipError_t e = dev_malloc(&p, need);
        if (e == hipErrorOutOfMemory) {                        // (what closed streams left parked may be what is missing)
            (void)hipGetLastError();
            release_parked();
            e = dev_malloc(&p, need);
        }
        if (e == hipSuccess) bytes = need;
        else p = nullptr;
        return e;
    }
    void release() {
        park_device(p, bytes, 1);
        p = nullptr;
        bytes = 0;
    }
};

struct Batch {                                // accepted chunks the consumer may translate
    int unit = 0, arena = -1;
    std::vector<Accepted> acc;
    uint64_t out_lo = 0, out_hi = 0;
    size_t next = 0;                          // consumer: first chunk not fully delivered
    bool eof = false;
    std::string error;
};

struct Arena {
    DevBuf syms, windows, acc, need, crc, carry, cursor;   // (need: scratch of the windows pass; cursor: slots of `syms` taken by the pass in flight)
    std::vector<DevBuf> gap;                  // follow-up jobs' symbols (reused from batch to batch)
    size_t gap_used = 0;
    bool busy = false;                        // a published batch lives in it
    hipEvent_t done = nullptr;                // recorded behind the consumer's last translate of that batch
    bool done_recorded = false;
};

// What ONE GPU holds of a stream.  The nominal pass of a unit's segment j + 1 (search + decode of its chunks: where the time goes) is
// launched BEFORE its segment j's results are walked, so that the GPU decodes while the host chains, the single-wave follow-up jobs
// run and the consumer translates; since the end of round 6 segment j + 1's pass is launched together with segment j's, on a second stream
// (hast_gz::ahead): job arrays in ahead + 2 copies, follow-up jobs in one more, ahead + 2 symbol arenas taking turns (one pass of a unit at
// a time, HAST_GZ_AHEAD=0: two), the work behind a nominal pass on a stream of its own, the reader's translate kernels (and the copies
// towards another GPU) on yet another.
struct Unit {
    int device = 0;
    uint32_t *d_in = nullptr;                 // the compressed file + zero padding
    static constexpr int kJobCopies = 4;      // (hast_gz::ahead + 1 passes in flight + the one being walked)
    DevBuf jobs[kJobCopies], fjobs, bounce;   // (bounce: where the reader's bytes are translated to when its buffer is on another GPU)
    ChunkJob *h_jobs[kJobCopies] = {}, *h_fjobs = nullptr;   // pinned
    hipEvent_t nom_done[kJobCopies] = {};
    uint32_t *h_crc = nullptr;                // pinned
    size_t h_crc_cap = 0;
    static constexpr int kArenas = 4;
    Arena arena[kArenas];
    static constexpr int kDecStreams = 3;     // (a unit's passes that run side by side: one stream each)
    hipStream_t up_stream = nullptr, dec_stream[kDecStreams] = {}, post_stream = nullptr, xl_stream = nullptr;
    hipEvent_t xl_done = nullptr;             // behind the reader's last launches on xl_stream
    int dec_masked_free[kDecStreams] = {};    // != 0: that dec_stream is a CU-masked stream out of the process's pool (goes back there)
};

}  // namespace

struct hast_gz {
    hast_ctx *ctx = nullptr;
    int fd = -1;
    std::string path;
    uint64_t file_size = 0;
    size_t chunk_bytes = 32768, seg_chunks = 4096;
    double room = 12.0;
    double slot_fraction = 0.70;              // slots of a pass's arena per chunk of the pass (HAST_GZ_SLOT_FRACTION): 4 chunks in 10 of a FASTQ hold no
                                              // block start and need none; a chunk that finds the pool empty is decoded by a follow-up job
    bool slot_adaptive = true;                // ... and the share follows what the passes walked so far have found (upwards only: a stream of small
                                              // blocks has a start in every chunk); a file of one pass takes a slot per chunk
    size_t pool_slots(size_t chunks) const { return std::min(chunks, std::max<size_t>(4, (size_t)((double)chunks * slot_fraction + 0.999))); }
    uint32_t slot_syms = 0;
    size_t h_jobs_cap = 0;
    std::vector<std::unique_ptr<Unit>> units; // segment k -> units[k % n]
    bool split_same_device = false;           // test switch: one unit per context even where contexts share a GPU, bounce path forced
    uint8_t *h_carry = nullptr;               // pinned: the 32 KB behind the last batch (in front of the next one, whichever unit has it)
    std::vector<std::pair<int, hipEvent_t>> in_events;   // reader: (device, event recorded on the caller's stream in front of a translate)
    // threads
    std::thread uploader, producer;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t uploaded = 0;                    // bytes of the file on EVERY unit
    // THE RING.  A file larger than `ring` bytes (default: 2 GB, or what the passes in flight need) is not kept whole on the device:
    // byte x of the file lies at x % ring, and the first `piece` bytes of every lap also behind the ring's end, so that what a job reads
    // -- from its start to a block boundary at most one piece behind the lap's end -- is contiguous (ChunkJob.in_adj_words /
    // limit_bits say where).  The uploader waits before it overwrites what may still be read: bytes behind `released` = the chain's
    // end, and no further than the first chunk of the oldest pass in flight.  ring == 0: the whole file (every file of this
    // repo's benchmarks; HAST_GZ_RING_BYTES forces a ring, tests)
    size_t piece = kPiece;
    uint64_t ring = 0, released = 0;
    std::string up_error;
    bool stop = false;
    std::deque<std::unique_ptr<Batch>> ready; // producer -> consumer
    // consumer
    std::unique_ptr<Batch> cur;
    uint64_t cursor = 0;                      // bytes delivered
    bool done = false;
    std::string err;
    // chain + member check (producer)
    Chain chain;
    uint32_t crc_acc = 0;
    uint64_t isize_acc = 0;
    bool crc_started = false;
    // stats
    hast_gz_stats st{};
    Unit &unit_of(size_t k) { return *units[k % units.size()]; }
    int n_arenas = 3;                         // symbol arenas a unit takes turns with (3 with `ahead`, else 2; HAST_GZ_ARENAS)
    Arena &arena_of(size_t k) { return unit_of(k).arena[(k / units.size()) % (size_t)n_arenas]; }
    int jobs_of(size_t k) const { return (int)((k / units.size()) % (size_t)(ahead + 2)); }
    // ahead = 1: a unit's passes j and j + 1 are on the GPU TOGETHER (two streams, three arenas, three job arrays): a pass lasts as long as
    // its slowest wave -- a whole deflate block -- and its last third runs a thinning set of waves; the next pass's
    // waves take the slots they leave.  HAST_GZ_AHEAD=0: one pass of a unit at a time, as until round 6; =2: three side by side (four arenas:
    // ~4 % more for 2.8 GB more, profiles/round6_gz_ahead_sweep.txt)
    int ahead = 1;
    hipStream_t dec_stream_of(size_t k) {
        Unit &U = unit_of(k);
        hipStream_t s = U.dec_stream[(k / units.size()) % (size_t)(ahead + 1)];
        return s ? s : U.dec_stream[0];
    }
    void job_view(ChunkJob &j) const {         // where job j finds the file's words (ring: the lap its first bit lies in)
        j.in_adj_words = 0;
        j.limit_bits = 0;
        if (!ring) return;
        const uint64_t lap = (j.from_bit >> 3) / ring;
        j.in_adj_words = lap * (ring / 4);
        j.limit_bits = ((lap + 1) * ring + piece) * 8;
    }
};

namespace {

#define GZ_HIP(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return std::string(#expr ": ") + hipGetErrorString(e_);          \
    } while (0)

// HAST_GZ_TRACE=1: where a segment's time goes on the producer's thread (allocations, waits), one line per step that took > 1 ms
bool gz_trace() {
    static const bool on = getenv("HAST_GZ_TRACE") != nullptr;
    return on;
}
struct TraceStep {
    const hast_gz *g;
    size_t k;
    const char *what;
    double t0;
    TraceStep(const hast_gz *g_, size_t k_, const char *w) : g(g_), k(k_), what(w), t0(gz_trace() ? now_s() : 0) {}
    ~TraceStep();
};

TraceStep::~TraceStep() {
    if (!gz_trace()) return;
    const double dt = now_s() - t0;
    if (dt > 1e-3) fprintf(stderr, "gz seg %zu of %s: %s %.3f s\n", k, g->path.c_str(), what, dt);
}

// CU-masked streams are created once per (device, free CUs) and handed from stream to stream of compressed input, never destroyed:
// creating one right after another had been destroyed hung inside the runtime every other time (a stream closed early with passes
// in flight, then the next file opened: tests/test_gz_gpu.py test_a_stream_closed_early_with_passes_in_flight).
struct MaskedStreams {
    std::mutex mu;
    std::vector<std::pair<std::pair<int, int>, hipStream_t>> idle;      // ((device, free CUs), stream)
};
MaskedStreams &masked_streams() {
    static MaskedStreams *p = new MaskedStreams();                       // (never destroyed: the process ends with it)
    return *p;
}
hipStream_t masked_stream_get(int device, int n_cu, int free_cus) {
    MaskedStreams &ms = masked_streams();
    {
        std::lock_guard<std::mutex> lk(ms.mu);
        for (size_t i = 0; i < ms.idle.size(); ++i)
            if (ms.idle[i].first == std::make_pair(device, free_cus)) {
                hipStream_t s = ms.idle[i].second;
                ms.idle.erase(ms.idle.begin() + (long)i);
                return s;
            }
    }
    std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0u);
    for (int i = 0; i < n_cu - free_cus; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return s;
}
void masked_stream_put(int device, int free_cus, hipStream_t s) {
    MaskedStreams &ms = masked_streams();
    std::lock_guard<std::mutex> lk(ms.mu);
    ms.idle.push_back({{device, free_cus}, s});
}

bool read_at(int fd, uint64_t off, size_t n, uint8_t *dst, size_t *got) {
    size_t have = 0;
    while (have < n) {
        const ssize_t r = pread(fd, dst + have, n - have, (off_t)(off + have));
        if (r < 0 && errno == EINTR) continue;
        if (r < 0) return false;
        if (r == 0) break;
        have += (size_t)r;
    }
    *got = have;
    return true;
}

// compressed bytes -> HBM of every unit, piece by piece; `uploaded` moves on when a piece IS there (on all of them: every GPU has
// its own PCIe link, and a unit may be handed a follow-up job anywhere in front of its own segments)
void upload_loop(hast_gz *g) {
    const size_t nu = g->units.size();
    (void)hipSetDevice(g->units[0]->device);
    uint8_t *h[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev[2];
    std::string bad;
    for (int i = 0; i < 2 && bad.empty(); ++i) {
        if (pinned_malloc((void **)&h[i], g->piece + kInPad, hipHostMallocPortable) != hipSuccess) bad = "gz: pinned staging allocation failed";
        ev[i].assign(nu, nullptr);
        for (size_t u = 0; u < nu && bad.empty(); ++u)
            if (hipSetDevice(g->units[u]->device) != hipSuccess || hipEventCreateWithFlags(&ev[i][u], hipEventDisableTiming) != hipSuccess) bad = "gz: pinned staging allocation failed";
    }
    const int nthr = 8;
    WorkerPool pool(nthr);
    uint64_t end_of[2] = {0, 0};
    bool in_flight[2] = {false, false};
    auto land = [&](int b) {                                       // piece in buffer b is on the devices
        if (!in_flight[b]) return;
        for (size_t u = 0; u < nu; ++u)
            if ((hipSetDevice(g->units[u]->device) != hipSuccess || hipEventSynchronize(ev[b][u]) != hipSuccess) && bad.empty()) bad = "gz: upload failed";
        in_flight[b] = false;
        std::lock_guard<std::mutex> lk(g->mu);
        g->uploaded = std::max(g->uploaded, end_of[b]);
        g->cv.notify_all();
    };
    size_t i = 0;
    const size_t P = g->piece;
    for (uint64_t off = 0; off < g->file_size && bad.empty(); off += P, ++i) {
        const size_t n = (size_t)std::min<uint64_t>(P, g->file_size - off);
        // (a ring: the bytes this piece overwrites must be behind everything that may still be read.  What is in flight is booked
        // first: the producer may be waiting for exactly those bytes in order to move `released` on)
        bool must_wait = false;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            must_wait = g->ring && !g->stop && !(off + n <= g->released + g->ring);
        }
        if (must_wait) {
            land(0);
            land(1);
        }
        {
            std::unique_lock<std::mutex> lk(g->mu);
            if (must_wait) {
                g->st.upload_waited_for_ring++;
                g->cv.wait(lk, [&] { return g->stop || off + n <= g->released + g->ring; });
            }
            if (g->stop) break;
        }
        const int b = (int)(i & 1);
        land(b);
        const size_t share = ((n / nthr) + 4095) & ~(size_t)4095;
        std::atomic<bool> short_read{false};
        pool.run([&](int t) {
            const size_t from = std::min(n, share * (size_t)t), to = std::min(n, from + share);
            size_t got = 0;
            if (to > from && (!read_at(g->fd, off + from, to - from, h[b] + from, &got) || got != to - from)) short_read = true;
        });
        if (short_read) { bad = "gz: read failed (the file is shorter than its size says, or an I/O error)"; break; }
        // (a ring has no zeros behind the file's last byte waiting: they travel with the last piece)
        const bool last_piece = off + n >= g->file_size;
        size_t n_up = n;
        if (g->ring && last_piece) {
            memset(h[b] + n, 0, kInPad);
            n_up = n + kInPad;
        }
        const uint64_t pos = g->ring ? off % g->ring : off;
        if (g->ring && off && pos < P) {                          // (this piece starts a new lap)
            std::lock_guard<std::mutex> lk(g->mu);
            g->st.ring_laps++;
        }
        for (size_t u = 0; u < nu && bad.empty(); ++u) {
            Unit &U = *g->units[u];
            uint8_t *const d = reinterpret_cast<uint8_t *>(U.d_in);
            if (hipSetDevice(U.device) != hipSuccess || hipMemcpyAsync(d + pos, h[b], n_up, hipMemcpyHostToDevice, U.up_stream) != hipSuccess) bad = "gz: upload failed";
            // the first piece of a lap also behind the ring's end: the jobs of the lap in front read on into it
            if (bad.empty() && g->ring && pos == 0 && off && hipMemcpyAsync(d + g->ring, h[b], n_up, hipMemcpyHostToDevice, U.up_stream) != hipSuccess) bad = "gz: upload failed";
            if (bad.empty() && hipEventRecord(ev[b][u], U.up_stream) != hipSuccess) bad = "gz: upload failed";
        }
        if (!bad.empty()) break;
        end_of[b] = off + n;
        in_flight[b] = true;
        land(b ^ 1);                                               // (the piece before this one: usually there by now)
    }
    land(0);
    land(1);
    for (int k = 0; k < 2; ++k) {
        park_pinned(h[k], g->piece + kInPad, 0);                              // (the other streams are in mid-file: hast_internal.h)
        for (hipEvent_t e : ev[k])
            if (e) (void)hipEventDestroy(e);
    }
    std::lock_guard<std::mutex> lk(g->mu);
    if (!bad.empty()) g->up_error = bad;
    g->cv.notify_all();
}

void publish(hast_gz *g, std::unique_ptr<Batch> b) {
    std::lock_guard<std::mutex> lk(g->mu);
    g->ready.push_back(std::move(b));
    g->cv.notify_all();
}
void publish_error(hast_gz *g, const std::string &msg) {
    std::unique_ptr<Batch> b(new Batch);
    b->error = msg;
    publish(g, std::move(b));
}

// the nominal pass of one segment: search + decode of every chunk, asynchronous on dec_stream
struct Nominal {
    bool launched = false, all_in = false;
    size_t k = 0, c0 = 0, c1 = 0, n_jobs = 0;
    uint64_t input_bits = 0;
    double t0 = 0;
};

// Launches segment k's nominal pass (on the unit whose turn it is) once its compressed bytes are on the device and its arena is
// free.  blocking = false: only if both are the case right now (N.launched says).  Returns "" or what failed; stopped: the stream
// is being closed.
std::string launch_nominal(hast_gz *g, size_t k, size_t n_chunks, uint64_t first_bit, bool blocking, Nominal &N, bool &stopped) {
    const size_t C = g->chunk_bytes, S = g->seg_chunks;
    // the first pass is a short one: the reader (the FASTQ framer, the classification behind it) has nothing to do until it is through
    const size_t S0 = std::min<size_t>(S, 1024);
    const size_t c0 = k == 0 ? 0 : S0 + (k - 1) * S, c1 = std::min(n_chunks, k == 0 ? S0 : c0 + S);
    const bool all_in = c1 >= n_chunks;
    // the compressed bytes this segment's chunks may read: their own and a margin behind them (a chunk runs on to the first
    // block boundary behind its stop)
    const uint64_t need_up = all_in ? g->file_size : std::min<uint64_t>(g->file_size, (uint64_t)c1 * C + std::min<uint64_t>((uint64_t)S * C, 16u << 20));
    Unit &U = g->unit_of(k);
    Arena &A = g->arena_of(k);
    uint64_t have_up = 0;
    N = Nominal{};
    GZ_HIP(hipSetDevice(U.device));
    {
        std::unique_lock<std::mutex> lk(g->mu);
        if (!blocking) {
            if (g->stop || !g->up_error.empty() || g->uploaded < need_up || A.busy) return "";
            if (A.done_recorded && hipEventQuery(A.done) != hipSuccess) return "";
        }
        const double t0 = now_s();
        // (a ring: a blocking launch means that no pass is in flight, i.e. nothing will move `released` on -- bytes the uploader may not
        // write before it does move will never come: a block longer than the ring's margin lies between the chain's end and this segment)
        if (g->ring && blocking && need_up + (uint64_t)g->piece > g->released + g->ring && need_up > g->uploaded)
            return "gz: the ring of compressed bytes on the device is too small for this stream (a deflate block of more than " + std::to_string(g->piece >> 10) + " KB?): segment " +
                   std::to_string(k) + " needs byte " + std::to_string(need_up) + ", the chain stands at " + std::to_string(g->chain.proven_end_bit() >> 3) + ", released " +
                   std::to_string(g->released) + ", uploaded " + std::to_string(g->uploaded) + ", ring " + std::to_string(g->ring) + ", chunks " + std::to_string(g->st.chunks) + " accepted " + std::to_string(g->st.accepted) +
                   " follow-up jobs " + std::to_string(g->st.followup_jobs) + " in " + std::to_string(g->st.followup_rounds) + " rounds";
        g->cv.wait(lk, [&] { return g->stop || !g->up_error.empty() || g->uploaded >= need_up; });
        g->st.wait_upload_s += now_s() - t0;
        if (g->stop) { stopped = true; return ""; }
        if (!g->up_error.empty()) return g->up_error;
        have_up = g->uploaded;
        const double t1 = now_s();
        g->cv.wait(lk, [&] { return g->stop || !A.busy; });
        g->st.wait_consumer_s += now_s() - t1;
        if (g->stop) { stopped = true; return ""; }
    }
    if (A.done_recorded) {                                          // the batch that lived here: its last translate must be through
        TraceStep ts(g, k, "waiting for the arena's last translate");
        GZ_HIP(hipEventSynchronize(A.done));
        A.done_recorded = false;
    }
    A.gap_used = 0;
    N.k = k;
    N.c0 = c0;
    N.c1 = c1;
    N.all_in = all_in;
    N.input_bits = (all_in ? g->file_size : have_up) * 8;
    N.t0 = now_s();
    const int jb = g->jobs_of(k);
    ChunkJob *hj = U.h_jobs[jb];
    hipStream_t ds = g->dec_stream_of(k);
    size_t n_jobs = 0;
    const size_t n_slots = g->pool_slots(c1 - c0);
    {
        TraceStep ts(g, k, "symbol arena (ensure)");
        GZ_HIP(A.syms.ensure(n_slots * g->slot_syms * sizeof(uint16_t) + 64));
        GZ_HIP(A.cursor.ensure(64));
    }
    const uint64_t sym_base = reinterpret_cast<uintptr_t>(A.syms.p) / 2;      // job.sym_off counts u16 from address 0
    for (size_t c = c0; c < c1; ++c) {
        ChunkJob &j = hj[n_jobs];
        memset(&j, 0, sizeof(j));
        const uint64_t nominal = (uint64_t)c * C * 8;
        if (nominal + C * 8 <= first_bit) continue;                           // nothing but the first member's header
        if (nominal <= first_bit) {
            j.from_bit = first_bit;
            j.flags = kJobKnown | kJobNoHistory;
        } else j.from_bit = nominal;
        j.stop_bit = (uint64_t)(c + 1) * C * 8;
        j.search_to_lo = (uint32_t)(j.stop_bit - j.from_bit);
        j.sym_cap = g->slot_syms;
        j.sym_off = sym_base;                                                 // (the pool: the kernel hands the slots out, ChunkJob)
        j.flags |= kJobPoolSlot;
        g->job_view(j);
        ++n_jobs;
    }
    N.n_jobs = n_jobs;
    if (n_jobs) {
        GZ_HIP(hipMemcpyAsync(U.jobs[jb].p, hj, n_jobs * sizeof(ChunkJob), hipMemcpyHostToDevice, ds));
        GZ_HIP(hipMemsetAsync(A.cursor.p, 0, 64, ds));
        GZ_HIP(launch_search((ChunkJob *)U.jobs[jb].p, (uint32_t)n_jobs, U.d_in, N.input_bits, ds));
        GZ_HIP(launch_decode((ChunkJob *)U.jobs[jb].p, (uint32_t)n_jobs, U.d_in, N.input_bits, (uint16_t *)A.syms.p, (uint32_t *)A.cursor.p, (uint32_t)n_slots, ds));
        GZ_HIP(hipMemcpyAsync(hj, U.jobs[jb].p, n_jobs * sizeof(ChunkJob), hipMemcpyDeviceToHost, ds));
    }
    GZ_HIP(hipEventRecord(U.nom_done[jb], ds));
    N.launched = true;
    return "";
}

// what is behind a nominal pass: the chain with its follow-up jobs, windows, CRC-32, member checks, hand-over; returns "" or what failed
std::string finish_segment(hast_gz *g, const Nominal &N, bool &finished) {
    Unit &U = g->unit_of(N.k);
    Arena &A = g->arena_of(N.k);
    GZ_HIP(hipSetDevice(U.device));
    const uint64_t input_bits = N.input_bits;
    const bool all_in = N.all_in;
    g->st.chunks += N.n_jobs;
    if (getenv("HAST_GZ_TRACE_JOBS"))                               // (debugging: every pass's first jobs and all its candidates, as they came back)
        for (size_t i = 0; i < N.n_jobs; ++i) {
            const ChunkJob &j = U.h_jobs[g->jobs_of(N.k)][i];
            if (!(j.status & kStFound) && N.k >= 1) continue;
            fprintf(stderr, "gz job k=%zu i=%zu from=%llu stop=%llu flags=%u status=%u start=%llu end=%llu n_out=%u err=%u adj=%llu limit=%llu input_bits=%llu\n", N.k, i, (unsigned long long)j.from_bit,
                    (unsigned long long)j.stop_bit, j.flags, j.status, (unsigned long long)j.start_bit, (unsigned long long)j.end_bit, j.n_out, j.err_code, (unsigned long long)j.in_adj_words,
                    (unsigned long long)j.limit_bits, (unsigned long long)input_bits);
        }
    // (chain_walk_s: what the host does ALONE between the kernels -- accepting chunks, planning follow-up jobs, combining CRCs: the
    // serial share of one deflate stream however many GPUs decode its passes)
    double t_cw = now_s();
    if (g->slot_adaptive && N.n_jobs >= 64) {                       // the share of chunks that had a block start: the later passes' pools follow it
        size_t found = 0;
        for (size_t i = 0; i < N.n_jobs; ++i) found += (U.h_jobs[g->jobs_of(N.k)][i].status & kStFound) ? 1 : 0;
        const double want = std::min(1.0, (double)found / (double)N.n_jobs * 1.08 + 0.02);
        if (want > g->slot_fraction) g->slot_fraction = want;
    }
    g->chain.add_candidates(U.h_jobs[g->jobs_of(N.k)], N.n_jobs, all_in);
    // ---- the chain, with its follow-up jobs ------------------------------------------------------------------------------------
    std::vector<Chain::Gap> gaps;
    for (;;) {
        const bool more = g->chain.plan(gaps, input_bits);
        g->st.chain_walk_s += now_s() - t_cw;
        if (!more) break;
        if (gaps.size() > g->h_jobs_cap) return "gz: internal: more follow-up jobs than chunks";
        size_t total = 0;
        std::vector<size_t> at(gaps.size());
        for (size_t i = 0; i < gaps.size(); ++i) {
            at[i] = total;
            total += ((size_t)std::min<uint64_t>(gaps[i].want_syms, 1ull << 26) + 520 + 7) & ~(size_t)7;
        }
        if (A.gap_used == A.gap.size()) A.gap.emplace_back();
        DevBuf &gb = A.gap[A.gap_used++];
        GZ_HIP(gb.ensure(total * sizeof(uint16_t) + 64));
        const uint64_t gbase = reinterpret_cast<uintptr_t>(gb.p) / 2;
        for (size_t i = 0; i < gaps.size(); ++i) {
            ChunkJob &j = U.h_fjobs[i];
            j = gaps[i].job;
            j.sym_cap = (uint32_t)std::min<uint64_t>(gaps[i].want_syms, 1ull << 26);
            j.sym_off = gbase + at[i];
            j.start_bit = j.from_bit;
            j.status = kStFound;
            g->job_view(j);
        }
        GZ_HIP(hipMemcpyAsync(U.fjobs.p, U.h_fjobs, gaps.size() * sizeof(ChunkJob), hipMemcpyHostToDevice, U.post_stream));
        GZ_HIP(launch_decode((ChunkJob *)U.fjobs.p, (uint32_t)gaps.size(), U.d_in, input_bits, (uint16_t *)gb.p, nullptr, 0, U.post_stream));
        GZ_HIP(hipMemcpyAsync(U.h_fjobs, U.fjobs.p, gaps.size() * sizeof(ChunkJob), hipMemcpyDeviceToHost, U.post_stream));
        GZ_HIP(hipStreamSynchronize(U.post_stream));
        t_cw = now_s();
        g->chain.gap_done(U.h_fjobs, gaps.size(), input_bits);
        g->st.followup_jobs += gaps.size();
        g->st.followup_rounds++;
    }
    g->st.decode_s += now_s() - N.t0;
    // ---- what became final: windows, CRC-32, member checks, hand-over --------------------------------------------------------------
    std::unique_ptr<Batch> b(new Batch);
    b->unit = (int)(N.k % g->units.size());
    b->arena = (int)((N.k / g->units.size()) % (size_t)g->n_arenas);
    t_cw = now_s();
    g->chain.take_confirmed(b->acc);
    g->st.chain_walk_s += now_s() - t_cw;
    const size_t n = b->acc.size();
    std::string bad;
    if (n) {
        const double t_w0 = now_s();
        std::vector<AccDev> host(n);
        for (size_t i = 0; i < n; ++i) {
            const Accepted &a = b->acc[i];
            host[i].sym = reinterpret_cast<const uint16_t *>((uintptr_t)(a.tag * 2));
            host[i].out_off = a.out_off;
            host[i].n_out = a.job.n_out;
            host[i].no_history = a.no_history ? 1u : 0u;
        }
        b->out_lo = b->acc.front().out_off;
        b->out_hi = b->acc.back().out_off + b->acc.back().job.n_out;
        {
            TraceStep ts(g, N.k, "windows / maps / crc buffers (ensure)");
            GZ_HIP(A.acc.ensure(n * sizeof(AccDev)));
            GZ_HIP(A.windows.ensure(n * (size_t)kWindow));
            GZ_HIP(A.need.ensure(windows_scratch_bytes((uint32_t)n)));
            GZ_HIP(A.crc.ensure(n * sizeof(uint32_t)));
            GZ_HIP(A.carry.ensure(kWindow));
        }
        if (U.h_crc_cap < n) {
            park_pinned(U.h_crc, 0, 2);
            U.h_crc = nullptr;
            U.h_crc_cap = 0;
            GZ_HIP(pinned_malloc((void **)&U.h_crc, (n + n / 2 + 64) * sizeof(uint32_t), hipHostMallocDefault));
            U.h_crc_cap = n + n / 2 + 64;
        }
        // (pageable source: the copy is done with `host` when the call returns)
        {
            TraceStep ts(g, N.k, "accepted chunks to the device");
            GZ_HIP(hipMemcpyAsync(A.acc.p, host.data(), n * sizeof(AccDev), hipMemcpyHostToDevice, U.post_stream));
            GZ_HIP(hipStreamSynchronize(U.post_stream));
        }
        TraceStep ts_w(g, N.k, "windows + CRC-32 kernels, carry back");
        // the 32 KB in front of this batch: what the batch before it left (pinned host memory: the batch before may live on another GPU)
        GZ_HIP(hipMemcpyAsync(A.carry.p, g->h_carry, kWindow, hipMemcpyHostToDevice, U.post_stream));
        GZ_HIP(launch_windows((const AccDev *)A.acc.p, (uint32_t)n, (uint8_t *)A.windows.p, (const uint8_t *)A.carry.p, A.need.p, U.post_stream));
        GZ_HIP(launch_crc((const AccDev *)A.acc.p, (uint32_t)n, (const uint8_t *)A.windows.p, (const uint8_t *)A.carry.p, (uint32_t *)A.crc.p, U.post_stream));
        GZ_HIP(hipMemcpyAsync(U.h_crc, A.crc.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, U.post_stream));
        GZ_HIP(hipStreamSynchronize(U.post_stream));                        // (h_carry has been read: the copy above is through)
        GZ_HIP(hipMemcpyAsync(g->h_carry, (uint8_t *)A.windows.p + (n - 1) * (size_t)kWindow, kWindow, hipMemcpyDeviceToHost, U.post_stream));
        GZ_HIP(hipStreamSynchronize(U.post_stream));
        t_cw = now_s();
        for (size_t i = 0; i < n && bad.empty(); ++i) {
            const Accepted &a = b->acc[i];
            const uint32_t len = a.job.n_out;
            if (len) {
                g->crc_acc = g->crc_started ? crc_combine_op(g->crc_acc, U.h_crc[i], crc_x2nmodp(len, 3)) : U.h_crc[i];
                g->crc_started = true;
                g->isize_acc += len;
            }
            if (a.member_end) {
                if (g->crc_acc != a.want_crc) bad = "gz: CRC-32 mismatch";
                else if ((uint32_t)g->isize_acc != a.want_isize) bad = "gz: length check (ISIZE) failed";
                g->crc_acc = 0;
                g->isize_acc = 0;
                g->crc_started = false;
                g->st.members++;
            }
            if (a.is_gap) g->st.followup_accepted++;
        }
        g->st.chain_walk_s += now_s() - t_cw;
        g->st.accepted += n;
        g->st.out_bytes = b->out_hi;
        g->st.windows_crc_s += now_s() - t_w0;
        if (bad.empty()) {
            std::lock_guard<std::mutex> lk(g->mu);
            A.busy = true;
            g->ready.push_back(std::move(b));
            g->cv.notify_all();
        }
    }
    if (!bad.empty()) return bad;
    if (g->chain.failed()) return g->chain.error();
    if (all_in) {
        if (!g->chain.finished()) return "gz: internal: the chain of chunks stalled";
        finished = true;
    }
    return "";
}

void produce_loop(hast_gz *g) {
    const uint64_t first_bit = g->chain.first_deflate_bit();
    const size_t n_chunks = first_bit == ~0ull ? 0 : (size_t)((g->file_size + g->chunk_bytes - 1) / g->chunk_bytes);
    const size_t nu = g->units.size();
    bool finished = n_chunks == 0, stopped = false, all_launched = false;
    std::string bad;
    // passes in flight, in stream order: while segment k is walked, k + 1 .. k + n_units are on the GPUs (each unit one of its own
    // plus, for the unit of k, the one behind it)
    std::deque<Nominal> fl;
    size_t next_k = 0;
    auto launch = [&](bool blocking) -> bool {                      // true: segment next_k went out
        Nominal N;
        bad = launch_nominal(g, next_k, n_chunks, first_bit, blocking, N, stopped);
        if (!bad.empty() || stopped || !N.launched) return false;
        fl.push_back(N);
        all_launched = N.all_in;
        ++next_k;
        return true;
    };
    const size_t depth = nu * (size_t)(1 + g->ahead);               // passes that may be on the GPUs behind the one being walked
    while (!finished && !stopped && bad.empty()) {
        if (fl.empty()) {
            if (all_launched) { bad = "gz: internal: the chain of chunks stalled"; break; }
            if (!launch(true)) break;
        }
        const Nominal cur = fl.front();
        Unit &U = g->unit_of(cur.k);
        // (ahead: the pass behind the one waited for goes out BEFORE the wait -- it runs beside it)
        while (g->ahead && !all_launched && next_k < cur.k + depth && bad.empty() && !stopped)
            if (!launch(false)) break;
        if (!bad.empty() || stopped) break;
        {
            TraceStep ts(g, cur.k, "waiting for the pass (search + decode)");
            if (hipSetDevice(U.device) != hipSuccess || hipEventSynchronize(U.nom_done[g->jobs_of(cur.k)]) != hipSuccess) { bad = "gz: the decode pass failed"; break; }
        }
        // the next segments' passes go to the GPUs now if their bytes and arenas are there (otherwise behind this segment's hand-over)
        while (!all_launched && next_k <= cur.k + depth && bad.empty() && !stopped)
            if (!launch(false)) break;
        if (!bad.empty() || stopped) break;
        bad = finish_segment(g, cur, finished);
        fl.pop_front();
        if (g->ring) {
            // what no job will read again: everything in front of the chain's end -- and of the first chunk of the oldest pass in flight
            const size_t S0 = std::min<size_t>(g->seg_chunks, 1024);
            const uint64_t next_c0 = S0 + (uint64_t)cur.k * g->seg_chunks;          // (first chunk of segment cur.k + 1)
            const uint64_t rel = std::min<uint64_t>(g->chain.proven_end_bit() >> 3, next_c0 * g->chunk_bytes);
            std::lock_guard<std::mutex> lk(g->mu);
            if (rel > g->released) {
                g->released = rel;
                g->cv.notify_all();
            }
        }
    }
    // (a pass that is still running when the loop is left early reads buffers hast_gz_close frees only after the streams have drained)
    if (stopped) return;
    if (!bad.empty()) publish_error(g, bad);
    else {
        std::unique_ptr<Batch> b(new Batch);
        b->eof = true;
        publish(g, std::move(b));
    }
}

}  // namespace

extern "C" {

hast_status hast_gz_open_multi_ex(hast_ctx *const *ctxs, int n_ctx, const char *path, size_t chunk_bytes, size_t seg_chunks, double room, hast_gz **out) {
    if (!ctxs || n_ctx < 1 || !path || !out) return set_error(HAST_ERR_INVALID, "null argument");
    for (int i = 0; i < n_ctx; ++i)
        if (!ctxs[i]) return set_error(HAST_ERR_INVALID, "null argument");
    *out = nullptr;
    const double t_open0 = now_s();
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return set_error(HAST_ERR_IO, "cannot read %s", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
        close(fd);
        return set_error(HAST_ERR_UNSUPPORTED, "%s is not a regular file: the device inflate reads by position", path);
    }
    std::unique_ptr<hast_gz> g(new (std::nothrow) hast_gz());
    if (!g) {
        close(fd);
        return set_error(HAST_ERR_OOM, "host allocation failed");
    }
    g->ctx = ctxs[0];
    g->fd = fd;
    g->path = path;
    g->file_size = (uint64_t)sb.st_size;
    // one unit per GPU: contexts that share a device share the unit (HAST_GZ_SPLIT=contexts: a unit per context all the same, and
    // every reader's bytes through the bounce buffer + peer copy -- the several-GPU code on one GPU, for tests)
    {
        const char *e = getenv("HAST_GZ_SPLIT");
        g->split_same_device = e && !strcmp(e, "contexts");
    }
    for (int i = 0; i < n_ctx; ++i) {
        const int dev = hast_ctx_device(ctxs[i]);
        bool have = false;
        for (const auto &u : g->units) have = have || u->device == dev;
        if (have && !g->split_same_device) continue;
        std::unique_ptr<Unit> u(new (std::nothrow) Unit());
        if (!u) {
            close(fd);
            return set_error(HAST_ERR_OOM, "host allocation failed");
        }
        u->device = dev;
        g->units.push_back(std::move(u));
    }
    // (test switches for callers that cannot pass a geometry, e.g. the classify program: a small file then spans many passes)
    if (!chunk_bytes)
        if (const char *e = getenv("HAST_GZ_CHUNK_BYTES")) chunk_bytes = (size_t)std::max(0L, atol(e));
    if (!seg_chunks)
        if (const char *e = getenv("HAST_GZ_PASS_CHUNKS")) seg_chunks = (size_t)std::max(0L, atol(e));
    // Round 6: 16-KB chunks, 6144 a pass, 20 symbols of room per compressed byte (until then 32 KB / 4096 / 12).  A decode pass lasts as long
    // as its slowest wave and its waves issue for 39 % of their lifetime (profiles/round5_gz_kernels_pmc.txt): what it loses is occupancy,
    // so a pass wants more and finer work units than the GPU has wave slots (4480).  With 16-KB chunks a wave's work is ~one deflate block
    // instead of one to two, 42 % of the chunks hold no block start and cost nothing, and the hardware hands a freed wave slot the
    // next chunk: the decode kernels' time on 20M reads 247 -> 180 ms (constant quality lines) / 339 -> 268 ms (noisy ones), with an arena of
    // 2.8 GB instead of 3.2 GB because its slots are handed out on the device (slot_fraction; profiles/round6_gz_geom.txt).
    const bool default_geometry = !chunk_bytes && !seg_chunks && room <= 0;
    g->chunk_bytes = chunk_bytes ? std::max<size_t>(chunk_bytes, 64) : 16384;
    g->chunk_bytes = (g->chunk_bytes + 3) & ~(size_t)3;
    // 4096 chunks a pass x 786 KB of symbol room = 3.2 GB an arena (measured against the tree before the pipelined
    // producer, alternating on one box: passes of 2048 chunks were slower than that tree, 4096 10 % faster, 8192 no faster).  Keep the
    // footprint small: on some boxes of the pool ONE HIP call of a process that starts right after another one freed tens of GB blocks
    // for 0.7-6 s (hipMalloc or hipStreamCreate, whichever comes first -- tools/probe/malloc_probe.py; the tree before did the same there).
    g->seg_chunks = seg_chunks ? seg_chunks : (chunk_bytes ? 4096 : 6144);
    if (const char *e = getenv("HAST_GZ_AHEAD")) g->ahead = std::min(Unit::kDecStreams - 1, std::max(0, atoi(e)));
    g->n_arenas = g->ahead + 2;
    if (const char *e = getenv("HAST_GZ_ARENAS")) g->n_arenas = std::min((int)Unit::kArenas, std::max(g->ahead + 2, atoi(e)));
    if (room <= 0)
        if (const char *e = getenv("HAST_GZ_ROOM")) room = atof(e);          // (measurements: symbols of room per compressed byte of a chunk)
    if (const char *e = getenv("HAST_GZ_SLOT_FRACTION")) {
        g->slot_fraction = std::min(1.0, std::max(0.01, atof(e)));
        g->slot_adaptive = false;
    }
    g->room = room > 0 ? room : (default_geometry ? 20.0 : 12.0);
    if (g->chunk_bytes > (1u << 26)) { close(fd); return set_error(HAST_ERR_INVALID, "chunk_bytes too large"); }
    g->slot_syms = (uint32_t)std::min<double>((double)(1u << 27), (double)g->chunk_bytes * g->room + 600);
    g->slot_syms = (g->slot_syms + 7) & ~7u;
    if (const char *e = getenv("HAST_GZ_PIECE_BYTES")) g->piece = (size_t)std::min<long>((long)kPiece, std::max(4096L, atol(e) & ~4095L));
    const int cfd = fd;
    g->chain.begin(g->file_size, [cfd](uint64_t off, size_t n, uint8_t *dst, size_t *got) { return read_at(cfd, off, n, dst, got); });
    if (g->chain.failed()) {
        // no gzip magic (zlib's gzread passes such a file through as it is), or a header this decoder does not take
        const std::string why = g->chain.error();
        close(fd);
        return set_error(HAST_ERR_UNSUPPORTED, "%s: %s", path, why.c_str());
    }
    const size_t n_chunks = (size_t)((g->file_size + g->chunk_bytes - 1) / g->chunk_bytes);
    const size_t seg = std::max<size_t>(1, std::min(g->seg_chunks, n_chunks));
    g->seg_chunks = seg;
    const size_t s0 = std::min<size_t>(seg, 1024);
    const size_t n_seg = n_chunks <= s0 ? 1 : 1 + (n_chunks - s0 + seg - 1) / seg;
    if (g->units.size() > n_seg) g->units.resize(std::max<size_t>(1, n_seg));       // (a unit without a segment would hold memory for nothing)
    if (n_seg <= 2 && g->slot_adaptive) g->slot_fraction = 1.0;                      // (a file of one or two passes: nothing to learn from, little to save)
    const size_t nu = g->units.size();
    {
        // the ring (hast_gz): at least what the passes in flight span -- the segment being walked, one per unit in front of it, the
        // margin a pass waits for behind its last chunk -- plus how far the chain's end may lie in front of the segment being walked
        // (a deflate block: at most a piece, or the stream is refused), the piece in flight, the piece being filled and one of slack;
        // by default 2 GB, so that only files beyond that (HAST's real inputs: 50-100 GB) go round
        const uint64_t pass = (uint64_t)seg * g->chunk_bytes;
        const uint64_t need = (uint64_t)(nu * (size_t)(1 + g->ahead) + 2) * pass + std::min<uint64_t>(pass, 16u << 20) + 4 * (uint64_t)g->piece;
        uint64_t ring = 2ull << 30;
        if (const char *e = getenv("HAST_GZ_RING_BYTES")) ring = (uint64_t)std::max(0L, atol(e));
        ring = std::max(ring, need);
        ring = (ring + g->piece - 1) / g->piece * g->piece;
        if (g->file_size + kInPad > ring + g->piece) g->ring = ring;
        g->chain.set_eager(g->ring != 0);
    }
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    const bool trace = getenv("HAST_GZ_TRACE") != nullptr;
    double t_tr = now_s();
    auto tr = [&](const char *what) {
        if (trace) fprintf(stderr, "gz open %s: %s %.3f s\n", path, what, now_s() - t_tr);
        t_tr = now_s();
    };
    g->h_jobs_cap = seg + 8;
    step(hipSetDevice(g->units[0]->device));
    step(pinned_malloc((void **)&g->h_carry, kWindow, hipHostMallocPortable));
    if (e == hipSuccess) memset(g->h_carry, 0, kWindow);
    for (size_t ui = 0; ui < nu && e == hipSuccess; ++ui) {
        Unit &U = *g->units[ui];
        step(hipSetDevice(U.device));
        // the file's bytes, then zeros: the kernels read whole words and a little past the last real bit
        const uint64_t alloc = g->ring ? g->ring + g->piece + 2 * kInPad : ((g->file_size + 3) & ~(uint64_t)3) + kInPad, tail_from = g->file_size & ~(uint64_t)3;
        step(dev_malloc((void **)&U.d_in, alloc));
        tr("input buffer");
        step(hipStreamCreateWithFlags(&U.up_stream, hipStreamNonBlocking));
        // (on the stream the file's pieces will come in on: a hipMemset of device memory does not wait for the host, and nothing orders
        // the legacy stream it runs on against a non-blocking one -- the zeros could land on the file's last bytes after the upload)
        if (e == hipSuccess && !g->ring) step(hipMemsetAsync(reinterpret_cast<uint8_t *>(U.d_in) + tail_from, 0, alloc - tail_from, U.up_stream));
        {
            // The nominal passes keep some CUs FREE (HAST_GZ_FREE_CUS, default 32 of 256): a decode wave lives for milliseconds and the passes
            // fill every LDS slot of the GPU, so whatever else wants to run -- this file's follow-up jobs, windows and CRC-32, the translate
            // kernel, the FASTQ framer and the classifier of the blocks already inflated -- would wait for waves to retire, one short kernel
            // after the other (measured: 0.2-0.35 s of "waiting for the GPU's framing" in a .gz run against 0.07 s on plain files).
            int n_cu = 0, free_cus = 32;
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, U.device) == hipSuccess) n_cu = prop.multiProcessorCount;
            if (const char *fc = getenv("HAST_GZ_FREE_CUS")) free_cus = atoi(fc);
            if (e == hipSuccess && n_cu > 0 && free_cus > 0 && free_cus < n_cu) {
                U.dec_stream[0] = masked_stream_get(U.device, n_cu, free_cus);
                if (U.dec_stream[0]) U.dec_masked_free[0] = free_cus;
            }
            if (!U.dec_stream[0]) step(hipStreamCreateWithFlags(&U.dec_stream[0], hipStreamNonBlocking));
            for (int i = 1; i <= g->ahead && n_seg > 2; ++i) {       // (the other passes that run beside it)
                if (U.dec_masked_free[0]) {
                    U.dec_stream[i] = masked_stream_get(U.device, n_cu, free_cus);
                    if (U.dec_stream[i]) U.dec_masked_free[i] = free_cus;
                }
                if (!U.dec_stream[i]) step(hipStreamCreateWithFlags(&U.dec_stream[i], hipStreamNonBlocking));
            }
        }
        // what follows a nominal pass (follow-up jobs, windows, CRC-32) must not queue behind the NEXT segment's pass: a stream of its own
        // (a high-priority one made no difference in an A/B: the free CUs are what lets its kernels start); the reader's translate kernels
        // on another one
        step(hipStreamCreateWithFlags(&U.post_stream, hipStreamNonBlocking));
        step(hipStreamCreateWithFlags(&U.xl_stream, hipStreamNonBlocking));
        step(hipEventCreateWithFlags(&U.xl_done, hipEventDisableTiming));
        tr("streams");
        for (int i = 0; i < Unit::kJobCopies; ++i) {
            step(pinned_malloc((void **)&U.h_jobs[i], g->h_jobs_cap * sizeof(ChunkJob), hipHostMallocDefault));
            step(U.jobs[i].ensure(g->h_jobs_cap * sizeof(ChunkJob)));
            step(hipEventCreateWithFlags(&U.nom_done[i], hipEventDisableTiming));
        }
        step(pinned_malloc((void **)&U.h_fjobs, g->h_jobs_cap * sizeof(ChunkJob), hipHostMallocDefault));
        step(U.fjobs.ensure(g->h_jobs_cap * sizeof(ChunkJob)));
        for (Arena &a : U.arena) step(hipEventCreateWithFlags(&a.done, hipEventDisableTiming));
        tr("job buffers");
        // the unit's first symbol arena now, the other one when its first pass is launched (launch_nominal) -- 3.2 GB each: 5-10 ms of
        // hipMalloc on most boxes of the pool, 0.6 s on some (profiles/round5_hipstall_slow_box_parked.txt), so there are as few of them
        // as the pipeline needs (one pass at a time: TWO arenas decode as fast as three, profiles/round5_ab_gz_two_arenas.txt; two passes side
        // by side: three) and the first one has its
        // final size at once although the file's first pass is a short one (it would be parked and allocated again two passes on).
        // That there IS room for the later one is checked here, so that a device without it is refused at the door (the caller then
        // inflates on the host) instead of failing in mid-file
        // HAST_GZ_PREALLOC=1: EVERYTHING a stream's passes will need is allocated here instead of at its first use (the later arenas, every
        // arena's window / map / CRC buffers).  Tried on the round's last day against the runs whose read phase takes 0.6-1.7 s instead of
        // 0.2 s (one in five on some boxes of the pool, none on others): there a hipMalloc in the middle of the read phase took 0.6-1.3 s
        // and held up every stream of the process while it did ("windows / maps / crc buffers (ensure) 1.330 s" in both files' producers
        // at once, profiles/round6_gz_slow_hunt.txt).  Allocating here MOVES that stall (open_s 0.8-2.5 s, the k-mer load beside it waiting
        // as long), it does not remove it: the whole process takes as long either way (40 runs each, alternating) -- a process that starts
        // right behind another one's exit pays it at one of its first large allocations.  Off by default
        static const bool prealloc = getenv("HAST_GZ_PREALLOC") && atoi(getenv("HAST_GZ_PREALLOC")) > 0;
        size_t later = 0;
        for (int i = 0; i < g->n_arenas && e == hipSuccess; ++i) {
            const size_t k = ui + (size_t)i * nu;
            if (k >= n_seg) break;
            const size_t first = n_seg == 1 ? std::max(s0, std::min(seg, n_chunks)) : s0;
            const size_t chunks = k == 0 ? (n_seg > (size_t)g->n_arenas * nu ? std::max(first, seg) : first) : seg;
            const size_t bytes = g->pool_slots(chunks) * (size_t)g->slot_syms * sizeof(uint16_t) + 64;
            Arena &a = U.arena[i];
            if (i == 0 || prealloc) step(a.syms.ensure(bytes));
            else later += bytes + chunks * ((size_t)kWindow * 3 + 64);              // (+ windows, maps, per-chunk words of a batch)
            if (prealloc) {
                const size_t n_max = chunks + 64;                                   // accepted chunks of a batch: a pass's, and a few follow-up jobs'
                step(a.cursor.ensure(64));
                step(a.acc.ensure(n_max * sizeof(AccDev)));
                step(a.windows.ensure(n_max * (size_t)kWindow));
                step(a.need.ensure(windows_scratch_bytes((uint32_t)n_max)));
                step(a.crc.ensure(n_max * sizeof(uint32_t)));
                step(a.carry.ensure(kWindow));
                a.gap.emplace_back();
                step(a.gap[0].ensure(8u << 20));                                    // (a round of follow-up jobs' symbols; grows when one needs more)
                if (U.h_crc_cap < n_max && e == hipSuccess) {
                    park_pinned(U.h_crc, 0, 2);
                    U.h_crc = nullptr;
                    U.h_crc_cap = 0;
                    step(pinned_malloc((void **)&U.h_crc, (n_max + n_max / 2 + 64) * sizeof(uint32_t), hipHostMallocDefault));
                    if (e == hipSuccess) U.h_crc_cap = n_max + n_max / 2 + 64;
                }
            }
        }
        if (e == hipSuccess && later) {
            size_t free_b = 0, total_b = 0;
            step(hipMemGetInfo(&free_b, &total_b));
            if (e == hipSuccess && free_b < later + (later >> 2) + (1ull << 30)) e = hipErrorOutOfMemory;
        }
        tr("arenas");
    }
    if (e != hipSuccess) {
        const hast_status st = set_error(e == hipErrorOutOfMemory ? HAST_ERR_UNSUPPORTED : HAST_ERR_HIP, "device inflate of %s: %s", path, hipGetErrorString(e));
        (void)hipGetLastError();
        hast_gz_close(g.release());
        return st;
    }
    hast_gz *raw = g.release();
    raw->st.open_s = now_s() - t_open0;
    raw->st.ring_bytes = raw->ring;
    raw->uploader = std::thread(upload_loop, raw);
    raw->producer = std::thread(produce_loop, raw);
    *out = raw;
    return HAST_OK;
}

hast_status hast_gz_open_multi(hast_ctx *const *ctxs, int n_ctx, const char *path, hast_gz **out) { return hast_gz_open_multi_ex(ctxs, n_ctx, path, 0, 0, 0, out); }
hast_status hast_gz_open_ex(hast_ctx *ctx, const char *path, size_t chunk_bytes, size_t seg_chunks, double room, hast_gz **out) {
    return hast_gz_open_multi_ex(&ctx, 1, path, chunk_bytes, seg_chunks, room, out);
}
hast_status hast_gz_open(hast_ctx *ctx, const char *path, hast_gz **out) { return hast_gz_open_multi_ex(&ctx, 1, path, 0, 0, 0, out); }
int hast_gz_units(const hast_gz *g) { return g ? (int)g->units.size() : 0; }

void hast_gz_close(hast_gz *g) {
    if (!g) return;
    const bool trace = getenv("HAST_GZ_TRACE") != nullptr;
    auto tr = [&](const char *what) {
        if (trace) fprintf(stderr, "gz close: %s\n", what);
    };
    tr("begin");
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->stop = true;
        for (auto &u : g->units)
            for (Arena &a : u->arena) a.busy = false;
        g->cv.notify_all();
    }
    if (g->uploader.joinable()) g->uploader.join();
    tr("uploader joined");
    if (g->producer.joinable()) g->producer.join();
    tr("producer joined");
    for (auto &up : g->units) {
        Unit &U = *up;
        (void)hipSetDevice(U.device);
        for (hipStream_t st : {U.dec_stream[0], U.dec_stream[1], U.dec_stream[2], U.post_stream, U.up_stream, U.xl_stream})
            if (st) (void)hipStreamSynchronize(st);
        tr("streams drained");
        for (Arena &a : U.arena) {
            if (a.done_recorded) (void)hipEventSynchronize(a.done);
            if (a.done) (void)hipEventDestroy(a.done);
            for (DevBuf *b : {&a.syms, &a.windows, &a.acc, &a.need, &a.crc, &a.carry, &a.cursor}) b->release();
            for (DevBuf &b : a.gap) b.release();
        }
        for (int i = 0; i < Unit::kJobCopies; ++i) {
            U.jobs[i].release();
            park_pinned(U.h_jobs[i], g->h_jobs_cap * sizeof(ChunkJob), 1);
            if (U.nom_done[i]) (void)hipEventDestroy(U.nom_done[i]);
        }
        U.fjobs.release();
        U.bounce.release();
        park_pinned(U.h_fjobs, g->h_jobs_cap * sizeof(ChunkJob), 1);
        park_device(U.d_in, g->ring ? (size_t)(g->ring + g->piece) : (size_t)g->file_size + kInPad, 1);
        park_pinned(U.h_crc, 0, 1);
        if (U.xl_done) (void)hipEventDestroy(U.xl_done);
        if (U.up_stream) (void)hipStreamDestroy(U.up_stream);
        for (int i = 0; i < Unit::kDecStreams; ++i) {
            if (U.dec_stream[i] && U.dec_masked_free[i]) masked_stream_put(U.device, U.dec_masked_free[i], U.dec_stream[i]);     // (drained above)
            else if (U.dec_stream[i]) (void)hipStreamDestroy(U.dec_stream[i]);
        }
        if (U.post_stream) (void)hipStreamDestroy(U.post_stream);
        if (U.xl_stream) (void)hipStreamDestroy(U.xl_stream);
    }
    for (auto &de : g->in_events) {
        (void)hipSetDevice(de.first);
        (void)hipEventDestroy(de.second);
    }
    park_pinned(g->h_carry, kWindow, 1);
    tr("freed");
    if (g->fd >= 0) close(g->fd);
    delete g;
}

hast_status hast_gz_read_device(hast_gz *g, uint8_t *d_dst, size_t cap, size_t *n_out, hast_stream s) {
    if (!g || !n_out || (cap && !d_dst)) return set_error(HAST_ERR_INVALID, "null argument");
    *n_out = 0;
    if (!g->err.empty()) return set_error(HAST_ERR_IO, "%s: %s", g->path.c_str(), g->err.c_str());
    // where the caller's buffer lives: the bytes are translated on the unit that decoded them, straight into the buffer when that is
    // the same GPU, through the unit's bounce buffer and a peer copy otherwise
    int dst_dev = g->units[0]->device;
    if (cap && g->units.size() + (g->split_same_device ? 1 : 0) > 1) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, d_dst) != hipSuccess) {
            (void)hipGetLastError();
            return set_error(HAST_ERR_INVALID, "hast_gz_read_device: the destination is not device memory");
        }
        dst_dev = at.device;
    }
    if (hipSetDevice(dst_dev) != hipSuccess) return set_error(HAST_ERR_HIP, "hipSetDevice failed");
    hipStream_t hs = s ? (hipStream_t)s : ctx_stream_of(g->ctx);
    // the translate kernels run on the unit's own stream: it waits for what the caller's stream holds now (the buffer's previous
    // readers), and the caller's stream waits for them in turn
    hipEvent_t ev_in = nullptr;
    for (auto &de : g->in_events)
        if (de.first == dst_dev) ev_in = de.second;
    if (!ev_in) {
        if (hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) != hipSuccess) return set_error(HAST_ERR_HIP, "hipEventCreate failed");
        g->in_events.push_back({dst_dev, ev_in});
    }
    bool in_recorded = false;
    int last_unit = -1;
    auto join_back = [&]() -> hast_status {                           // the caller's stream behind the launches on last_unit's stream
        if (last_unit < 0) return HAST_OK;
        Unit &U = *g->units[(size_t)last_unit];
        if (hipSetDevice(U.device) != hipSuccess || hipEventRecord(U.xl_done, U.xl_stream) != hipSuccess || hipSetDevice(dst_dev) != hipSuccess ||
            hipStreamWaitEvent(hs, U.xl_done, 0) != hipSuccess)
            return set_error(HAST_ERR_HIP, "hast_gz_read_device: stream hand-over failed");
        last_unit = -1;
        return HAST_OK;
    };
    size_t got = 0;
    while (got < cap && !g->done) {
        if (!g->cur) {
            std::unique_lock<std::mutex> lk(g->mu);
            const double t0 = now_s();
            g->cv.wait(lk, [&] { return !g->ready.empty(); });
            g->st.wait_decode_s += now_s() - t0;
            g->cur = std::move(g->ready.front());
            g->ready.pop_front();
        }
        Batch &b = *g->cur;
        if (!b.error.empty()) {
            g->err = b.error;
            g->cur.reset();
            // what was decoded in front of the damage has been delivered (as zlib's gzread does); the error comes with the next call
            if (got) break;
            return set_error(HAST_ERR_IO, "%s: %s", g->path.c_str(), g->err.c_str());
        }
        if (b.eof) {
            g->done = true;
            g->cur.reset();
            break;
        }
        Unit &U = *g->units[(size_t)b.unit];
        Arena &A = U.arena[b.arena];
        const bool direct = U.device == dst_dev && !g->split_same_device;
        if (last_unit >= 0 && last_unit != b.unit)
            if (hast_status st = join_back()) return st;
        if (!in_recorded) {
            if (hipSetDevice(dst_dev) != hipSuccess || hipEventRecord(ev_in, hs) != hipSuccess) return set_error(HAST_ERR_HIP, "hipEventRecord failed");
            in_recorded = true;
        }
        if (hipSetDevice(U.device) != hipSuccess) return set_error(HAST_ERR_HIP, "hipSetDevice failed");
        if (last_unit != b.unit && hipStreamWaitEvent(U.xl_stream, ev_in, 0) != hipSuccess) return set_error(HAST_ERR_HIP, "hipStreamWaitEvent failed");
        last_unit = b.unit;
        // the bytes [cursor, cursor + want) out of this batch's chunks
        const uint64_t o_lo = g->cursor, o_hi = std::min<uint64_t>(b.out_hi, g->cursor + (cap - got));
        if (o_hi > o_lo) {
            uint8_t *xl_dst = d_dst + got;
            if (!direct) {
                if (U.bounce.bytes < cap) {
                    // (the stream still reads the old one: drained first; happens once per unit and reader block size)
                    if (hipStreamSynchronize(U.xl_stream) != hipSuccess || U.bounce.ensure(cap) != hipSuccess) return set_error(HAST_ERR_OOM, "hast_gz_read_device: no room for the hand-over buffer");
                }
                xl_dst = static_cast<uint8_t *>(U.bounce.p);
            }
            size_t c = b.next;
            while (c < b.acc.size()) {
                // launches of at most 65535 chunks (grid.y)
                size_t c_end = c;
                uint32_t max_syms = 0;
                while (c_end < b.acc.size() && c_end - c < 65535 && b.acc[c_end].out_off < o_hi) {
                    max_syms = std::max(max_syms, b.acc[c_end].job.n_out);
                    ++c_end;
                }
                if (c_end == c) break;
                const hipError_t e = launch_translate((const AccDev *)A.acc.p, (uint32_t)c, (uint32_t)(c_end - c), max_syms, (const uint8_t *)A.windows.p,
                                                      (const uint8_t *)A.carry.p, o_lo, o_hi, xl_dst, U.xl_stream);
                if (e != hipSuccess) return set_error(HAST_ERR_HIP, "translate: %s", hipGetErrorString(e));
                c = c_end;
            }
            if (!direct && hipMemcpyPeerAsync(d_dst + got, dst_dev, xl_dst, U.device, (size_t)(o_hi - o_lo), U.xl_stream) != hipSuccess)
                return set_error(HAST_ERR_HIP, "hast_gz_read_device: peer copy failed");
            got += (size_t)(o_hi - o_lo);
            g->cursor = o_hi;
            while (b.next < b.acc.size() && b.acc[b.next].out_off + b.acc[b.next].job.n_out <= g->cursor) b.next++;
        }
        if (g->cursor >= b.out_hi) {
            // the batch is through: its arena is free once the unit's stream has run the launches above
            if (hipEventRecord(A.done, U.xl_stream) != hipSuccess) return set_error(HAST_ERR_HIP, "hipEventRecord failed");
            {
                std::lock_guard<std::mutex> lk(g->mu);
                A.done_recorded = true;
                A.busy = false;
                g->cv.notify_all();
            }
            g->cur.reset();
        }
    }
    if (hast_status st = join_back()) return st;
    (void)hipSetDevice(dst_dev);
    *n_out = got;
    return HAST_OK;
}

hast_status hast_gz_get_stats(hast_gz *g, hast_gz_stats *out) {
    if (!g || !out) return set_error(HAST_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    *out = g->st;
    out->compressed_bytes = g->file_size;
    return HAST_OK;
}

}  // extern "C"

// ingest.h -- multi-threaded FASTQ ingest + barcode dictionary for the classify CLI.
//
// Replaces the reference's single producer thread (01.classify_stlfr_reads/classify.cpp:257-273:
// one getline at a time through a 303-byte gzstream buffer) while accepting exactly the same input
// framing (classify.cpp:238-278):
//   * gzip iff the file NAME ends in ".gz", otherwise raw bytes;
//   * a record is four '\n'-separated lines counted from the start of the file; only line 1 (header)
//     and line 2 (bases) are used; no '@'/'+' validation, no CR stripping;
//   * input ends at the first header line that reaches EOF before a '\n' (that record is dropped);
//     an unterminated line 2..4 of the last record is still accepted.
// Design: a reader thread streams the file in large blocks (pread / gzread); T workers index the
// newlines of a block in parallel, then parse whole records in parallel: barcode name -> dense id through a
// sharded concurrent dictionary (parseName, classify.cpp:112-119), bases copied straight into the pinned
// staging buffer the GPU library handed out (hast_batch_begin).  Record order inside a batch is preserved,
// although nothing downstream depends on it (counts are sums).
#pragma once
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include "../../include/hast.h"
#include "bgzf_reader.h"
#include "fast_inflate.h"
#include "par_inflate.h"
#include "worker_pool.h"

namespace hast {

// ------------------------------------------------------------------------------------------------
// barcode dictionary: BarcodeCache keys (classify.cpp:51) -> dense ids, concurrent
// ------------------------------------------------------------------------------------------------
class BarcodeDict {
  public:
    BarcodeDict() : shards_(kShards) {}
    static uint64_t hash(std::string_view s) {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (s.size() * 0xff51afd7ed558ccdull);
        size_t i = 0;
        for (; i + 8 <= s.size(); i += 8) {
            uint64_t w;
            memcpy(&w, s.data() + i, 8);
            h = (h ^ w) * 0x9FB21C651E98DF25ull;
            h ^= h >> 29;
        }
        uint64_t w = 0;
        memcpy(&w, s.data() + i, s.size() - i);
        h = (h ^ w) * 0x9FB21C651E98DF25ull;
        return h ^ (h >> 32);
    }
    // thread-safe; `cache` is a per-thread front cache (hot barcodes such as 0_0_0 never touch a lock)
    static constexpr size_t kCacheSlots = 1u << 15;         // 1.5 MB per thread: holds the barcodes in flight
    struct Cache {
        // the text of short barcodes sits IN the entry (no second cache miss for the comparison)
        struct E { uint64_t h = 0; const char *p = nullptr; uint32_t len = 0, id = 0; char inl[16] = {0}; };
        std::vector<E> e = std::vector<E>(kCacheSlots);
    };
    uint32_t get(std::string_view bc, Cache &cache) {
        const uint64_t h = hash(bc);
        Cache::E &ce = cache.e[h & (kCacheSlots - 1)];
        if (ce.p && ce.h == h && ce.len == bc.size() && memcmp(bc.size() <= 16 ? ce.inl : ce.p, bc.data(), bc.size()) == 0) return ce.id;
        Shard &s = shards_[(h >> 40) & (kShards - 1)];
        std::lock_guard<std::mutex> g(s.mu);
        if (s.slots.empty()) s.slots.resize(64);
        size_t mask = s.slots.size() - 1, i = (size_t)h & mask;
        for (;;) {
            Slot &sl = s.slots[i];
            if (!sl.p) break;
            if (sl.h == h && sl.len == bc.size() && memcmp(sl.p, bc.data(), bc.size()) == 0) {
                fill(ce, h, sl.p, sl.len, sl.id);
                return sl.id;
            }
            i = (i + 1) & mask;
        }
        // insert
        const char *p = s.store(bc);
        const uint32_t id = next_id_.fetch_add(1, std::memory_order_relaxed);
        s.slots[i] = {h, p, (uint32_t)bc.size(), id};
        if (++s.n * 2 > s.slots.size()) s.grow();
        fill(ce, h, p, (uint32_t)bc.size(), id);
        return id;
    }
    static void fill(Cache::E &ce, uint64_t h, const char *p, uint32_t len, uint32_t id) {
        ce.h = h;
        ce.p = p;
        ce.len = len;
        ce.id = id;
        if (len <= 16) memcpy(ce.inl, p, len);
    }
    size_t size() const { return next_id_.load(std::memory_order_relaxed); }
    // names by id (call when no thread is inserting); names_range: the share of shards [lo, hi) of n_shards(), for several threads
    static constexpr size_t n_shards() { return kShards; }
    void names_range(std::vector<std::string_view> &out, size_t lo, size_t hi) const {
        for (size_t k = lo; k < hi && k < shards_.size(); ++k)
            for (const Slot &sl : shards_[k].slots)
                if (sl.p) out[sl.id] = std::string_view(sl.p, sl.len);
    }
    std::vector<std::string_view> names() const {
        std::vector<std::string_view> out(size());
        for (const Shard &s : shards_)
            for (const Slot &sl : s.slots)
                if (sl.p) out[sl.id] = std::string_view(sl.p, sl.len);
        return out;
    }

  private:
    static constexpr size_t kShards = 4096;
    struct Slot { uint64_t h = 0; const char *p = nullptr; uint32_t len = 0, id = 0; };
    struct Shard {
        std::mutex mu;
        std::vector<Slot> slots;
        size_t n = 0;
        std::vector<std::unique_ptr<char[]>> arena;
        size_t arena_used = 0, arena_cap = 0;
        const char *store(std::string_view s) {
            if (arena_used + s.size() + 1 > arena_cap) {
                arena_cap = std::max<size_t>(1 << 14, s.size() + 1);
                arena.emplace_back(new char[arena_cap]);
                arena_used = 0;
            }
            char *p = arena.back().get() + arena_used;
            memcpy(p, s.data(), s.size());
            p[s.size()] = 0;
            arena_used += s.size() + 1;
            return p;
        }
        void grow() {
            std::vector<Slot> old;
            old.swap(slots);
            slots.resize(old.size() * 2);
            size_t mask = slots.size() - 1;
            for (const Slot &o : old)
                if (o.p) {
                    size_t i = (size_t)o.h & mask;
                    while (slots[i].p) i = (i + 1) & mask;
                    slots[i] = o;
                }
        }
    };
    std::vector<Shard> shards_;
    std::atomic<uint32_t> next_id_{0};
};

// ------------------------------------------------------------------------------------------------
// block source: raw or gz bytes in large blocks, produced by a background reader thread.  gz files go through the
// in-tree decoders; HAST_INFLATE=zlib in the environment switches back to zlib's gzread:
//   * an ordinary gzip file (one long deflate stream: what `gzip` writes) is inflated by SEVERAL threads at once
//     (par_inflate.h: block boundaries searched, chunks decoded with an unknown window, windows resolved in order, every
//     member CRC-checked; HAST_GZ_THREADS, 1 = off);
//   * blocked gzip (BGZF) is recognised by its first member and inflated member by member, side by side
//     (bgzf_reader.h; HAST_BGZF_THREADS);
//   * anything else (a pipe, a ".gz" that is not gzip) goes through the serial decoder (fast_inflate.h: about twice zlib's
//     speed on FASTQ, members CRC-checked).
// A damaged gz file ends the stream early and sets error().
// ------------------------------------------------------------------------------------------------
class BlockSource {
  public:
    ~BlockSource() { close(); }
    // threaded = false: no background reader; the caller pulls the bytes with read_into() from a thread of its own
    // (the GPU-framing path reads straight into pinned staging memory)
    bool open(const std::string &path, size_t block_bytes, bool threaded = true) {
        close();
        const size_t n = path.size();
        gz_mode_ = n > 3 && path.compare(n - 3, 3, ".gz") == 0;               // classify.cpp:245-249
        const char *which = getenv("HAST_INFLATE");
        use_zlib_ = which && strcmp(which, "zlib") == 0;
        error_.clear();
        if (gz_mode_ && use_zlib_) {
            gz_ = gzopen(path.c_str(), "rb");
            if (!gz_) return false;
            gzbuffer(gz_, 4u << 20);
        } else if (gz_mode_) {
            fp_ = fopen(path.c_str(), "rb");
            if (!fp_) return false;
            // blocked gzip (BGZF) is inflated by several threads at once; anything else is one serial stream
            bgzf_ = BgzfReader::probe(fp_);
            if (bgzf_) {
                const unsigned hw = std::thread::hardware_concurrency();
                const char *e = getenv("HAST_BGZF_THREADS");
                bgzf_reader_.open(fp_, e ? atoi(e) : (int)std::min(16u, std::max(2u, hw / 8)));
            } else if (gz_threads() > 1 && ParGzReader::usable(fp_)) {
                // an ordinary gzip stream in a regular file: several threads at once (par_inflate.h).  The files open at
                // the same time share a budget of inflate threads (a thread inflates ~1 GB/s alone, ~0.6 GB/s as one of 16:
                // four files with 8 threads each were slower than with 4 each on the 1-GPU share of an MI355X host)
                pargz_ = true;
                const int open_now = gz_open_files().fetch_add(1) + 1;
                par_reader_.open(fp_, std::max(2, std::min(gz_threads(), gz_budget() / open_now)));
            } else inflater_.open(fp_, 4u << 20);
        } else if (path == "-") {
            fp_ = stdin;
        } else {
            fp_ = fopen(path.c_str(), "rb");
            if (!fp_) return false;
        }
        block_bytes_ = block_bytes;
        plain_off_ = -1;
        eof_ = false;
        stop_ = false;
        if (threaded) reader_ = std::thread([this] { pump(); });
        return true;
    }
    void close() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (reader_.joinable()) reader_.join();
        if (pargz_) {
            par_reader_.close();
            gz_open_files().fetch_sub(1);
        }
        pargz_ = false;
        if (gz_) gzclose(gz_);
        if (fp_ && fp_ != stdin) fclose(fp_);
        gz_ = nullptr;
        fp_ = nullptr;
        ready_.clear();
        free_.clear();
    }
    // Every block starts with kFrontPad free bytes so that the tail of the previous block (an incomplete
    // record) can be put in front of the new data without copying the block.
    static constexpr size_t kFrontPad = 1u << 20;
    // Next block: kFrontPad free bytes + raw data (empty vector at end of input).  Return a finished block
    // with recycle().
    std::vector<char> next() {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [this] { return !ready_.empty() || eof_; });
        if (ready_.empty()) return {};
        std::vector<char> b = std::move(ready_.front());
        ready_.erase(ready_.begin());
        cv_.notify_all();
        return b;
    }
    // after next() has returned an empty block: empty = clean end of input, otherwise what went wrong
    std::string error() {
        std::lock_guard<std::mutex> g(mu_);
        return error_;
    }
    void recycle(std::vector<char> &&b) {
        std::lock_guard<std::mutex> g(mu_);
        free_.push_back(std::move(b));
    }

    // plain files: number of threads that pread one block together (one thread copies ~3 GB/s out of the page cache)
    void set_readers(int n) { readers_ = n < 1 ? 1 : (n > kMaxReaders ? kMaxReaders : n); }
    // Next up-to-`want` bytes of the (decompressed) input into dst; fewer than `want` only at the end of the input.  On a
    // damaged input returns what could be read and sets `trouble`.  Only for sources opened with threaded = false.
    size_t read_into(char *dst, size_t want, std::string &trouble) { return fill(dst, want, trouble); }

  private:
    size_t fill(char *dst, size_t want, std::string &trouble) {
        size_t got = 0;
        while (got < want) {
            long r;
            if (gz_mode_ && use_zlib_) {
                r = (long)gzread(gz_, dst + got, (unsigned)std::min<size_t>(want - got, 1u << 30));
                if (r <= 0) {                      // 0 is also what a stream that ends too early gives: ask
                    int en = 0;
                    const char *msg = gzerror(gz_, &en);
                    if (en != Z_OK && en != Z_STREAM_END) trouble = std::string("gz: ") + msg;
                }
            } else if (gz_mode_ && bgzf_) {
                r = bgzf_reader_.read(reinterpret_cast<uint8_t *>(dst + got), want - got);
                if (r == -2) {                     // a member that is not BGZF: the serial decoder takes over from there
                    bgzf_ = false;
                    if (fseek(fp_, (long)bgzf_reader_.resume_offset(), SEEK_SET) != 0) trouble = "gz: cannot seek";
                    else {
                        inflater_.open(fp_, 4u << 20, true);
                        continue;
                    }
                }
                if (r == -1) trouble = bgzf_reader_.error();
            } else if (gz_mode_ && pargz_) {
                r = par_reader_.read(reinterpret_cast<uint8_t *>(dst + got), want - got);
                if (r < 0) trouble = par_reader_.error();
            } else if (gz_mode_) {
                r = inflater_.read(reinterpret_cast<uint8_t *>(dst + got), want - got);
                if (r < 0) trouble = inflater_.error();
            } else if (fp_ != stdin && want >= (4u << 20) && plain_off_ != -2) {
                r = parallel_pread(dst + got, want - got);     // regular file: several readers per block (never mixed with fread)
                if (r < 0) trouble = io_error_;
            } else {
                r = (long)fread(dst + got, 1, want - got, fp_);
                if (r == 0 && ferror(fp_)) trouble = "read failed";
            }
            if (r <= 0) break;
            got += (size_t)r;
        }
        // the decoders read through stdio: a failed read looks like the end of the input to them (and a gzip stream that ends
        // at a member border is a complete stream) -- the stream's error flag tells the two apart
        if (trouble.empty() && fp_ && ferror(fp_)) trouble = std::string("read failed: ") + strerror(errno ? errno : EIO);
        return got;
    }
    void pump() {
        for (;;) {
            std::vector<char> b;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return stop_ || ready_.size() < 3; });
                if (stop_) return;
                if (!free_.empty()) {
                    b = std::move(free_.back());
                    free_.pop_back();
                }
            }
            b.resize(kFrontPad + block_bytes_);
            std::string trouble;
            const size_t got = fill(b.data() + kFrontPad, block_bytes_, trouble);
            b.resize(kFrontPad + got);
            std::lock_guard<std::mutex> g(mu_);
            if (!trouble.empty()) error_ = trouble;
            if (got) ready_.push_back(std::move(b));
            if (got == 0 || !trouble.empty()) {
                eof_ = true;
                cv_.notify_all();
                return;
            }
            cv_.notify_all();
        }
    }
    // plain file: the block is filled by kReaders threads, each pread()ing its share (one thread copies ~4 GB/s out of
    // the page cache; storage likes several requests in flight, too).  Falls back to fread when the file cannot be pread.
    long parallel_pread(char *dst, size_t want) {
        const int fd = fileno(fp_);
        if (plain_off_ < 0) {
            plain_off_ = ftello(fp_);
            if (plain_off_ < 0 || pread(fd, dst, 0, 0) != 0) plain_off_ = -2;
        }
        if (plain_off_ == -2) {                                           // nothing has been pread yet: fread from here on
            const size_t r = fread(dst, 1, want, fp_);
            if (r == 0 && ferror(fp_)) {
                io_error_ = std::string("read failed: ") + strerror(errno ? errno : EIO);
                return -1;
            }
            return (long)r;
        }
        const int kReaders = (int)std::max<size_t>(1, std::min<size_t>((size_t)readers_, want >> 20));     // at least 1 MB per reader
        const size_t share = (want / kReaders + 4095) & ~(size_t)4095;
        size_t got[kMaxReaders] = {0};
        int io_errno[kMaxReaders] = {0};             // a failed read is an error, not the end of the file
        auto work = [&](int t) {
            const size_t from = std::min(want, share * (size_t)t), to = std::min(want, from + share);
            size_t done = 0;
            while (from + done < to) {
                const ssize_t r = pread(fd, dst + from + done, to - from - done, plain_off_ + (off_t)(from + done));
                if (r < 0 && errno == EINTR) continue;
                if (r < 0) io_errno[t] = errno ? errno : EIO;
                if (r <= 0) break;
                done += (size_t)r;
            }
            got[t] = done;
        };
        // a persistent pool: spawning the readers anew for every 16-MB block costs as much as the copy itself
        if (!pread_pool_ || pread_pool_->size() != readers_) pread_pool_.reset(new WorkerPool(readers_));
        pread_pool_->run([&](int t) { if (t < kReaders) work(t); });
        size_t total = 0;                            // contiguous bytes from the start: a short share ends the data
        for (int t = 0; t < kReaders; ++t) {
            const size_t from = std::min(want, share * (size_t)t), to = std::min(want, from + share);
            total += got[t];
            if (got[t] < to - from) break;
        }
        plain_off_ += (off_t)total;
        for (int t = 0; t < kReaders; ++t)
            if (io_errno[t]) {
                io_error_ = std::string("read failed: ") + strerror(io_errno[t]);
                return -1;
            }
        return (long)total;
    }
    // threads that inflate ONE ordinary .gz file (HAST_GZ_THREADS; 1 = the serial decoder).  Default: an eighth of the
    // machine, at most 8 -- several files are open at once, and on the 1-GPU share of an MI355X host two files x 8 threads
    // was the fastest setting (one file alone: 9.3 GB/s with 16 threads, 6.7 with 8, 1.4 serial; profiles/r02p_gz_e2e.log)
    static int gz_threads() {
        if (const char *e = getenv("HAST_GZ_THREADS")) return std::max(1, atoi(e));
        const unsigned hw = std::thread::hardware_concurrency();
        return (int)std::min(8u, std::max(2u, hw / 8));
    }
    // inflate threads over all ordinary .gz files open at once (HAST_GZ_BUDGET)
    static int gz_budget() {
        if (const char *e = getenv("HAST_GZ_BUDGET")) return std::max(1, atoi(e));
        // what the container may really use (cgroup v2 cpu.max: a 1-GPU share of an MI355X host shows 256 hardware threads and
        // grants 16 cores), else 16
        static const int quota = [] {
            long q = 0, per = 0;
            if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
                if (fscanf(f, "%ld %ld", &q, &per) != 2) q = per = 0;
                fclose(f);
            }
            return (q > 0 && per > 0) ? (int)std::max(2L, std::min(64L, q / per)) : 16;
        }();
        return quota;
    }
    static std::atomic<int> &gz_open_files() {
        static std::atomic<int> n{0};
        return n;
    }
    static constexpr int kMaxReaders = 32;
    int readers_ = 4;
    std::unique_ptr<WorkerPool> pread_pool_;
    off_t plain_off_ = -1;
    bool gz_mode_ = false, use_zlib_ = false, bgzf_ = false, pargz_ = false;
    BgzfReader bgzf_reader_;
    ParGzReader par_reader_;
    gzFile gz_ = nullptr;
    GzInflater inflater_;
    std::string error_;
    std::string io_error_;       // a failed pread of a plain file
    FILE *fp_ = nullptr;
    size_t block_bytes_ = 0;
    std::thread reader_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::vector<char>> ready_, free_;
    bool eof_ = false, stop_ = false;
};

}  // namespace hast

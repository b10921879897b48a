// unshared_kmers -- MI355X replacement for the compute of the reference's stage 00
// (00.build_unshare_kmers_by_jellyfish/build_unshared_kmers.sh, cited below as s00:N): from the parents' read files to
// paternal.unique.filter.mer / maternal.unique.filter.mer in the working directory.
//
// Same options as the script (s00:6-38,57-118), same argument checks (s00:141-158,166-185), same final products:
// the two .mer files (one upper-case canonical k-mer per line; here in sorted order, the reference's order is the
// third-party counter's hash order) and, with --auto_bounds, {maternal,paternal}.histo and
// {maternal,paternal}.bounds.txt (analysis_kmercount.sh:7-13, find_bounds.awk).  The script's intermediate files
// (*.jf, *.mer.fa, *.mer.filter.fa, *.mer.unique.fa, step_NN_done markers) have no counterpart: one count table in
// HBM holds both parents' counts and the products are read out of it (include/hast.h, hast_kc_*).
//
// Extra options: --device N (repeat it, or --devices a,b,c, to split the key space over several GPUs), --table-gb X (size of the count table per GPU; default: from the input size, at most 85 % of the free HBM), --slices S (process
// the key space in S passes over the input; doubled automatically when the table overflows), --save-table FILE (the
// two sets as a binary stage-01 table for `classify --load-table`), --stats.
// Exit status: 0 ok / usage; 1 bad arguments, missing or malformed input (the script: exit 1); 4 GPU trouble.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "../../include/hast.h"
#include "ingest.h"
#include "seqstream.h"

namespace {

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void usage(FILE *f) {
    fputs("Usage: unshared_kmers [options]\n"
          "  Parent-specific k-mer sets from paternal and maternal short reads, counted on the GPU.\n"
          "  Writes paternal.unique.filter.mer and maternal.unique.filter.mer into the current directory.\n"
          "    --paternal FILE   paternal reads, FASTA or FASTQ, gzip if the name ends in .gz (repeatable)\n"
          "    --maternal FILE   maternal reads (repeatable); gz and plain files cannot be mixed for one parent\n"
          "    --mer K           k-mer size, 11..32 (default 21)\n"
          "    --thread N        host threads for reading and parsing (default 8)\n"
          "    --memory G        accepted for compatibility (the table size is --table-gb)\n"
          "    --m-lower N / --m-upper N   keep maternal k-mers seen N..N times (default 9 / 33)\n"
          "    --p-lower N / --p-upper N   the same for paternal k-mers (default 9 / 33)\n"
          "    --auto_bounds     derive the four bounds from the count histograms (also writes *.histo, *.bounds.txt)\n"
          "    --device N (repeatable) / --devices a,b,c   GPUs; the k-mer space is split between them\n"
          "    --table-gb X  --slices S  --save-table FILE  --stats\n",
          f);
}

bool ends_gz(const std::string &s) { return s.size() >= 3 && s.compare(s.size() - 3, 3, ".gz") == 0; }   // s00:169

struct Options {
    long mer = 21, cpu = 8, memory = 10, lower[2] = {9, 9}, upper[2] = {33, 33};     // s00:44-55; [0] paternal, [1] maternal
    std::vector<std::string> files[2];
    bool auto_bounds = false, stats = false;
    std::vector<int> devices;        // --device N (repeatable) or --devices a,b,c; default: device 0
    double table_gb = 0;
    long slices = 1;
    std::string save_table;
};

// ---- ingest: files -> byte stream of bases -> GPU ---------------------------------------------------------------
// One count table per GPU.  With several GPUs the KEY SPACE is split between them (device d of D owns the slices
// d, d+D, ... of the minimizer hash): every GPU sees every chunk of bases and counts only its own k-mers, so there is no
// exchange between the GPUs at all; histograms add up and selections concatenate on the host.
struct Gpu {
    std::vector<hast_kc *> all;
    hast_kc *kc = nullptr;           // all[0]: also sorts and formats the output
    std::vector<std::unique_ptr<std::mutex>> dev_mu;
    std::mutex mu;
    std::string error;               // first failure of a submit
    bool ok() {
        std::lock_guard<std::mutex> g(mu);
        return error.empty();
    }
    void fail(const char *what) {
        std::lock_guard<std::mutex> g(mu);
        if (error.empty()) error = what;
    }
    void destroy() {
        for (hast_kc *k : all) hast_kc_destroy(k);
        all.clear();
        kc = nullptr;
    }
};

// collects the parser's output into chunks; a full chunk goes to the GPU and its last K-1 bytes open the next one, so
// that the windows across the cut are counted exactly once
class ChunkSink {
  public:
    ChunkSink(Gpu &gpu, int parent, int k) : gpu_(gpu), parent_(parent), keep_((size_t)k - 1) { buf_.reserve(kChunk + 64); }
    void append(const char *p, size_t n) {
        bases_ += n;
        while (n) {
            const size_t room = kChunk - buf_.size();
            const size_t take = std::min(room, n);
            buf_.insert(buf_.end(), p, p + take);
            p += take;
            n -= take;
            if (buf_.size() >= kChunk) flush(false);
        }
    }
    void separator() {
        buf_.push_back('\n');
        if (buf_.size() >= kChunk) flush(false);
    }
    void flush(bool last) {
        if (buf_.size() > fresh_from_ && gpu_.ok())
            for (size_t d = 0; d < gpu_.all.size(); ++d) {
                std::lock_guard<std::mutex> g(*gpu_.dev_mu[d]);
                if (hast_kc_count(gpu_.all[d], parent_, reinterpret_cast<const uint8_t *>(buf_.data()), buf_.size()) != HAST_OK) {
                    gpu_.fail(hast_last_error());
                    break;
                }
            }
        if (last) {
            buf_.clear();
            fresh_from_ = 0;
            return;
        }
        const size_t keep = std::min(keep_, buf_.size());
        if (keep) memmove(buf_.data(), buf_.data() + buf_.size() - keep, keep);
        buf_.resize(keep);
        fresh_from_ = keep;          // nothing new yet: a chunk that only holds the carried bytes is not sent again
    }
    size_t bases() const { return bases_; }

  private:
    static constexpr size_t kChunk = 32u << 20;
    Gpu &gpu_;
    int parent_;
    size_t keep_, fresh_from_ = 0, bases_ = 0;
    std::vector<char> buf_;
};

struct IngestResult {
    std::string error;
    size_t bases = 0, records = 0, bytes = 0;
    bool clean_end = true;           // the stream ended exactly between two records
    char first = 0;                  // its first byte
};

// one input stream of the counter: the files one after the other, either as separate inputs (plain) or as one
// concatenated stream (gz: `zcat files | ...`, s00:187-188)
void ingest_stream(Gpu &gpu, int parent, int k, const std::vector<std::string> &paths, bool concatenated, IngestResult &res) {
    ChunkSink sink(gpu, parent, k);
    hast::SeqParser<ChunkSink> parser(sink);
    for (size_t i = 0; i < paths.size() && res.error.empty(); ++i) {
        hast::BlockSource src;
        if (!src.open(paths[i], 16u << 20)) {
            res.error = "cannot open " + paths[i];
            break;
        }
        for (;;) {
            std::vector<char> b = src.next();
            if (b.empty()) {
                if (!src.error().empty()) res.error = paths[i] + ": " + src.error();
                break;
            }
            const size_t n = b.size() - hast::BlockSource::kFrontPad;
            res.bytes += n;
            if (!parser.feed(b.data() + hast::BlockSource::kFrontPad, n)) res.error = paths[i] + ": " + parser.error();
            src.recycle(std::move(b));
            if (!res.error.empty() || !gpu.ok()) break;
        }
        if (res.error.empty() && (!concatenated || i + 1 == paths.size())) {
            res.clean_end = parser.at_record_boundary();
            res.first = parser.first_byte();
            if (!parser.finish()) res.error = paths[i] + ": " + parser.error();
        }
    }
    sink.flush(true);
    res.bases = sink.bases();
    res.records = parser.records();
}

// Both parents' files, read and parsed by up to --thread workers at once.  Plain files are independent inputs of the
// counter (s00:190).  The gz files of a parent are ONE concatenated stream in the reference (s00:187-188); they are
// still read in parallel, which gives the same result whenever every file ends exactly between two records and all
// start with the same byte -- if not (gz_in_order comes back true), the caller starts over and reads them in order.
struct ParentTotals { size_t bases = 0, records = 0, bytes = 0; };
bool ingest_all(Gpu &gpu, const Options &o, bool &gz_in_order, std::string &err, ParentTotals tot[2]) {
    struct Job { int parent; std::vector<std::string> paths; bool concatenated, check; IngestResult res; };
    std::vector<Job> jobs;
    for (int p = 1; p >= 0; --p) {                                              // maternal first, as the script does
        std::vector<std::string> order(o.files[p].rbegin(), o.files[p].rend());   // s00:105,109: each new file is put in front
        const bool gz = ends_gz(order[0]);
        if (gz && gz_in_order) jobs.push_back({p, order, true, false, {}});
        else
            for (size_t i = 0; i < order.size(); ++i) jobs.push_back({p, {order[i]}, false, gz && order.size() > 1, {}});
    }
    std::atomic<size_t> next{0};
    const int nt = (int)std::max<long>(1, std::min<long>(o.cpu, (long)jobs.size()));
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&] {
            for (size_t i; (i = next.fetch_add(1)) < jobs.size();) ingest_stream(gpu, jobs[i].parent, (int)o.mer, jobs[i].paths, jobs[i].concatenated, jobs[i].res);
        });
    for (auto &t : th) t.join();
    bool redo = false;
    for (size_t i = 0; i < jobs.size(); ++i) {
        const Job &j = jobs[i];
        if (j.check) {
            const bool last = i + 1 == jobs.size() || jobs[i + 1].parent != j.parent;
            const bool first = i == 0 || jobs[i - 1].parent != j.parent;
            if ((!last && !j.res.clean_end) || (!first && j.res.first != jobs[i - 1].res.first && j.res.first && jobs[i - 1].res.first)) redo = true;
            if (!j.res.error.empty() && !first) redo = true;                    // may parse differently as part of the whole stream
        }
    }
    if (redo && !gz_in_order) {
        gz_in_order = true;
        return false;
    }
    for (const auto &j : jobs) {
        if (!j.res.error.empty() && err.empty()) err = j.res.error;
        tot[j.parent].bases += j.res.bases;
        tot[j.parent].records += j.res.records;
        tot[j.parent].bytes += j.res.bytes;
    }
    if (err.empty() && !gpu.ok()) err = gpu.error;
    return err.empty();
}

bool write_histo(const char *path, const std::vector<uint64_t> &h) {
    FILE *f = fopen(path, "w");
    if (!f) return false;
    for (unsigned c = 1; c <= HAST_KC_HISTO_HIGH + 1; ++c)
        if (h[c]) fprintf(f, "%u %llu\n", c, (unsigned long long)h[c]);       // `jellyfish histo` rows
    return fclose(f) == 0;
}

}  // namespace

int main(int argc, char **argv) {
    Options o;
    if (argc == 1) {                                                            // s00:57-60
        usage(stdout);
        return 0;
    }
    printf("CMD :");
    for (int i = 0; i < argc; ++i) printf(" %s", argv[i]);
    printf("\n");
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "-h" || a == "--help") { usage(stdout); return 0; }
        else if (a == "--memory") o.memory = atol(val());
        else if (a == "--thread") o.cpu = atol(val());
        else if (a == "--m-lower") o.lower[1] = atol(val());
        else if (a == "--m-upper") o.upper[1] = atol(val());
        else if (a == "--p-lower") o.lower[0] = atol(val());
        else if (a == "--p-upper") o.upper[0] = atol(val());
        else if (a == "--mer") o.mer = atol(val());
        else if (a == "--auto_bounds") o.auto_bounds = true;
        else if (a == "--paternal") o.files[0].push_back(val());
        else if (a == "--maternal") o.files[1].push_back(val());
        else if (a == "--device") o.devices.push_back(atoi(val()));
        else if (a == "--devices") {
            for (const char *q = val(); *q;) {
                o.devices.push_back(atoi(q));
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
        }
        else if (a == "--table-gb") o.table_gb = atof(val());
        else if (a == "--slices") o.slices = atol(val());
        else if (a == "--save-table") o.save_table = val();
        else if (a == "--stats") o.stats = true;
        else {                                                                  // s00:113-116: message, then a bare `exit`
            printf("unknown option \"%s\"\n", a.c_str());
            return 0;
        }
    }
    if (o.memory < 1 || o.cpu < 1 || o.files[0].empty() || o.files[1].empty() || o.mer < 11 || o.lower[1] < 1 ||
        o.upper[1] > 100000000 || o.lower[0] < 1 || o.upper[0] > 100000000 || o.slices < 1 || o.slices > 4096) {   // s00:141-152
        printf("ERROR: invalid arguments\n");
        return 1;
    }
    if (o.mer > 32) {
        printf("ERROR: --mer %ld: this build handles k-mers up to 32 bases\n", o.mer);
        return 1;
    }
    for (int p = 1; p >= 0; --p)                                                // s00:153-158
        for (const auto &f : o.files[p])
            if (access(f.c_str(), F_OK) != 0) {
                printf("ERROR: input file \"%s\" does not exist\n", f.c_str());
                return 1;
            }
    for (int p = 1; p >= 0; --p)                                                // s00:166-185, 197-216
        for (const auto &f : o.files[p])
            if (ends_gz(f) != ends_gz(o.files[p][0])) {
                printf("ERROR: gz and plain inputs mixed for one parent\n");
                return 1;
            }
    const char *pname[2] = {"paternal", "maternal"};
    const double t_start = now();
    double t_ingest = 0;

    // Table size when not given: from the input -- an upper bound on its windows from the file sizes (about half the bytes of a
    // FASTQ are bases, a gz file holds at most ~3 bases per byte) and a slot (16 B) per window: every window a different k-mer would
    // fill the table to the brim, sequencing data (30 x coverage, 15 % error k-mers) fills a quarter to a third of it.  Round 4 took
    // twice that, and the record buffers "what is left of the device": 100 GB for a 20-Mbp job, which a box that had just freed
    // them took 6 s to hand out (profiles/round4_measure.txt).  Too small a guess only costs a restart with more slices; capped by
    // the library at 85 % of the free HBM.
    size_t table_bytes = (size_t)(o.table_gb * (double)(1ull << 30));
    double windows = 0;
    {
        for (int p = 0; p < 2; ++p)
            for (const auto &f : o.files[p]) {
                struct stat sb;
                if (stat(f.c_str(), &sb) != 0) continue;
                bool fasta = false;
                if (!ends_gz(f)) {
                    if (FILE *fp = fopen(f.c_str(), "rb")) {
                        fasta = fgetc(fp) == '>';
                        fclose(fp);
                    }
                }
                windows += ends_gz(f) ? 3.0 * (double)sb.st_size : fasta ? (double)sb.st_size : 0.55 * (double)sb.st_size;
            }
        if (table_bytes == 0) table_bytes = (size_t)std::max(256.0 * (1 << 20), windows * 16.0);
    }
    if (o.devices.empty()) o.devices.push_back(0);
    const long n_dev = (long)o.devices.size();
    Gpu gpu;
    for (int dev : o.devices) {
        hast_kc *k = nullptr;
        // --table-gb is per GPU; the automatic size is for the whole key space, i.e. divided between the GPUs
        const size_t per_dev = o.table_gb > 0 ? table_bytes : table_bytes / (size_t)n_dev + 1;
        if (hast_kc_create_ex(dev, (int)o.mer, per_dev, (uint64_t)(windows / (double)n_dev * 1.1) + 1, &k) != HAST_OK) {
            fprintf(stderr, "unshared_kmers: device %d: %s\n", dev, hast_last_error());
            gpu.destroy();
            return 4;
        }
        gpu.all.push_back(k);
        gpu.dev_mu.emplace_back(new std::mutex());
    }
    gpu.kc = gpu.all[0];
    const double t_table = now();
    auto gpu_fail = [&](const char *what) {
        fprintf(stderr, "unshared_kmers: %s: %s\n", what, hast_last_error());
        gpu.destroy();
        return 4;
    };

    std::vector<uint64_t> histo[2];
    long slices = o.slices;
    bool gz_in_order = false;
    uint64_t stats_sum[6] = {0, 0, 0, 0, 0, 0};
    size_t bases[2] = {0, 0}, records[2] = {0, 0}, bytes[2] = {0, 0};
    // One sweep = every slice of the key space: count both parents, then take what this sweep is for.
    // Returns 0 ok, 1 input error, 4 GPU error, -1 table full (caller retries with more slices), -2 read the gz files in order.
    // the sets of the tables as they stand, appended to the first GPU's selection
    auto select_all = [&]() -> bool {
        for (hast_kc *k : gpu.all) {
            for (int p = 0; p < 2; ++p) {
                // a bound pair that selects nothing (upper < lower, e.g. from an empty histogram) is an empty set
                if (o.upper[p] < o.lower[p] || o.upper[p] < 1) continue;
                if (hast_kc_select(k, p, (uint32_t)o.lower[p], (uint32_t)std::min<long>(o.upper[p], 0xFFFFFFFFl), nullptr) != HAST_OK) return false;
            }
            if (k != gpu.kc && hast_kc_selection_adopt(gpu.kc, k) != HAST_OK) return false;
        }
        return true;
    };
    auto sweep = [&](bool take_histo, bool take_sets) -> int {
        for (int p = 0; p < 2; ++p) {
            if (take_histo) histo[p].assign(HAST_KC_HISTO_HIGH + 2, 0);
            bases[p] = records[p] = bytes[p] = 0;
        }
        for (auto &x : stats_sum) x = 0;
        if (take_sets)                                                          // a sweep that starts over starts from nothing
            for (hast_kc *k : gpu.all)
                if (hast_kc_selection_clear(k) != HAST_OK) return 4;
        for (long s = 0; s < slices; ++s) {
            for (long d = 0; d < n_dev; ++d)
                if (hast_kc_set_slice(gpu.all[d], (uint32_t)(s * n_dev + d), (uint32_t)(slices * n_dev)) != HAST_OK) return 4;
            auto sync_all = [&]() -> hast_status {                              // table-full on any device wins
                hast_status worst = HAST_OK;
                for (hast_kc *k : gpu.all) {
                    const hast_status st = hast_kc_sync(k);
                    if (st == HAST_ERR_TABLE_FULL || (st != HAST_OK && worst == HAST_OK)) worst = st;
                }
                return worst;
            };
            {
                std::string err;
                ParentTotals tot[2];
                const bool was_in_order = gz_in_order;
                const double t_in = now();
                const bool ok = ingest_all(gpu, o, gz_in_order, err, tot);
                t_ingest += now() - t_in;
                if (!ok) {
                    if (gz_in_order && !was_in_order) return -2;               // a gz file ends inside a record: read them in order
                    const bool gpu_side = !gpu.error.empty();
                    if (gpu_side && sync_all() == HAST_ERR_TABLE_FULL) return -1;
                    fprintf(gpu_side ? stderr : stdout, "ERROR: %s\n", err.c_str());
                    return gpu_side ? 4 : 1;
                }
                for (int p = 0; p < 2; ++p) {
                    bases[p] = tot[p].bases;
                    records[p] = tot[p].records;
                    bytes[p] = tot[p].bytes;
                }
            }
            const hast_status st = sync_all();
            if (st == HAST_ERR_TABLE_FULL) return -1;
            if (st != HAST_OK) return 4;
            stats_sum[3] = 0;
            for (hast_kc *k : gpu.all) {
                uint64_t stt[6];
                if (hast_kc_stats(k, stt) != HAST_OK) return 4;
                for (int i = 0; i < 6; ++i) stats_sum[i] += stt[i];
                if (take_histo)
                    for (int p = 0; p < 2; ++p)
                        if (hast_kc_histo(k, p, histo[p].data()) != HAST_OK) return 4;
            }
            if (take_sets && !select_all()) return 4;
        }
        return 0;
    };
    auto run = [&](bool take_histo, bool take_sets) -> int {
        for (;;) {
            gpu.error.clear();
            const int rc = sweep(take_histo, take_sets);
            if (rc == -2) {
                fprintf(stderr, "a gz input ends inside a record: reading each parent's gz files in order, as one stream\n");
                for (hast_kc *k : gpu.all) hast_kc_sync(k);
                continue;
            }
            if (rc != -1) return rc;
            if (slices >= 4096) {
                fprintf(stderr, "unshared_kmers: the count table is too small even with %ld slices\n", slices);
                return 4;
            }
            slices *= 2;
            fprintf(stderr, "count table full: starting over with %ld slices of the key space\n", slices);
        }
    };

    // Everything in one sweep when the bounds are known up front or the table holds the whole key space (1 slice:
    // histogram, bounds and sets all come out of the resident table); otherwise histograms first, sets in a second sweep.
    int rc;
    if (!o.auto_bounds) rc = run(false, true);
    else {
        rc = run(true, false);
        if (rc == 0) {
            for (int p = 1; p >= 0; --p) {                                      // analysis_kmercount.sh:7-13
                long b[4];
                hast_kc_find_bounds(histo[p].data(), b);
                o.lower[p] = b[2];
                o.upper[p] = b[3];
                const std::string hp = std::string(pname[p]) + ".histo", bp = std::string(pname[p]) + ".bounds.txt";
                FILE *f = write_histo(hp.c_str(), histo[p]) ? fopen(bp.c_str(), "w") : nullptr;
                if (!f) {
                    printf("ERROR: cannot write %s / %s\n", hp.c_str(), bp.c_str());
                    gpu.destroy();
                    return 1;
                }
                fprintf(f, "MIN_INDEX=%ld\nMAX_INDEX=%ld\nLOWER_INDEX=%ld\nUPPER_INDEX=%ld\n", b[0], b[1], b[2], b[3]);   // find_bounds.awk:31
                fclose(f);
            }
            if (slices == 1) {                                                  // the tables still hold everything
                if (!select_all()) rc = 4;
            } else rc = run(false, true);
        }
    }
    if (rc == 4) return gpu_fail("counting");
    if (rc != 0) {
        gpu.destroy();
        return rc;
    }
    const double t_count = now();
    printf("bounds used for maternal: [%ld, %ld]\n", o.lower[1], o.upper[1]);   // s00:254-255
    printf("bounds used for paternal: [%ld, %ld]\n", o.lower[0], o.upper[0]);

    for (hast_kc *k : gpu.all)
        if (hast_kc_release_table(k) != HAST_OK) return gpu_fail("releasing the table");
    size_t n_sel[2] = {0, 0};
    for (int p = 0; p < 2; ++p) {
        if (hast_kc_selection_sort(gpu.kc, p, &n_sel[p]) != HAST_OK) return gpu_fail("sorting the selection");
        const std::string path = std::string(pname[p]) + ".unique.filter.mer";
        FILE *f = fopen(path.c_str(), "w");
        if (!f) {
            printf("ERROR: cannot write %s\n", path.c_str());
            gpu.destroy();
            return 1;
        }
        const size_t rows = 4u << 20, width = (size_t)o.mer + 1;
        std::vector<char> text(std::min(rows, std::max<size_t>(n_sel[p], 1)) * width);
        for (size_t at = 0; at < n_sel[p]; at += rows) {
            const size_t n = std::min(rows, n_sel[p] - at);
            if (hast_kc_selection_text(gpu.kc, p, at, n, text.data()) != HAST_OK) {
                fclose(f);
                return gpu_fail("formatting the selection");
            }
            if (fwrite(text.data(), 1, n * width, f) != n * width) {
                printf("ERROR: short write to %s\n", path.c_str());
                fclose(f);
                gpu.destroy();
                return 1;
            }
        }
        if (fclose(f) != 0) {
            printf("ERROR: cannot write %s\n", path.c_str());
            gpu.destroy();
            return 1;
        }
    }
    if (!o.save_table.empty()) {                                                // hap 0 = paternal, hap 1 = maternal (classify -p / -m)
        hast_ctx *ctx = nullptr;
        if (hast_ctx_create(o.devices[0], (int)o.mer, &ctx) != HAST_OK) return gpu_fail("--save-table");
        bool ok = hast_table_reserve(ctx, n_sel[0] + n_sel[1] + 64, 0.0) == HAST_OK;
        std::vector<uint64_t> keys;
        for (int p = 0; p < 2 && ok; ++p)
            for (size_t at = 0; at < n_sel[p] && ok; at += 8u << 20) {
                const size_t n = std::min<size_t>(8u << 20, n_sel[p] - at);
                keys.resize(n);
                ok = hast_kc_selection_keys(gpu.kc, p, at, n, keys.data()) == HAST_OK && hast_table_insert_keys(ctx, p, keys.data(), n) == HAST_OK;
            }
        ok = ok && hast_table_save(ctx, o.save_table.c_str()) == HAST_OK;
        if (!ok) fprintf(stderr, "unshared_kmers: --save-table: %s\n", hast_last_error());
        hast_ctx_destroy(ctx);
        if (!ok) {
            gpu.destroy();
            return 4;
        }
    }
    printf("paternal-unique k-mers kept: %zu (paternal.unique.filter.mer)\n", n_sel[0]);      // s00:300-303 (wc -l of the products)
    printf("maternal-unique k-mers kept: %zu (maternal.unique.filter.mer)\n", n_sel[1]);
    const double t_end = now();
    if (o.stats) {
        fprintf(stderr, "[stats] K=%ld gpus=%ld slices=%ld table_slots=%llu keys_in_table=%llu\n", o.mer, n_dev, slices * n_dev,
                (unsigned long long)stats_sum[3], (unsigned long long)stats_sum[2]);
        for (int p = 0; p < 2; ++p)
            fprintf(stderr, "[stats] %s: %zu input bytes, %zu records, %zu bases, %llu k-mers counted, %llu distinct, %zu selected\n", pname[p],
                    bytes[p], records[p], bases[p], (unsigned long long)stats_sum[4 + p], (unsigned long long)stats_sum[p], n_sel[p]);
        fprintf(stderr, "[stats] table %.3f s, read+parse+count %.3f s, table passes %.3f s, output %.3f s, total %.3f s\n", t_table - t_start, t_ingest,
                t_count - t_table - t_ingest, t_end - t_count, t_end - t_start);
    }
    gpu.destroy();
    return 0;
}

// classify_main.cpp -- the drop-in `classify` executable for HAST stage 01 on MI355X.
//
// Same command line, same stdout rows and same stderr skeleton as the reference program
// (/root/reference/01.classify_stlfr_reads/classify.cpp:373-450), so
// 01.classify_stlfr_reads/classify_stlfr_reads.sh:148-149 runs unchanged against it.  All k-mer
// work (table build, adaptor scrub, read classification) happens on the GPU through the C ABI of
// include/hast.h; this file only does what the reference does on the host around it: flag parsing,
// FASTQ framing, barcode names -> dense ids, getHap + printing.
//
// Deliberate, documented deviations (all on inputs the reference does not survive either):
//   * unopenable k-mer/read file: error + exit 2 (the reference loops forever, classify.cpp:41);
//   * ragged k-mer line / read shorter than K: error + exit 3 (the reference assert-aborts,
//     kmer.h:154,171);
//   * K must be in [1,32] like the reference's correct range (it is silently wrong above 32).
// Additive flags: --device N (GPU ordinal, default 0), --devices A,B,... (several GPUs of this node: the table is built
// on the first one and copied to the others over xGMI; the BLOCKS of every input file are dealt to the GPUs in turn and framed
// there -- the reference spreads the reads of one file over all its workers, classify.cpp:211-219 -- (HAST_DEAL=files: whole
// files in turn, as round 2 did; --host-parse: host-framed batches in turn); the per-barcode counters are
// summed with ONE RCCL all-reduce at the end -- the thread merge of classify.cpp:226-229,276-277 across GPUs; integer sums,
// so stdout is byte-identical to a single-GPU run), --block-mb N (ingest block size), --batch-reads N
// (approximate records per GPU batch, for tests), --initial-barcodes N, --stats (timings on stderr), --host-parse (frame the
// FASTQ records on the host as round 1 did; default: raw file bytes go to the GPU and are framed there, hast_fq_*).
// -t/--thread N is honoured as the number of host parser threads.
#include <getopt.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <deque>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/hast.h"
#include "ingest.h"
#include "quartering.h"

namespace {

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void logtime() {                                  // classify.cpp:17-21
    time_t now = time(0);
    fprintf(stderr, "%s\n", ctime(&now));
}

void print_usage() {                              // same flags as the reference (classify.cpp:375-387); stderr is free-form
    fputs("\nclassify (MI355X) -- per-barcode haplotype votes for stLFR reads\n\n"
          "  classify --hap0 PATERNAL.mer --hap1 MATERNAL.mer --read READS.fq[.gz] [--read ...] [options]\n\n"
          "  -p, --hap0 FILE       parent-0 specific k-mers, one per line (K = length of the first line, K <= 32)\n"
          "  -m, --hap1 FILE       parent-1 specific k-mers\n"
          "  -r, --read FILE       child reads, 4-line FASTQ; gzip if the name ends in .gz; may be repeated\n"
          "  -t, --thread N        host parser threads (default 8)\n"
          "  -w, --weight0 F       weight of hap0 in the call (default 1.0)\n"
          "  -u, --weight1 F       weight of hap1 in the call (default 1.0)\n"
          "  -f, --adaptor_f SEQ   forward adaptor whose k-mers are removed from both sets\n"
          "  -q, --adaptor_r SEQ   reverse adaptor whose k-mers are removed from both sets\n"
          "      --device N        GPU ordinal (default 0)\n"
          "      --devices A,B,..  classify on several GPUs of this node (reads are dealt out, counts summed over RCCL)\n"
          "      --save-table FILE write the built k-mer table (after the adaptor scrub) as a binary key set\n"
          "      --load-table FILE use such a file instead of --hap0/--hap1\n"
          "      --stats           timings and set sizes on stderr\n"
          "      --stats-json FILE the same as one JSON object (- = stderr)\n"
          "      --phase-reads     also do the wrapper's steps 10-11: write the three barcode lists and route every record of every\n"
          "                        input to <name>.{paternal,maternal,homozygous,nobarcode}.fastq (HAST_PHASE_READS=1)\n"
          "      --route MODE      device (default): records routed on the GPU; host: parsed again by host threads\n"
          "      --inflate MODE    device (default) | host | zlib: who inflates .gz inputs (HAST_INFLATE)\n"
          "      --gz-ring-bytes N compressed bytes of a .gz input kept on the device at a time (default: whole files up to 2 GB;\n"
          "                        HAST_GZ_RING_BYTES)\n"
          "      --park-gb X       device memory of closed streams kept for reuse instead of freed (default 32; HAST_PARK_GB)\n"
          "      --name-cache N    barcodes the device-side dictionary holds (default 16M; HAST_NAME_CACHE)\n"
          "      --deal MODE       with --devices: blocks (default) | files (HAST_DEAL)\n"
          "  -h, --help            this text\n\n"
          "stdout: barcode <TAB> haplotype(0/1/-1) <TAB> hits_hap0 <TAB> hits_hap1, sorted by barcode\n\n",
          stderr);
}

[[noreturn]] void die(int code, const char *what) {
    fprintf(stderr, "classify: ERROR: %s", what);
    const char *e = hast_last_error();
    if (e && *e) fprintf(stderr, " (%s)", e);
    fputc('\n', stderr);
    // (no destructors: reader / inflate threads of the library may be at work, and nothing is left to save)
    fflush(stdout);
    fflush(stderr);
    _exit(code);
}
#define CK(call, what) do { if ((call) != HAST_OK) die(4, what); } while (0)
[[noreturn]] void die_output() {
    fprintf(stderr, "classify: ERROR: writing the result to stdout failed (%s)\n", strerror(errno));
    fflush(stderr);
    _exit(2);
}

// --stats: every "__stats_<section>__ key=value ..." line goes to stderr and is kept for --stats-json FILE, which writes the same
// numbers as ONE JSON object {"section": {"key": value, ...}, ...} (a section printed several times, e.g. one line per .gz input,
// becomes an array) -- SURVEY section 5's machine-readable summary.
struct StatLog {
    std::mutex mu;
    std::vector<std::string> lines;
};
StatLog &stat_log() {
    static StatLog *l = new StatLog();
    return *l;
}
void stat_line(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void stat_line(const char *fmt, ...) {
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    fputs(buf, stderr);
    std::string l(buf);
    while (!l.empty() && l.back() == '\n') l.pop_back();
    std::lock_guard<std::mutex> g(stat_log().mu);
    stat_log().lines.push_back(std::move(l));
}
bool write_stats_json(const std::string &path) {
    std::vector<std::string> lines;
    {
        std::lock_guard<std::mutex> g(stat_log().mu);
        lines = stat_log().lines;
    }
    auto quote = [](const std::string &v) {
        std::string o = "\"";
        for (char c : v) {
            if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back(c); }
            else if ((unsigned char)c < 0x20) { char t[8]; snprintf(t, sizeof(t), "\\u%04x", c); o += t; }
            else o.push_back(c);
        }
        return o + "\"";
    };
    auto is_number = [](const std::string &v) {
        if (v.empty()) return false;
        char *end = nullptr;
        (void)strtod(v.c_str(), &end);
        if (*end) return false;
        const char c0 = v[0] == '-' ? (v.size() > 1 ? v[1] : 'x') : v[0];        // (JSON numbers: no hex, inf, nan, leading '+' or '.')
        return c0 >= '0' && c0 <= '9' && v.find_first_of("xXnN") == std::string::npos && !(v.size() > 1 && v[0] == '0' && v[1] >= '0' && v[1] <= '9');
    };
    std::vector<std::pair<std::string, std::vector<std::string>>> sections;       // name -> one object per line, in order of appearance
    for (const std::string &l : lines) {
        size_t sp = l.find(' ');
        std::string name = l.substr(0, sp);
        while (!name.empty() && name.front() == '_') name.erase(name.begin());
        while (!name.empty() && name.back() == '_') name.pop_back();
        std::string obj = "{";
        bool first = true;
        while (sp != std::string::npos) {
            const size_t a = sp + 1, b = l.find(' ', a);
            const std::string tok = l.substr(a, b == std::string::npos ? std::string::npos : b - a);
            sp = b;
            const size_t eq = tok.find('=');
            if (eq == std::string::npos || eq == 0) continue;                     // (free text inside a line)
            const std::string k = tok.substr(0, eq), v = tok.substr(eq + 1);
            obj += (first ? "" : ", ") + quote(k) + ": " + (is_number(v) ? v : quote(v));
            first = false;
        }
        obj += "}";
        size_t i = 0;
        while (i < sections.size() && sections[i].first != name) ++i;
        if (i == sections.size()) sections.push_back({name, {}});
        sections[i].second.push_back(obj);
    }
    std::string js = "{";
    for (size_t i = 0; i < sections.size(); i++) {
        js += (i ? ", " : "") + quote(sections[i].first) + ": ";
        if (sections[i].second.size() == 1) js += sections[i].second[0];
        else {
            js += "[";
            for (size_t j = 0; j < sections[i].second.size(); j++) js += (j ? ", " : "") + sections[i].second[j];
            js += "]";
        }
    }
    js += "}\n";
    if (path == "-") return fputs(js.c_str(), stderr) >= 0;
    FILE *f = fopen(path.c_str(), "wb");
    return f && fwrite(js.data(), 1, js.size(), f) == js.size() && fclose(f) == 0;
}

// the whole input, front to back (also from a pipe, which has no size to ask for)
bool slurp(const std::string &path, std::vector<char> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    out.clear();
    size_t have = 0;
    for (;;) {
        if (out.size() - have < (1u << 20)) out.resize(std::max<size_t>(out.size() * 2, 4u << 20));
        const size_t n = fread(out.data() + have, 1, out.size() - have, f);
        have += n;
        if (n == 0) break;
    }
    const bool ok = !ferror(f);
    fclose(f);
    out.resize(have);
    return ok;
}

// the reference's startup self-test (TestAll, classify.cpp:341-367) on our host primitives
bool self_test() {
    size_t s, n;
    const char *h = "VSDSDS#XXX_xxx_s/1";
    hast_parse_barcode(h, strlen(h), &s, &n);
    if (std::string(h + s, n) != "XXX_xxx_s") return false;
    if (hast_canon_kmer("AGCTC", 5) != 0xD9 || hast_canon_kmer("GAGCT", 5) != 0xD9) return false;
    uint64_t km[2];
    if (hast_chop_read("GAGCTA", 6, 5, km) != 2 || km[0] != 0xD9 || km[1] != 0xD8) return false;
    char buf[8];
    hast_kmer_to_str(km[0], 5, buf);
    if (strcmp(buf, "AGCTC")) return false;
    hast_kmer_to_str(km[1], 5, buf);
    return strcmp(buf, "AGCTA") == 0;
}

struct Counts {                                    // host accumulators behind the device counters
    std::vector<uint64_t> c0, c1, neg;             // by device-dictionary id (one dictionary), by merged id (several), by host id (no device dictionary)
    std::vector<uint64_t> h0, h1, hneg;            // by host-dictionary id, for what a device dictionary left to the host
    size_t device_cap = 0;
};

// Who numbers the barcodes.  Round 6: the GPU's own dictionary (hast_names_create_dict) -- ids 0 .. limit-1 in the order in which its
// naming kernel meets new texts; the host's dictionary only names what the device leaves to it (texts longer than 15 bytes, and what
// arrives when every device id is out) in the range [host_base, ...) above.  One dictionary per GPU (contexts of one GPU share it): with
// several GPUs the dictionaries number independently and their counters are MERGED BY TEXT, on the GPUs: the texts another dictionary has
// learnt go through the FIRST dictionary's naming kernel (hast_names_merge: their ids there, new ones claimed), the counters of that
// dictionary's contexts are renumbered on their device (hast_counts_permute), then the one all-reduce runs over one id space.
struct Naming {
    bool device_dict = false;
    size_t host_base = 0;                          // ids of the host dictionary start here in the device counters
    std::vector<hast_names *> groups;              // the distinct dictionaries
    std::vector<int> group_of;                     // per context
    std::vector<std::vector<uint32_t>> perm;       // per dictionary but the first: its ids in the first one's numbering, as far as merged
    double merge_s = 0;
};

inline void add_into(std::vector<uint64_t> &dst, const std::vector<uint64_t> &src, size_t n) {
    if (dst.size() < n) dst.resize(n);
    for (size_t i = 0; i < n; i++) dst[i] += src[i];       // (64-bit on the device and here: nothing wraps)
}

void flush_counts(std::vector<hast_ctx *> &ctxs, Counts &acc, Naming &nm, hast::WorkerPool &, size_t n_host, size_t new_cap) {
    // fold what the devices have counted so far into the host sums, then (re)size the device arrays.  Several GPUs: ONE
    // all-reduce(sum,u64) over RCCL/xGMI leaves the totals on every device (collectBarcodes + data.Add, classify.cpp:226-229,277)
    hast_ctx *ctx = ctxs[0];
    if (acc.device_cap) {
        std::vector<uint64_t> a, b, c;
        auto read_range = [&](size_t first, size_t n, std::vector<uint64_t> &d0, std::vector<uint64_t> &d1, std::vector<uint64_t> &d2) {
            n = first < acc.device_cap ? std::min(n, acc.device_cap - first) : 0;
            a.assign(n, 0); b.assign(n, 0); c.assign(n, 0);
            if (n) CK(hast_counts_read_range(ctx, first, n, a.data(), b.data(), c.data()), "reading counters");
            add_into(d0, a, n); add_into(d1, b, n); add_into(d2, c, n);
        };
        if (nm.device_dict && nm.groups.size() > 1) {
            // several dictionaries: what each has learnt since the last merge into the first one's numbering, its contexts' counters with it
            const double t0 = now_s();
            nm.perm.resize(nm.groups.size());
            for (size_t g = 1; g < nm.groups.size(); g++) {
                size_t n_g = 0;
                CK(hast_names_count(nm.groups[g], &n_g), "asking a dictionary for its size");
                const size_t have = nm.perm[g].size();
                if (n_g > have) {
                    nm.perm[g].resize(n_g);
                    if (hast_names_merge(nm.groups[0], nm.groups[g], have, n_g - have, nm.perm[g].data() + have) != HAST_OK)
                        die(4, "merging the GPUs' barcode dictionaries (more barcodes than --name-cache allows?)");
                }
            }
            // (the first dictionary may have grown past what the counters were sized for -- they follow the ids of each dictionary's own
            // blocks: every context's counters move into arrays that hold the merged numbering)
            size_t n_merged = 0;
            CK(hast_names_count(nm.groups[0], &n_merged), "asking the dictionary for its size");
            const size_t need_cap = std::max(acc.device_cap, n_merged);
            for (size_t i = 0; i < ctxs.size(); i++) {
                const size_t g = (size_t)nm.group_of[i];
                if (g == 0 && need_cap == acc.device_cap) continue;
                const size_t n_perm = g ? std::min(nm.perm[g].size(), acc.device_cap) : 0;
                CK(hast_counts_permute(ctxs[i], g ? nm.perm[g].data() : nullptr, n_perm, need_cap), "renumbering the counters of a GPU");
            }
            acc.device_cap = need_cap;
            nm.merge_s += now_s() - t0;
        }
        if (ctxs.size() > 1) CK(hast_counts_allreduce(ctxs.data(), (int)ctxs.size()), "summing the counters of the GPUs");
        if (!nm.device_dict) read_range(0, n_host, acc.c0, acc.c1, acc.neg);      // (only the barcodes that exist: the counters are sized ahead of the dictionary)
        else {
            size_t n_dev = 0;
            CK(hast_names_count(nm.groups[0], &n_dev), "asking the dictionary for its size");
            read_range(0, n_dev, acc.c0, acc.c1, acc.neg);
            read_range(nm.host_base, n_host, acc.h0, acc.h1, acc.hneg);
        }
    }
    for (hast_ctx *c : ctxs) CK(hast_counts_resize(c, new_cap), "allocating counters");
    acc.device_cap = new_cap;
}

}  // namespace

int main(int argc, char **argv) {
    if (!self_test()) {
        fprintf(stderr, "classify: self-test failed\n");
        return 1;
    }
    static struct option long_options[] = {                       // classify.cpp:375-386 + additive
        {"hap0", required_argument, NULL, 'p'},      {"hap1", required_argument, NULL, 'm'},
        {"read", required_argument, NULL, 'r'},      {"thread", required_argument, NULL, 't'},
        {"weight0", required_argument, NULL, 'w'},   {"weight1", required_argument, NULL, 'u'},
        {"adaptor_f", required_argument, NULL, 'f'}, {"adaptor_r", required_argument, NULL, 'q'},
        {"help", no_argument, NULL, 'h'},            {"device", required_argument, NULL, 1001},
        {"batch-reads", required_argument, NULL, 1002}, {"stats", no_argument, NULL, 1003},
        {"block-mb", required_argument, NULL, 1004},    {"initial-barcodes", required_argument, NULL, 1005},
        {"save-table", required_argument, NULL, 1006},  {"load-table", required_argument, NULL, 1007},
        {"devices", required_argument, NULL, 1008},     {"host-parse", no_argument, NULL, 1009},
        {"phase-reads", no_argument, NULL, 1010},       {"inflate", required_argument, NULL, 1011},
        {"gz-ring-bytes", required_argument, NULL, 1012}, {"stats-json", required_argument, NULL, 1013},
        {"park-gb", required_argument, NULL, 1014},     {"name-cache", required_argument, NULL, 1015},
        {"deal", required_argument, NULL, 1016},        {"route", required_argument, NULL, 1017},
        {0, 0, 0, 0}};
    static char optstring[] = "p:m:l:r:t:w:u:f:q:h";             // classify.cpp:387
    std::string hap0, hap1, save_table, load_table;
    std::string r1("CTGTCTCTTATACACATCTTAGGAAGACAAGCACTGACGACATGA");   // classify.cpp:312
    std::string r2("TCTGCTGAGTCGAGAACGTCTCTGTGAGCCAAGGAGTTGCTCTGG");   // classify.cpp:313
    std::vector<std::string> read;
    int t_num = 8, device = 0;
    std::vector<int> devices;
    // (counters for 16M barcodes from the start: 512 MB of HBM and a fill -- a regrowth reads everything back and allocates anew, four
    // times on the way to BASELINE config 3's 10M barcodes)
    size_t batch_reads = 0, block_mb = 256, initial_barcodes = 1u << 24;
    bool stats = false, host_parse = false;
    // What changes what a production run allocates or writes is a FLAG; the environment variable each one replaces stays as an alias
    // (a flag wins).  The library reads its switches from the environment, once: a flag is put there before the first library call.
    bool phase_reads = false;
    std::string stats_json, route_mode;
    {
        const char *pr = getenv("HAST_PHASE_READS");
        phase_reads = pr && *pr && strcmp(pr, "0") != 0;
    }
    double w0 = 1.0, w1 = 1.0;
    for (;;) {
        int c = getopt_long(argc, argv, optstring, long_options, NULL);
        if (c < 0) break;
        switch (c) {
        case 'f': r1 = optarg; break;
        case 'q': r2 = optarg; break;
        case 'p': hap0 = optarg; break;
        case 'm': hap1 = optarg; break;
        case 'r': read.push_back(optarg); break;
        case 't': t_num = atoi(optarg); break;
        case 'u': w1 = atof(optarg); break;
        case 'w': w0 = atof(optarg); break;
        case 1001: device = atoi(optarg); break;
        case 1002: batch_reads = (size_t)std::max(1L, atol(optarg)); break;
        case 1003: stats = true; break;
        case 1004: block_mb = (size_t)std::max(1L, atol(optarg)); break;
        case 1005: initial_barcodes = (size_t)std::max(1L, atol(optarg)); break;
        case 1006: save_table = optarg; break;
        case 1007: load_table = optarg; break;
        case 1009: host_parse = true; break;
        case 1010: phase_reads = true; break;
        case 1011:
            if (strcmp(optarg, "host") && strcmp(optarg, "device") && strcmp(optarg, "zlib")) { print_usage(); return -1; }
            setenv("HAST_INFLATE", optarg, 1);
            break;
        case 1012: setenv("HAST_GZ_RING_BYTES", optarg, 1); break;
        case 1013: stats_json = optarg; stats = true; break;
        case 1014: setenv("HAST_PARK_GB", optarg, 1); break;
        case 1015: setenv("HAST_NAME_CACHE", optarg, 1); break;
        case 1016:
            if (strcmp(optarg, "files") && strcmp(optarg, "blocks")) { print_usage(); return -1; }
            setenv("HAST_DEAL", optarg, 1);
            break;
        case 1017:
            if (strcmp(optarg, "host") && strcmp(optarg, "device")) { print_usage(); return -1; }
            route_mode = optarg;
            break;
        case 1008:
            for (const char *q = optarg; *q;) {
                char *end;
                const long v = strtol(q, &end, 10);
                if (end == q || v < 0) { print_usage(); return -1; }
                devices.push_back((int)v);
                q = *end == ',' ? end + 1 : end;
                if (*end && *end != ',') { print_usage(); return -1; }
            }
            break;
        case 'h':
        default: print_usage(); return -1;
        }
    }
    if (((hap0.empty() || hap1.empty()) && load_table.empty()) || read.empty() || t_num < 1) {   // classify.cpp:425-428
        print_usage();
        return -1;
    }
    if (devices.empty()) devices.push_back(device);
    device = devices[0];
    fprintf(stderr, "__START__\n");
    fprintf(stderr, " use hap0 weight %g\n", w0);
    fprintf(stderr, " use hap1 weight %g\n", w1);
    logtime();
    const double t_start = now_s();

    // --batch-reads N (tests / small inputs): shrink the blocks so that a batch holds about N records
    size_t block_bytes = block_mb << 20;
    if (batch_reads) block_bytes = std::max<size_t>(4096, std::min(block_bytes, batch_reads * 320));
    const bool block_given = block_mb != 256 || batch_reads;
    // GPU framing: bytes per block of a file, blocks a file may have between its reader and the commit
    // (a multiple of 4 KB, as the library's blocks are: the blocks of a striped stream must be full)
    const size_t fq_cap = (std::max<size_t>(4096, block_given ? std::min<size_t>(block_bytes, 256u << 20) : (16u << 20)) + 4095) & ~(size_t)4095;
    const int fq_bufs = 6;

    // ---- load_kmers (classify.cpp:30-46): both files to memory, table built on the GPU --------
    size_t K = 0;
    hast_ctx *ctx = nullptr;
    std::vector<hast_ctx *> ctxs;
    // The FASTQ streams of the first files (pinned staging + device buffers: ~0.1 s of page pinning) are set up by a thread
    // of their own while this one reads the k-mer files and builds the table.
    std::vector<hast_fq *> pre_fq, done_fq;
    std::vector<hast_gz *> pre_gz;                         // .gz inputs opened (and being inflated) while the table is built
    std::vector<hast_status> pre_gz_status;
    std::vector<std::string> pre_gz_error;
    std::vector<std::thread> gz_closers;
    std::vector<hast_names *> name_caches, own_caches;     // per context / per GPU: device-side dictionary barcode text -> id
    Naming naming;
    std::thread pre_thread;
    std::string pre_error;
    // several GPUs: the blocks of every file go to all of them in turn (a striped stream); HAST_DEAL=files deals whole files
    bool stripe = false;
    // .gz inputs are inflated ON THE GPU (hast_gz_*: the compressed bytes cross PCIe, the framer reads the inflated bytes where they
    // lie) when they are ordinary gzip files; blocked gzip (BGZF: thousands of one-block members, which the host inflates side by
    // side), pipes and ".gz" files that are not gzip stay with the host decoders, as does everything under HAST_INFLATE=host|zlib
    std::vector<char> dev_gz(read.size(), 0);
    {
        const char *which = getenv("HAST_INFLATE");
        const bool allow = !host_parse && (!which || !strcmp(which, "device"));
        for (size_t i = 0; allow && i < read.size(); i++) {
            const std::string &r = read[i];
            struct stat sb;
            if (r.size() <= 3 || r.compare(r.size() - 3, 3, ".gz") != 0 || stat(r.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) continue;
            FILE *fp = fopen(r.c_str(), "rb");
            if (!fp) continue;                                      // (reported where the file is opened for good)
            unsigned char magic[2] = {0, 0};
            const bool gzip_magic = fread(magic, 1, 2, fp) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
            rewind(fp);
            dev_gz[i] = (gzip_magic || sb.st_size == 0) && !hast::BgzfReader::probe(fp);
            fclose(fp);
        }
    }
    // Bytes per block of an input.  A block costs a dozen small kernels and two host round trips besides its bytes.  For a .gz input
    // inflated on the GPU that is what bounds the read phase at BASELINE size (2 x 34 GB = 4 090 blocks of 16 MB): 64-MB blocks 1.70-1.82 s
    // against 2.23-2.26 s, alternating on one box (profiles/round6_cli_c2_ab_blocks.txt) -- and such a stream has no block to pin on
    // the host.  Plain files stay at 16 MB: their read phase is the PCIe upload whatever the block (1.43-1.48 s with 32 MB against
    // 1.34-1.50 s), their blocks are pinned host memory (0.7 ms per MB, six buffers a stream), and small .gz inputs want their first
    // block early.
    auto cap_of = [&](size_t file_index) -> size_t {
        if (block_given || host_parse || !dev_gz[file_index]) return fq_cap;
        struct stat sb;
        if (stat(read[file_index].c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) return fq_cap;
        return (uint64_t)sb.st_size >= (1ull << 30) ? (size_t)(64u << 20) : fq_cap;
    };
    auto make_fq = [&](size_t file_index, hast_fq **out) -> hast_status {
        const size_t fq_cap = cap_of(file_index);
        // (a .gz file inflated on the GPUs: the passes of its one deflate stream go to the GPUs in turn, hast_gz_open_multi, and its
        // inflated blocks -- written on the device -- are dealt to the contexts like a plain file's)
        if (stripe)
            return hast_fq_create_striped_ex(ctxs.data(), (int)ctxs.size(), fq_cap, std::max(2, (fq_bufs + (int)ctxs.size() - 1) / (int)ctxs.size()), name_caches.data(),
                                             dev_gz[file_index] ? 1 : 0, out);
        return hast_fq_create_ex(ctxs[file_index % ctxs.size()], fq_cap, fq_bufs, name_caches[file_index % ctxs.size()], dev_gz[file_index] ? 1 : 0, out);
    };
    // --stats: the device's free memory, sampled every 20 ms from here to the end (the HBM head-room line)
    std::atomic<bool> hbm_stop{false};
    std::atomic<size_t> hbm_free_min{~(size_t)0}, hbm_total{0};
    std::thread hbm_thread;
    auto contexts_ready = [&]() {
        if (stats)
            hbm_thread = std::thread([&hbm_stop, &hbm_free_min, &hbm_total, ctx] {
                while (!hbm_stop.load()) {
                    size_t f = 0, t = 0;
                    if (hast_dev_mem_info(ctx, &f, &t, nullptr) == HAST_OK) {
                        hbm_total = t;
                        if (f < hbm_free_min.load()) hbm_free_min = f;
                    }
                    std::this_thread::sleep_for(std::chrono::milliseconds(20));
                }
            });
        ctxs.push_back(ctx);
        {
            // (the other GPUs' contexts at the same time: a context is a device's first HIP calls and a few streams, 50-100 ms each)
            std::vector<hast_ctx *> made(devices.size(), nullptr);
            std::vector<std::thread> makers;
            for (size_t i = 1; i < devices.size(); i++)
                makers.emplace_back([&, i] {
                    if (hast_ctx_create(devices[i], (int)K, &made[i]) != HAST_OK) made[i] = nullptr;
                });
            for (std::thread &t : makers) t.join();
            for (size_t i = 1; i < devices.size(); i++) {
                if (!made[i]) die(4, "cannot create GPU context");
                ctxs.push_back(made[i]);
            }
        }
        if (host_parse) return;
        const char *deal = getenv("HAST_DEAL");
        stripe = ctxs.size() > 1 && !(deal && !strcmp(deal, "files"));
        // (2 x 32 B per barcode it can hold + 16 B of text by id: 1.3 GB for 16M -- BASELINE config 3 has 10M barcodes; what does not fit
        // is named by the host, in an id range of its own)
        size_t name_cap = std::max<size_t>(initial_barcodes, 1u << 24);
        if (const char *e = getenv("HAST_NAME_CACHE")) name_cap = (size_t)atol(e);
        // HAST_NAME_DICT=0: the table only caches what the host's dictionary names (up to round 5); HAST_NAME_DICT=context: a dictionary
        // per CONTEXT even where contexts share a GPU (tests: the merge by text of several GPUs' dictionaries, on one GPU)
        const char *nd = getenv("HAST_NAME_DICT");
        naming.device_dict = name_cap && !(nd && !strcmp(nd, "0"));
        const bool per_context = nd && !strcmp(nd, "context");
        for (size_t i = 0; i < ctxs.size(); i++) {
            // one per GPU: contexts that share a device (--devices 0,0) share it
            hast_names *nm = nullptr;
            int group = -1;
            for (size_t j = 0; j < i && !nm && !per_context; j++)
                if (devices[j] == devices[i]) { nm = name_caches[j]; group = naming.group_of[j]; }
            if (!nm && name_cap) {
                CK(naming.device_dict ? hast_names_create_dict(ctxs[i], name_cap, &nm) : hast_names_create(ctxs[i], name_cap, &nm), "creating the barcode dictionary");
                own_caches.push_back(nm);
                group = (int)naming.groups.size();
                naming.groups.push_back(nm);
            }
            name_caches.push_back(nm);
            naming.group_of.push_back(group);
        }
        if (naming.device_dict) naming.host_base = hast_names_limit(naming.groups[0]);
        pre_fq.assign(std::min<size_t>(read.size(), stripe ? 2 : std::max<size_t>(4, 2 * ctxs.size())), nullptr);
        pre_gz.assign(pre_fq.size(), nullptr);
        pre_gz_status.assign(pre_fq.size(), HAST_OK);
        pre_thread = std::thread([&] {
            // the first .gz files are opened NOW: their compressed bytes go to the device and their first passes are decoded while the
            // k-mer files are loaded and the table is built (an inflated stream needs a GPU, not the table; the symbols wait in the
            // stream's arenas) -- ~0.1 s of decode that used to start when the read phase did.  A thread per input: opening a stream and
            // creating its FASTQ framer is mostly page pinning and device allocation, which the inputs need not queue up for.
            std::mutex err_mu;
            std::vector<std::thread> per_file;
            for (size_t i = 0; i < pre_fq.size(); i++)
                per_file.emplace_back([&, i] {
                    if (dev_gz[i]) {
                        pre_gz_status[i] = stripe ? hast_gz_open_multi(ctxs.data(), (int)ctxs.size(), read[i].c_str(), &pre_gz[i])
                                                  : hast_gz_open(ctxs[i % ctxs.size()], read[i].c_str(), &pre_gz[i]);
                        if (pre_gz_status[i] != HAST_OK) {
                            std::lock_guard<std::mutex> g(err_mu);
                            pre_gz_error.push_back(hast_last_error());
                        }
                    }
                    if (make_fq(i, &pre_fq[i]) != HAST_OK) {
                        std::lock_guard<std::mutex> g(err_mu);
                        pre_error = hast_last_error();
                    }
                });
            for (std::thread &t : per_file) t.join();
        });
    };
    double t_loaded = 0, t_ctx = 0;
    if (!load_table.empty()) {
        // binary key-set cache written by --save-table (both sets, after the adaptor scrub of that run)
        int kk = 0;
        if (hast_table_file_info(load_table.c_str(), &kk, nullptr) != HAST_OK) die(2, "cannot use --load-table file");
        K = (size_t)kk;
        if (hast_ctx_create(device, kk, &ctx) != HAST_OK) die(4, "cannot create GPU context");
        t_ctx = now_s();
        contexts_ready();
        fprintf(stderr, "__load kmer table %s__\n", load_table.c_str());
        CK(hast_table_load(ctx, load_table.c_str(), 0.0), "loading the k-mer table");
        t_loaded = now_s();
    } else {
    // K = length of the first line of hap0 (classify.cpp:35-36); the files themselves are streamed into the table by the
    // library (hast_table_insert_text_file), a pipe or the like is read into memory first
    // (a pipe can be read only once: it is read whole now and its first bytes serve as the head)
    const std::string *hap_path[2] = {&hap0, &hap1};
    std::vector<char> txt[2], head;
    bool streamed[2] = {false, false};
    size_t text_bytes[2] = {0, 0};
    for (int h = 0; h < 2; h++) {
        struct stat sb;
        if (stat(hap_path[h]->c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) {
            streamed[h] = true;
            text_bytes[h] = (size_t)sb.st_size;
        } else {
            if (!slurp(*hap_path[h], txt[h])) die(2, ("cannot read " + *hap_path[h]).c_str());
            text_bytes[h] = txt[h].size();
        }
    }
    if (streamed[0]) {
        FILE *f = fopen(hap0.c_str(), "rb");
        if (!f) die(2, ("cannot read " + hap0).c_str());
        head.resize(4096);
        head.resize(fread(head.data(), 1, head.size(), f));
        fclose(f);
    } else head.assign(txt[0].begin(), txt[0].begin() + (long)std::min<size_t>(txt[0].size(), 4096));
    const void *nl0 = memchr(head.data(), '\n', head.size());
    K = nl0 ? (size_t)((const char *)nl0 - head.data()) : head.size();   // :35-36
    if (K < 1 || K > 32) {
        fprintf(stderr, "classify: ERROR: K=%zu%s (length of the first line of %s) is outside [1,32]\n", K, (!nl0 && head.size() == 4096) ? " or more" : "", hap0.c_str());
        return 3;
    }
    if (hast_ctx_create(device, (int)K, &ctx) != HAST_OK) die(4, "cannot create GPU context");
    t_ctx = now_s();
    contexts_ready();
    CK(hast_table_reserve(ctx, text_bytes[0] / (K + 1) + text_bytes[1] / (K + 1) + 2, 0.0), "allocating the k-mer table");
    for (int h = 0; h < 2; h++) {
        fprintf(stderr, "__load hap%d kmers__\n", h);
        uint64_t lines = 0;
        hast_status st = streamed[h] ? hast_table_insert_text_file(ctx, h, hap_path[h]->c_str(), &lines)
                                     : hast_table_insert_text(ctx, h, txt[h].data(), txt[h].size(), &lines);
        if (st == HAST_ERR_FORMAT) die(3, "k-mer file is not one K-mer per line");
        if (st == HAST_ERR_IO) die(2, ("cannot read " + *hap_path[h]).c_str());
        if (st != HAST_OK) die(4, "building the k-mer table");
        if (h == 0 && !nl0 && text_bytes[0] == K) {
            // a single unterminated line: the reference still inserts the FIRST line of hap0 (:35-39)
            uint64_t key = hast_canon_kmer(head.data(), (int)K);
            CK(hast_table_insert_keys(ctx, 0, &key, 1), "building the k-mer table");
            lines = 1;
        }
        fprintf(stderr, "Recorded %llu haplotype %d specific %zu-mers\n", (unsigned long long)lines, h, K);   // :45
        std::vector<char>().swap(txt[h]);
    }
    t_loaded = now_s();
    }

    // ---- InitAdaptor (classify.cpp:314-339) ---------------------------------------------------
    fprintf(stderr, "Adaptor forward :%s\n", r1.c_str());
    fprintf(stderr, "Adaptor reverse :%s\n", r2.c_str());
    {
        std::vector<uint64_t> keys;
        for (const std::string *ad : {&r1, &r2}) {
            if (ad->size() < K) {
                fprintf(stderr, " WARN : adaptor shorter than K ignored\n");
                continue;
            }
            std::vector<uint64_t> km(ad->size() - K + 1);
            size_t n = hast_chop_read(ad->data(), ad->size(), (int)K, km.data());
            for (size_t i = 0; i < n; i++)
                if (std::find(keys.begin(), keys.end(), km[i]) == keys.end()) keys.push_back(km[i]);   // a repeat finds nothing the 2nd time
        }
        std::vector<uint8_t> hit(keys.size());
        CK(hast_table_erase(ctx, keys.data(), keys.size(), hit.data()), "adaptor scrub");
        char buf[40];
        for (size_t i = 0; i < keys.size(); i++)
            for (int h = 0; h < 2; h++)
                if (hit[i] & (1 << h)) {
                    hast_kmer_to_str(keys[i], (int)K, buf);
                    fprintf(stderr, " INFO : erase a adaptor kmer from hap %d ; kmer= %s\n", h, buf);   // :321,325
                }
    }
    uint64_t n_set[2] = {0, 0};
    CK(hast_table_sizes(ctx, &n_set[0], &n_set[1]), "counting set sizes");
    if (!save_table.empty()) CK(hast_table_save(ctx, save_table.c_str()), "writing --save-table file");
    // the other GPUs get a copy of the finished table (after the adaptor scrub), peer to peer
    // (all at once: every GPU pulls its copy over its own xGMI link from the first one -- one after the other, seven copies of the 50 GB
    // of BASELINE config 3's table and filter are seven times the one copy's time.  The source's filter is built first, once: the clones
    // only read the source then.)
    if (ctxs.size() > 1) {
        CK(hast_filter_build(ctx), "building the k-mer filter");
        std::vector<std::thread> cloners;
        std::vector<std::string> clone_err(ctxs.size());
        for (size_t i = 1; i < ctxs.size(); i++)
            cloners.emplace_back([&, i] {
                if (hast_table_clone(ctxs[i], ctx) != HAST_OK) clone_err[i] = std::string("copying the k-mer table to another GPU (") + hast_last_error() + ")";
            });
        for (std::thread &t : cloners) t.join();
        for (const std::string &e : clone_err)
            if (!e.empty()) {
                fprintf(stderr, "classify: ERROR: %s\n", e.c_str());
                fflush(stderr);
                _exit(4);
            }
    }
    size_t next_ctx = 0;
    const double t_pre_wait0 = now_s();
    if (pre_thread.joinable()) {
        pre_thread.join();
        if (!pre_error.empty()) {
            fprintf(stderr, "classify: ERROR: creating the FASTQ stream (%s)\n", pre_error.c_str());
            fflush(stderr);
            _exit(4);                                          // (threads of this program and of the library are at work: no destructors)
        }
    }
    const double t_pre_waited = now_s() - t_pre_wait0;
    logtime();
    const double t_scrubbed = now_s();

    // ---- processFastq (classify.cpp:238-278) for each --read, in order ------------------------
    // reader thread -> blocks of raw bytes -> t_num workers index newlines and parse records in parallel
    // -> pinned staging of the GPU library -> classify (asynchronous, double-buffered)
    hast::WorkerPool pool(t_num);
    hast::BarcodeDict dict;
    std::vector<hast::BarcodeDict::Cache> caches(pool.size());
    Counts acc;
    // (a device dictionary hands out ids below host_base; the first id the host has to give lies there: counters for both from the start,
    // unless --initial-barcodes asks for less, tests)
    flush_counts(ctxs, acc, naming, pool, 0, initial_barcodes == (1u << 24) && naming.device_dict ? naming.host_base + 4096 : initial_barcodes);
    const int T = pool.size();
    uint64_t total_reads = 0, total_bases = 0;
    std::vector<std::vector<uint32_t>> nl(T);          // per-worker newline positions of the current block
    std::vector<uint32_t> allnl;
    std::vector<uint64_t> part_bytes(T + 1);
    std::vector<uint32_t> part_max(T), part_err(T);
    // Several --read files are streamed CONCURRENTLY (each has its own reader thread, so .gz files inflate in
    // parallel -- zlib is the serial bottleneck of real inputs) and their blocks are parsed round-robin.  Counts are
    // sums, so the interleaving cannot change the output; the reference handles the files one after the other.
    struct FileState {
        std::string name;
        std::unique_ptr<hast::BlockSource> src;
        std::vector<char> carry;                       // bytes of an incomplete record at the end of a block
    };
    const std::string *cur_name = nullptr;
    // parse `n_rec` complete records whose newline positions are in allnl (4 per record) from `data`
    auto parse_records = [&](const char *data, size_t n_rec) {
        if (n_rec == 0) return;
        uint8_t *hb;
        uint64_t *ho;
        uint32_t *hi;
        const size_t span = (size_t)allnl[4 * n_rec - 1] + 1;
        hast_ctx *bctx = ctxs[next_ctx++ % ctxs.size()];                               // batches are dealt round-robin to the GPUs
        CK(hast_batch_begin(bctx, span, n_rec, &hb, &ho, &hi), "staging a batch");
        auto rec_range = [&](int t, size_t &lo, size_t &hi_) { lo = n_rec * (size_t)t / T; hi_ = n_rec * (size_t)(t + 1) / T; };
        pool.run([&](int t) {                          // pass 1: bytes of bases per worker
            size_t lo, hi_;
            rec_range(t, lo, hi_);
            uint64_t sum = 0;
            uint32_t mx = 0;
            for (size_t i = lo; i < hi_; i++) {
                uint32_t len = allnl[4 * i + 1] - allnl[4 * i] - 1;
                sum += len;
                mx = std::max(mx, len);
            }
            part_bytes[t + 1] = sum;
            part_max[t] = mx;
        });
        part_bytes[0] = 0;
        for (int t = 0; t < T; t++) part_bytes[t + 1] += part_bytes[t];
        pool.run([&](int t) {                          // pass 2: barcode ids + bases into pinned staging
            size_t lo, hi_;
            rec_range(t, lo, hi_);
            uint64_t off = part_bytes[t];
            uint32_t err = 0;
            for (size_t i = lo; i < hi_; i++) {
                const size_t h0 = i ? (size_t)allnl[4 * i - 1] + 1 : 0, h1 = allnl[4 * i], s1 = allnl[4 * i + 1];
                size_t bs, bn;
                hast_parse_barcode(data + h0, h1 - h0, &bs, &bn);                          // classify.cpp:189
                hi[i] = dict.get(std::string_view(data + h0 + bs, bn), caches[t]);
                const size_t len = s1 - h1 - 1;
                ho[i] = off;
                memcpy(hb + off, data + h1 + 1, len);
                if (len < K && !memchr(data + h1 + 1, 'N', len)) err = 1;                  // kmer.h:171
                off += len;
            }
            part_err[t] = err;
        });
        ho[n_rec] = part_bytes[T];
        uint32_t mx = 0;
        for (int t = 0; t < T; t++) {
            mx = std::max(mx, part_max[t]);
            if (part_err[t]) {
                fprintf(stderr, "classify: ERROR: read shorter than K=%zu in %s\n", K, cur_name->c_str());
                fflush(stdout);                                                            // reference: assert abort.  _exit: the HBM sampler and the
                fflush(stderr);                                                            // library's threads are inside the HIP runtime -- exit() would take
                _exit(3);                                                                  // it down under them (seen: SIGSEGV instead of status 3)
            }
        }
        if (dict.size() > acc.device_cap) flush_counts(ctxs, acc, naming, pool, dict.size(), std::max(dict.size() * 2, acc.device_cap * 2));
        CK(hast_batch_submit(bctx, n_rec, mx), "classifying a batch");
        total_reads += n_rec;
        total_bases += part_bytes[T];
    };
    // one block of one file; returns false when that file is finished
    auto process_block = [&](FileState &fs) -> bool {
        cur_name = &fs.name;
        hast::BlockSource &src = *fs.src;
        std::vector<char> &carry = fs.carry;
        std::vector<char> blk = src.next();
        const bool last = blk.empty();
        if (last && !src.error().empty()) die(2, (fs.name + ": " + src.error()).c_str());
        constexpr size_t kPad = hast::BlockSource::kFrontPad;
        // work area = carry (incomplete record of the previous block) + this block's data
        const char *data;
        size_t len;
        if (last) {
            data = carry.data();
            len = carry.size();
        } else if (carry.size() <= kPad) {
            if (!carry.empty()) memcpy(blk.data() + kPad - carry.size(), carry.data(), carry.size());   // in front, no block copy
            data = blk.data() + kPad - carry.size();
            len = blk.size() - kPad + carry.size();
        } else {                                                                     // giant record: slow path
            carry.insert(carry.end(), blk.begin() + kPad, blk.end());
            data = carry.data();
            len = carry.size();
        }
        if (len == 0) return false;
        if (len >= (1ull << 32)) die(3, "a single FASTQ record spans more than 4 GB");
        // newline index, in parallel
        pool.run([&](int t) {
            auto &v = nl[t];
            v.clear();
            const size_t lo = len * (size_t)t / T, hi_ = len * (size_t)(t + 1) / T;
            const char *p = data + lo, *e = data + hi_;
            while (p < e && (p = (const char *)memchr(p, '\n', (size_t)(e - p)))) {
                v.push_back((uint32_t)(p - data));
                ++p;
            }
        });
        std::vector<size_t> cum(T + 1, 0);
        for (int t = 0; t < T; t++) cum[t + 1] = cum[t] + nl[t].size();
        const size_t total_nl = cum[T];
        allnl.resize(total_nl);
        pool.run([&](int t) { if (!nl[t].empty()) memcpy(allnl.data() + cum[t], nl[t].data(), nl[t].size() * 4); });
        const size_t n_rec = total_nl / 4;
        parse_records(data, n_rec);
        const size_t consumed = n_rec ? (size_t)allnl[4 * n_rec - 1] + 1 : 0;
        if (last) {
            // end of input: what is left holds < 4 newlines.  Reference framing (classify.cpp:257-268): the
            // header must be newline-terminated; bases are whatever follows up to the next newline or EOF.
            const char *p = data + consumed, *e = data + len;
            const char *h_end = (const char *)memchr(p, '\n', (size_t)(e - p));
            if (h_end) {
                const char *s0 = h_end + 1;
                const char *s_end = (const char *)memchr(s0, '\n', (size_t)(e - s0));
                if (!s_end) s_end = e;
                std::string tail(p, (size_t)(h_end - p));
                tail.push_back('\n');
                tail.append(s0, (size_t)(s_end - s0));
                tail.append("\n+\n\n");
                allnl.clear();
                for (size_t i = 0; i < tail.size(); i++)
                    if (tail[i] == '\n') allnl.push_back((uint32_t)i);
                parse_records(tail.data(), 1);
            }
            return false;
        }
        // keep the incomplete record for the next block
        std::vector<char> rest(data + consumed, data + len);
        carry.swap(rest);
        src.recycle(std::move(blk));
        return true;
    };
    if (host_parse) {
        const size_t max_active = 4;                   // concurrent reader threads (3 prefetched blocks each)
        std::vector<FileState> active;
        size_t next_file = 0;
        auto open_next = [&]() {
            const std::string &r = read[next_file++];
            fprintf(stderr, "__process read: %s\n", r.c_str());
            FileState fs;
            fs.name = r;
            fs.src.reset(new hast::BlockSource());
            if (!fs.src->open(r, block_bytes)) die(2, ("cannot open " + r).c_str());
            active.push_back(std::move(fs));
        };
        while (next_file < read.size() && active.size() < max_active) open_next();
        while (!active.empty()) {
            for (size_t i = 0; i < active.size();) {
                if (process_block(active[i])) {
                    ++i;
                    continue;
                }
                logtime();
                fprintf(stderr, "__process read done__\n");
                active.erase(active.begin() + (long)i);
                if (next_file < read.size()) open_next();
            }
        }
    }
    else {
        // ---- raw bytes to the GPU, records framed there (hast_fq_*, fq_kernels.hip) -------------------------------------
        // per file: a reader thread fills the pinned buffers the library hands out (pread / inflate straight into them); this
        // thread submits them, maps the barcode text of every record to its id (in parallel) and commits.  Files go to the
        // GPUs round-robin; a file's blocks stay on one GPU (the unfinished record at the end of a block is carried on the device).
        struct Feed {
            std::string name;
            hast::BlockSource src;
            hast_fq *fq = nullptr;
            hast_gz *gz = nullptr;                                 // the file is inflated on the GPU: blocks are filled there
            std::thread th;
            std::mutex mu;
            std::condition_variable cv;
            std::deque<std::pair<uint8_t *, hast_stream>> empty;   // acquired, waiting for the reader (device blocks: the stream their writes go on)
            struct Filled { size_t n; bool last; std::string err; };
            std::deque<Filled> filled;                             // filled, in order, waiting for hast_fq_submit
            bool stop = false, eof_acquired = false;
            size_t held = 0;                                       // acquired and not yet committed
            size_t submitted = 0, opened = 0, acquired = 0;
            size_t ctx_index = 0;                                  // a whole file on one context (not striped)
        };
        const int n_buf = stripe ? std::max(2, (fq_bufs + (int)ctxs.size() - 1) / (int)ctxs.size()) : fq_bufs;     // per context
        double t_create = 0;
        std::mutex wake_mu;
        std::condition_variable wake_cv;
        uint64_t wake_gen = 0;
        std::vector<std::unique_ptr<Feed>> active;
        size_t next_file = 0;
        auto open_next = [&]() {
            const std::string &r = read[next_file];
            fprintf(stderr, "__process read: %s\n", r.c_str());
            std::unique_ptr<Feed> f(new Feed());
            f->name = r;
            if (dev_gz[next_file]) {
                hast_status gs;
                if (next_file < pre_gz.size()) {                   // opened ahead, by the set-up thread
                    gs = pre_gz_status[next_file];
                    f->gz = pre_gz[next_file];
                } else
                    gs = stripe ? hast_gz_open_multi(ctxs.data(), (int)ctxs.size(), r.c_str(), &f->gz) : hast_gz_open(ctxs[next_file % ctxs.size()], r.c_str(), &f->gz);
                if (gs == HAST_ERR_UNSUPPORTED) {                  // e.g. no room on the device: the host inflates
                    f->gz = nullptr;
                    dev_gz[next_file] = 0;
                    if (next_file < pre_fq.size() && pre_fq[next_file]) {   // (a stream of device-side blocks was set up for it)
                        hast_fq_destroy(pre_fq[next_file]);
                        pre_fq[next_file] = nullptr;
                    }
                } else if (gs != HAST_OK) die(2, ("cannot open " + r).c_str());
            }
            const size_t cap = cap_of(next_file);              // (after a .gz input has gone to the host decoders, if it had to)
            if (!f->gz) {
                if (!f->src.open(r, cap, false)) die(2, ("cannot open " + r).c_str());
                f->src.set_readers(std::max(4, std::min(16, t_num / (int)std::min<size_t>(read.size(), 2))));
            }
            const double tc0 = now_s();
            if (next_file < pre_fq.size() && pre_fq[next_file]) f->fq = pre_fq[next_file];     // set up while the table was built
            else CK(make_fq(next_file, &f->fq), "creating the FASTQ stream");
            t_create += now_s() - tc0;
            f->ctx_index = next_file % ctxs.size();
            next_file++;
            Feed *fp = f.get();
            if (hast_fq_block_bytes(f->fq) != cap) die(4, "internal: a stream's block size is not its input's");
            f->th = std::thread([fp, cap, &wake_mu, &wake_cv, &wake_gen] {
                for (;;) {
                    uint8_t *buf;
                    hast_stream fill_stream;
                    {
                        std::unique_lock<std::mutex> g(fp->mu);
                        fp->cv.wait(g, [fp] { return fp->stop || !fp->empty.empty(); });
                        if (fp->stop) return;
                        buf = fp->empty.front().first;
                        fill_stream = fp->empty.front().second;
                        fp->empty.pop_front();
                    }
                    Feed::Filled fl{0, false, std::string()};
                    const double tr0 = now_s();
                    if (fp->gz) {                                   // (buf is a DEVICE address: the translate kernel writes the block there)
                        size_t n = 0;
                        if (hast_gz_read_device(fp->gz, buf, cap, &n, fill_stream) != HAST_OK) fl.err = hast_last_error();
                        else if (n < cap) {
                            // a short read is the end of the stream -- or what could be decoded in front of damage (delivered first, as
                            // gzread does): the next call says which.  Without it a file damaged behind its first pass would end here
                            // as if it were complete.
                            size_t more = 0;
                            if (hast_gz_read_device(fp->gz, buf + n, cap - n, &more, fill_stream) != HAST_OK) fl.err = hast_last_error();
                            n += more;
                        }
                        fl.n = n;
                    } else
                    fl.n = fp->src.read_into(reinterpret_cast<char *>(buf), cap, fl.err);
                    if (getenv("HAST_TRACE_BLOCKS")) fprintf(stderr, "trace %s fill %.3f ms at %.4f\n", fp->name.c_str() + (fp->name.size() > 5 ? fp->name.size() - 5 : 0), (now_s() - tr0) * 1e3, now_s());
                    fl.last = fl.n < cap || !fl.err.empty();
                    {
                        std::lock_guard<std::mutex> g(fp->mu);
                        fp->filled.push_back(fl);
                    }
                    {
                        std::lock_guard<std::mutex> g(wake_mu);
                        ++wake_gen;
                    }
                    wake_cv.notify_one();
                    if (fl.last) return;
                }
            });
            active.push_back(std::move(f));
        };
        const size_t max_active = stripe ? 2 : std::max<size_t>(4, 2 * ctxs.size());
        while (next_file < read.size() && active.size() < max_active) open_next();
        uint64_t seen_gen = 0;
        std::vector<uint64_t> whole_file_records(ctxs.size(), 0);      // records of files dealt whole, per context
        double t_gpu_wait = 0, t_names = 0, t_commit = 0, t_idle = 0;
        uint64_t total_named = 0;
        // names the barcodes of the oldest submitted block of a feed and commits it
        auto open_block = [&](Feed &f) {
            hast_fq_block b;
            const double t0 = now_s();
            CK(hast_fq_next(f.fq, &b), "framing a block");
            const double t1 = now_s();
            t_gpu_wait += t1 - t0;
            if (b.short_read) {
                fprintf(stderr, "classify: ERROR: read shorter than K=%zu in %s\n", K, f.name.c_str());
                fflush(stdout);                                                            // reference: assert abort (kmer.h:171); _exit as above
                fflush(stderr);
                _exit(3);
            }
            const size_t n = (size_t)b.n_records;
            if (hast_fq_lanes(f.fq) <= 1 || !stripe) whole_file_records[f.ctx_index] += n;
            // records the device-side name cache did not know (all of them without a cache): text -> id in the job's dictionary
            const size_t nu = b.unknown ? (size_t)b.n_unknown : n;
            if (!b.bytes) {
                // a block that was filled on the device: the host copy is fetched only when a record's barcode text did not fit the
                // framer's 16-byte copy (longer than 15 bytes), or when there are no such copies (more records than they hold)
                bool need = !b.bc_text && nu > 0;
                for (size_t j = 0; !need && b.bc_text && j < nu; j++) need = b.bc_text[16 * (b.unknown ? b.unknown[j] : j)] == 0xFF;
                if (need) CK(hast_fq_block_host_bytes(f.fq, &b.bytes), "fetching a block");
            }
            auto name_range = [&](int t, size_t lo, size_t hi_) {
                for (size_t j = lo; j < hi_; j++) {
                    const size_t i = b.unknown ? b.unknown[j] : j;
                    const uint8_t *txt = b.bc_text ? b.bc_text + 16 * i : nullptr;      // the framer's compact copy of the barcode text
                    b.ids[i] = (uint32_t)naming.host_base +
                               (txt && txt[0] != 0xFF
                                    ? dict.get(std::string_view(reinterpret_cast<const char *>(txt) + 1, txt[0]), caches[t])
                                    : dict.get(std::string_view(reinterpret_cast<const char *>(b.bytes) + b.bc_pos[i], b.bc_len[i]), caches[t]));
                }
            };
            if (nu < 4096) name_range(0, 0, nu);
            else pool.run([&](int t) { name_range(t, nu * (size_t)t / T, nu * (size_t)(t + 1) / T); });
            total_named += nu;
            const double t2 = now_s();
            t_names += t2 - t1;
            // the counters must hold every id of this block: the device dictionary's (below dict_ids) and the host's (from host_base on)
            {
                const size_t need = std::max<size_t>(naming.device_dict ? (size_t)b.dict_ids : 0, dict.size() ? naming.host_base + dict.size() : 0);
                if (need > acc.device_cap)
                    flush_counts(ctxs, acc, naming, pool, dict.size(), std::max(dict.size() ? naming.host_base + 2 * dict.size() + 4096 : 2 * need, acc.device_cap * 2));
            }
            CK(hast_fq_commit(f.fq), "classifying a block");
            t_commit += now_s() - t2;
            if (getenv("HAST_TRACE_BLOCKS")) fprintf(stderr, "trace %s open wait %.3f name %.3f commit %.3f ms at %.4f (sub %zu open %zu)\n", f.name.c_str() + (f.name.size() > 5 ? f.name.size() - 5 : 0), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (now_s() - t2) * 1e3, now_s(), f.submitted, f.opened);
            f.opened++;
            f.held--;
            total_reads += n;
            total_bases += b.n_bases;
        };
        while (!active.empty()) {
            bool progress = false;
            for (size_t fi = 0; fi < active.size(); ++fi) {
                Feed &f = *active[fi];
                // 1. hand empty buffers to the reader
                // (device-side blocks: one buffer fewer may be in hand unsubmitted, include/hast.h)
                while (!f.eof_acquired && f.held < (size_t)n_buf * (size_t)hast_fq_lanes(f.fq) && (!f.gz || f.acquired - f.submitted + 1 < (size_t)n_buf * (size_t)hast_fq_lanes(f.fq))) {
                    uint8_t *buf;
                    hast_stream fill_stream = nullptr;
                    CK(hast_fq_acquire(f.fq, &buf), "staging a block");
                    if (f.gz) CK(hast_fq_device_block(f.fq, &buf, &fill_stream), "staging a block");
                    f.held++;
                    f.acquired++;
                    std::lock_guard<std::mutex> g(f.mu);
                    f.empty.push_back({buf, fill_stream});
                    f.cv.notify_one();
                    progress = true;
                }
                // 2. submit what the reader has filled (copy + framing run on the GPU from here on)
                for (;;) {
                    Feed::Filled fl;
                    {
                        std::lock_guard<std::mutex> g(f.mu);
                        if (f.filled.empty()) break;
                        fl = f.filled.front();
                        f.filled.pop_front();
                    }
                    if (!fl.err.empty()) die(2, (f.name + ": " + fl.err).c_str());
                    CK(f.gz ? hast_fq_submit_device(f.fq, fl.n, fl.last ? 1 : 0) : hast_fq_submit(f.fq, fl.n, fl.last ? 1 : 0), "framing a block");
                    if (getenv("HAST_TRACE_BLOCKS")) fprintf(stderr, "trace %s submit at %.4f\n", f.name.c_str() + (f.name.size() > 5 ? f.name.size() - 5 : 0), now_s());
                    f.submitted++;
                    if (fl.last) f.eof_acquired = true;
                    progress = true;
                }
            }
            // 3. a block whose record table has arrived: name its barcodes, commit.  One block per round, so that what the
            //    readers have filled meanwhile is submitted between two blocks (the GPU must never run out of queued copies)
            for (size_t fi = 0; fi < active.size();) {
                Feed &f = *active[fi];
                if (f.opened < f.submitted && hast_fq_poll(f.fq)) {
                    open_block(f);
                    progress = true;
                }
                if (f.eof_acquired && f.opened == f.submitted) {
                    {
                        std::lock_guard<std::mutex> g(f.mu);
                        f.stop = true;
                    }
                    f.cv.notify_all();
                    f.th.join();
                    if (f.gz) {
                        hast_gz_stats gs;
                        if (stats && hast_gz_get_stats(f.gz, &gs) == HAST_OK)
                            stat_line("__stats_gz__ file=%s compressed_bytes=%llu inflated_bytes=%llu chunks=%llu accepted=%llu followup_jobs=%llu followup_rounds=%llu members=%llu "
                                            "open_s=%.3f decode_s=%.3f windows_crc_s=%.3f chain_walk_s=%.3f producer_waited_for_upload_s=%.3f producer_waited_for_reader_s=%.3f reader_waited_for_decode_s=%.3f ring_bytes=%llu ring_laps=%llu upload_waited_for_ring=%llu\n",
                                    f.name.c_str(), (unsigned long long)gs.compressed_bytes, (unsigned long long)gs.out_bytes, (unsigned long long)gs.chunks,
                                    (unsigned long long)gs.accepted, (unsigned long long)gs.followup_jobs, (unsigned long long)gs.followup_rounds, (unsigned long long)gs.members,
                                    gs.open_s, gs.decode_s, gs.windows_crc_s, gs.chain_walk_s, gs.wait_upload_s, gs.wait_consumer_s, gs.wait_decode_s, (unsigned long long)gs.ring_bytes,
                                    (unsigned long long)gs.ring_laps, (unsigned long long)gs.upload_waited_for_ring);
                        // its device memory (the compressed file, the symbol arenas, windows) goes back now, not at the end of the run: a
                        // dozen finished .gz files would otherwise crowd the table out of HBM.  On a thread of its own: freeing synchronises.
                        hast_gz *z = f.gz;
                        gz_closers.emplace_back([z] { hast_gz_close(z); });
                    }
                    done_fq.push_back(f.fq);                       // (freed after the output: unpinning costs as much as pinning)
                    logtime();
                    fprintf(stderr, "__process read done__\n");
                    active.erase(active.begin() + (long)fi);
                    if (next_file < read.size()) open_next();
                    progress = true;
                    continue;
                }
                ++fi;
            }
            if (!progress) {                                       // everything waits for a reader thread
                const double t0 = now_s();
                std::unique_lock<std::mutex> g(wake_mu);
                // (a GPU event may be what we wait for; a striped stream relays the newline count in front of EVERY block through this loop:
                // count kernel -> host -> framing launch, so its wait is short)
                wake_cv.wait_for(g, std::chrono::microseconds(stripe ? 10 : 100), [&] { return wake_gen != seen_gen; });
                seen_gen = wake_gen;
                t_idle += now_s() - t0;
            }
        }
        if (stats && stripe) {
            std::string per;
            for (size_t g = 0; g < ctxs.size(); g++) {
                uint64_t n = whole_file_records[g];              // (files dealt whole: HAST_DEAL=files)
                for (hast_fq *q : done_fq) n += hast_fq_lane_records(q, (int)g);
                per += (g ? "," : "") + std::to_string(n);
            }
            stat_line("__stats_devices__ blocks_of_every_file_dealt_to=%zu records_per_context=%s\n", ctxs.size(), per.c_str());
        }
        if (stats)
            stat_line("__stats_read_phase__ waiting_for_file_bytes_s=%.3f waiting_for_gpu_framing_s=%.3f naming_barcodes_s=%.3f commit_s=%.3f stream_setup_s=%.3f records_named_on_host=%llu\n",
                    t_idle, t_gpu_wait, t_names, t_commit, t_create, (unsigned long long)total_named);
    }
    const double t_read_done = now_s();
    flush_counts(ctxs, acc, naming, pool, dict.size(), 1);
    const double t_classified = now_s();
    // the names by row: the device dictionary's texts by id (read once, now), then what the host named
    std::vector<uint8_t> dev_texts;
    size_t n_dev_names = 0;
    if (naming.device_dict) {                       // (several dictionaries: the first one holds every text since the merge)
        CK(hast_names_count(naming.groups[0], &n_dev_names), "asking the dictionary for its size");
        dev_texts.resize(16 * n_dev_names);
        CK(hast_names_texts(naming.groups[0], 0, n_dev_names, dev_texts.data()), "reading the dictionary's texts");
    }
    const size_t n_host_names = dict.size();
    std::vector<std::string_view> names(n_dev_names + n_host_names);
    if (naming.device_dict)
        pool.run([&](int t) {
            for (size_t i = n_dev_names * (size_t)t / T, e = n_dev_names * (size_t)(t + 1) / T; i < e; i++)
                names[i] = std::string_view(reinterpret_cast<const char *>(dev_texts.data()) + 16 * i + 1, dev_texts[16 * i]);
        });
    {
        std::vector<std::string_view> hn(n_host_names);
        pool.run([&](int t) { dict.names_range(hn, hast::BarcodeDict::n_shards() * (size_t)t / T, hast::BarcodeDict::n_shards() * (size_t)(t + 1) / T); });
        std::copy(hn.begin(), hn.end(), names.begin() + (long)n_dev_names);
    }
    // one run of counters in the order of `names`
    if (naming.device_dict) {
        acc.c0.resize(n_dev_names); acc.c1.resize(n_dev_names); acc.neg.resize(n_dev_names);
        acc.h0.resize(n_host_names); acc.h1.resize(n_host_names);
        acc.c0.insert(acc.c0.end(), acc.h0.begin(), acc.h0.end());
        acc.c1.insert(acc.c1.end(), acc.h1.begin(), acc.h1.end());
    }

    // ---- printBarcodeInfos (classify.cpp:93-102): byte-wise sorted rows ------------------------
    // (the reference walks a std::map<std::string, ...>: byte-wise lexicographic order, a prefix in front of what it is a prefix of.
    // BASELINE configs 2 / 3 have 1M / 10M barcodes: one std::sort of 10M names and one snprintf per row took seconds on one thread;
    // the names are dealt into 65536 buckets by their first two bytes (bucket order = byte order), the buckets are sorted and the rows
    // formatted by the parser threads)
    fprintf(stderr, "__print result__\n");
    const size_t nb = names.size();
    std::vector<uint32_t> order(nb);
    auto key16 = [&](uint32_t i) -> uint32_t {
        const std::string_view v = names[i];
        return (v.size() > 0 ? (uint32_t)(uint8_t)v[0] << 8 : 0u) | (v.size() > 1 ? (uint32_t)(uint8_t)v[1] : 0u);
    };
    if (nb < (1u << 16) || T == 1) {
        for (uint32_t i = 0; i < nb; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return names[a] < names[b]; });
    } else {
        constexpr size_t kB = 65536;
        std::vector<std::vector<uint32_t>> hist((size_t)T, std::vector<uint32_t>(kB, 0));
        pool.run([&](int t) {
            std::vector<uint32_t> &h = hist[(size_t)t];
            for (size_t i = nb * (size_t)t / T, e = nb * (size_t)(t + 1) / T; i < e; i++) h[key16((uint32_t)i)]++;
        });
        std::vector<size_t> bucket_at(kB + 1, 0);
        {
            size_t at = 0;
            for (size_t k = 0; k < kB; k++) {
                bucket_at[k] = at;
                for (int t = 0; t < T; t++) {
                    const uint32_t c = hist[(size_t)t][k];
                    hist[(size_t)t][k] = (uint32_t)at;              // (nb < 2^32: ids are 32-bit)
                    at += c;
                }
            }
            bucket_at[kB] = at;
        }
        pool.run([&](int t) {
            std::vector<uint32_t> &h = hist[(size_t)t];
            for (size_t i = nb * (size_t)t / T, e = nb * (size_t)(t + 1) / T; i < e; i++) order[h[key16((uint32_t)i)]++] = (uint32_t)i;
        });
        // the buckets, largest first, one at a time to whoever is free (stLFR barcodes are digits: a hundred buckets hold everything, and
        // ten of them sit next to each other -- dealt in runs of 64 keys, one thread got a tenth of the job); inside a bucket: by the next
        // byte, and the next (the ids are moved, the names are looked at once per level), std::sort below 2048 names
        std::vector<uint32_t> big;
        for (size_t k = 0; k < kB; k++)
            if (bucket_at[k + 1] - bucket_at[k] > 1) big.push_back((uint32_t)k);
        std::sort(big.begin(), big.end(), [&](uint32_t a, uint32_t b) { return bucket_at[a + 1] - bucket_at[a] > bucket_at[b + 1] - bucket_at[b]; });
        std::atomic<size_t> next_bucket{0};
        pool.run([&](int) {
            std::vector<uint32_t> tmp;
            struct Range { size_t lo, hi, depth; };
            std::vector<Range> todo;
            for (;;) {
                const size_t bi = next_bucket.fetch_add(1);
                if (bi >= big.size()) break;
                todo.push_back({bucket_at[big[bi]], bucket_at[big[bi] + 1], 2});
                while (!todo.empty()) {
                    const Range r = todo.back();
                    todo.pop_back();
                    const size_t n = r.hi - r.lo;
                    if (n < 2048 || r.depth > 64) {
                        std::sort(order.begin() + (long)r.lo, order.begin() + (long)r.hi, [&](uint32_t a, uint32_t b) { return names[a] < names[b]; });
                        continue;
                    }
                    // counting sort by the byte at r.depth; names that end here (shorter: a prefix of the others) come first
                    size_t cnt[257] = {0};
                    auto byte_at = [&](uint32_t id) -> size_t { const std::string_view v = names[id]; return v.size() > r.depth ? (size_t)(uint8_t)v[r.depth] + 1 : 0; };
                    for (size_t i = r.lo; i < r.hi; i++) cnt[byte_at(order[i])]++;
                    size_t at[258];
                    at[0] = 0;
                    for (int b = 0; b < 257; b++) at[b + 1] = at[b] + cnt[b];
                    tmp.resize(n);
                    {
                        size_t pos[257];
                        for (int b = 0; b < 257; b++) pos[b] = at[b];
                        for (size_t i = r.lo; i < r.hi; i++) tmp[pos[byte_at(order[i])]++] = order[i];
                    }
                    std::copy(tmp.begin(), tmp.begin() + (long)n, order.begin() + (long)r.lo);
                    for (int b = 1; b < 257; b++)                          // (slot 0: identical names cannot be, ids are one per name)
                        if (cnt[b] > 1) todo.push_back({r.lo + at[b], r.lo + at[b + 1], r.depth + 1});
                }
            }
        });
    }
    bool past_int = false;
    // rows: formatted by all threads, each a contiguous share of the sorted order, written in that order
    auto put_u64 = [](std::string &out, uint64_t v) {
        char tmp[24];
        int n = 0;
        do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (n) out.push_back(tmp[--n]);
    };
    auto format_rows = [&](size_t lo, size_t hi, std::string &out, bool &past) {
        for (size_t r = lo; r < hi; r++) {
            const uint32_t i = order[r];
            const std::string_view bc = names[i];
            const uint64_t c0 = i < acc.c0.size() ? acc.c0[i] : 0, c1 = i < acc.c1.size() ? acc.c1[i] : 0;
            const int hap = hast_get_hap(bc.data(), bc.size(), c0, c1, n_set[0], n_set[1], w0, w1);
            out.append(bc.data(), bc.size());
            // the reference prints `int` counters (classify.cpp:51,98-100): the same digits up to INT_MAX; past it the reference's
            // counter has overflowed (undefined behaviour there) -- the exact count is printed and the run says so once
            past = past || c0 > 0x7FFFFFFFull || c1 > 0x7FFFFFFFull;
            out.push_back('\t');
            if (hap < 0) out.append("-1");
            else out.push_back((char)('0' + hap));
            out.push_back('\t');
            put_u64(out, c0);
            out.push_back('\t');
            put_u64(out, c1);
            out.push_back('\n');
        }
    };
    // (the wrapper redirects stdout into phased.barcodes and tests only the exit status, classify_stlfr_reads.sh:149: a full disk
    // or a closed pipe must not leave a truncated table behind exit 0)
    const size_t kRowsPerPiece = 1u << 16;
    for (size_t r0 = 0; r0 < nb; r0 += kRowsPerPiece * (size_t)T) {
        const size_t r1 = std::min(nb, r0 + kRowsPerPiece * (size_t)T);
        std::vector<std::string> piece((size_t)T);
        std::vector<char> past((size_t)T, 0);
        pool.run([&](int t) {
            const size_t lo = r0 + (r1 - r0) * (size_t)t / T, hi = r0 + (r1 - r0) * (size_t)(t + 1) / T;
            bool p = false;
            piece[(size_t)t].reserve((hi - lo) * 24);
            format_rows(lo, hi, piece[(size_t)t], p);
            past[(size_t)t] = p;
        });
        for (int t = 0; t < T; t++) {
            past_int = past_int || past[(size_t)t];
            if (fwrite(piece[(size_t)t].data(), 1, piece[(size_t)t].size(), stdout) != piece[(size_t)t].size()) die_output();
        }
    }
    if (fflush(stdout) != 0) die_output();
    // ---- HAST_PHASE_READS=1: steps 10 and 11 of the wrapper (classify_stlfr_reads.sh:155-190) done here -------------------------
    // The wrapper derives three barcode lists from the table above with awk and then routes every record of every input to
    // <name>.{paternal,maternal,homozygous,nobarcode}.fastq with a single-threaded awk program, re-reading (and re-inflating) every
    // input.  This program has the barcodes' classes in memory and a GPU that inflates .gz inputs: with HAST_PHASE_READS set it writes the
    // lists, routes the records (quartering.h: the same bytes as the awk program's, incl. filter_reads.log and the ERROR lines) and
    // leaves the marker files step_10_done / step_11_done, at which the UNCHANGED wrapper skips its own steps 10 and 11.
    if (phase_reads) {
        const double t_ph0 = now_s();
        namespace hq = hast::quartering;
        const char *list_name[3] = {"paternal.unique.barcodes", "maternal.unique.barcodes", "homozygous.unique.barcodes"};
        // the call of every barcode once (getHap, classify.cpp:66-86): list 1 paternal (hap 0), 2 maternal (hap 1), 3 homozygous (-1)
        std::vector<uint8_t> list_of(nb);
        std::vector<char> has_sep((size_t)T, 0);
        pool.run([&](int t) {
            for (size_t i = nb * (size_t)t / T, e = nb * (size_t)(t + 1) / T; i < e; i++) {
                const std::string_view bc = names[i];
                const uint64_t c0 = i < acc.c0.size() ? acc.c0[i] : 0, c1 = i < acc.c1.size() ? acc.c1[i] : 0;
                const int hap = hast_get_hap(bc.data(), bc.size(), c0, c1, n_set[0], n_set[1], w0, w1);
                list_of[i] = hap == 0 ? 1 : hap == 1 ? 2 : 3;
                if (bc.find('#') != std::string_view::npos || bc.find('/') != std::string_view::npos) has_sep[(size_t)t] = 1;
            }
        });
        {
            // (one walk over the sorted rows, each thread a contiguous share: the three lists are the rows' first columns in row order)
            std::vector<std::string> part((size_t)T * 3);
            pool.run([&](int t) {
                for (size_t r = nb * (size_t)t / T, e = nb * (size_t)(t + 1) / T; r < e; r++) {
                    const uint32_t i = order[r];
                    std::string &text = part[(size_t)t * 3 + (size_t)(list_of[i] - 1)];
                    text.append(names[i].data(), names[i].size());
                    text.push_back('\n');
                }
            });
            for (int l = 0; l < 3; l++) {
                FILE *lf = fopen(list_name[l], "wb");
                bool ok = lf != nullptr;
                for (int t = 0; ok && t < T; t++) {
                    const std::string &text = part[(size_t)t * 3 + (size_t)l];
                    ok = fwrite(text.data(), 1, text.size(), lf) == text.size();
                }
                if (!ok || fclose(lf) != 0) die(2, (std::string("cannot write ") + list_name[l]).c_str());
            }
        }
        const double t_lists = now_s();
        // awk's three arrays as one map: a list line's first field under -F '#|/', first list wins (awk :12-16,23-35).  10M insertions of
        // std::string take seconds: built only when the host has to route something
        hq::ClassMap cls_of;
        bool cls_ready = false;
        auto need_cls = [&]() {
            if (cls_ready) return;
            for (int l = 0; l < 3; l++)
                for (size_t r = 0; r < nb; r++) {
                    const uint32_t i = order[r];
                    if (list_of[i] == l + 1) cls_of.emplace(std::string(hq::field(names[i], 0)), (uint8_t)(l + 1));
                }
            cls_ready = true;
        };
        bool any_sep = false;
        for (char c : has_sep) any_sep = any_sep || c;
        // On the GPU (default): the inputs go through the framer once more (.gz inputs inflated there again), a kernel sorts the records
        // of every block into four runs by the class of their barcode, the runs come back over PCIe and are written as they are -- no
        // host thread looks at a record.  On the host (--route host; --host-parse; a barcode that itself holds '#' or '/', whose list
        // line awk cuts short): the inputs are parsed again by the worker threads, quartering.h.
        const bool route_on_device = !host_parse && route_mode != "host" && !any_sep;
        uint64_t blocks_routed = 0, blocks_host = 0, bytes_routed = 0;
        double t_wait_write = 0, t_wait_gpu = 0;
        if (route_on_device) {
            // the table text -> class of every GPU
            std::vector<hast_names *> tabs(ctxs.size(), nullptr), own_tabs;
            {
                std::vector<uint8_t> text16;
                std::vector<uint32_t> cls;
                text16.reserve(nb * 16);
                cls.reserve(nb);
                for (size_t i = 0; i < nb; i++) {
                    const std::string_view bc = names[i];
                    if (bc.size() > 15) continue;                    // (a field that long is the host's: the kernel hands the block over)
                    uint8_t rec[16] = {0};
                    rec[0] = (uint8_t)bc.size();
                    memcpy(rec + 1, bc.data(), bc.size());
                    text16.insert(text16.end(), rec, rec + 16);
                    cls.push_back(list_of[i]);
                }
                for (size_t i = 0; i < ctxs.size(); i++) {
                    for (size_t j = 0; j < i && !tabs[i]; j++)
                        if (devices[j] == devices[i]) tabs[i] = tabs[j];
                    if (tabs[i]) continue;
                    CK(hast_names_create(ctxs[i], std::max<size_t>(cls.size(), 1024), &tabs[i]), "creating the routing table");
                    own_tabs.push_back(tabs[i]);
                    CK(hast_names_insert(tabs[i], text16.data(), cls.data(), cls.size()), "filling the routing table");
                }
            }
            struct WriteJob { const uint8_t *p[4]; size_t n[4]; std::shared_ptr<std::vector<std::string>> own; };
            struct RFeed {
                std::string name, prefix, log_name;
                size_t file_index = 0;
                hast::BlockSource src;
                hast_fq *fq = nullptr;
                hast_gz *gz = nullptr;
                std::thread th, wth[4];
                std::mutex mu, wmu;
                std::condition_variable cv, wcv;
                std::deque<std::pair<uint8_t *, hast_stream>> empty;
                struct Filled { size_t n; bool last; std::string err; };
                std::deque<Filled> filled;
                std::deque<WriteJob> wq[4];                             // per class: a writer thread per output file
                std::atomic<bool> write_failed{false};
                bool stop = false, eof_acquired = false, wstop = false, block_open = false;
                size_t held = 0, submitted = 0, opened = 0, acquired = 0, jobs = 0, written = 0;
                FILE *out[4] = {nullptr, nullptr, nullptr, nullptr};
                long long counts[5] = {0, 0, 0, 0, 0};
                bool any_input = false;
                std::string err_lines;
            };
            static const char *suffix[4] = {".nobarcode.fastq", ".paternal.fastq", ".maternal.fastq", ".homozygous.fastq"};
            const int n_buf = stripe ? std::max(2, (fq_bufs + (int)ctxs.size() - 1) / (int)ctxs.size()) : fq_bufs;
            std::mutex wake_mu;
            std::condition_variable wake_cv;
            uint64_t wake_gen = 0, seen_gen = 0;
            auto wake = [&] {
                { std::lock_guard<std::mutex> g(wake_mu); ++wake_gen; }
                wake_cv.notify_one();
            };
            std::vector<std::unique_ptr<RFeed>> active;
            std::vector<std::unique_ptr<RFeed>> finished(read.size());
            std::vector<hast_fq *> spare(done_fq);                      // the streams of the first pass, ready to be used again
            done_fq.clear();
            std::vector<std::thread> closers;
            size_t next_file = 0;
            // inputs with one basename write the same four files: the awk loop lets the later one overwrite the earlier -- one at a time then
            bool same_prefix = false;
            {
                std::vector<std::string> pf;
                for (const std::string &x : read) {
                    std::string nm = x.substr(x.find_last_of('/') == std::string::npos ? 0 : x.find_last_of('/') + 1);
                    if (nm.size() >= 3 && nm.compare(nm.size() - 3, 3, ".gz") == 0) nm.resize(nm.size() - 3);
                    same_prefix = same_prefix || std::find(pf.begin(), pf.end(), nm) != pf.end();
                    pf.push_back(nm);
                }
            }
            auto open_next = [&]() {
                const std::string &x = read[next_file];
                std::unique_ptr<RFeed> f(new RFeed());
                f->name = x;
                f->file_index = next_file;
                f->prefix = x.substr(x.find_last_of('/') == std::string::npos ? 0 : x.find_last_of('/') + 1);
                const bool gz_name = f->prefix.size() >= 3 && f->prefix.compare(f->prefix.size() - 3, 3, ".gz") == 0;   // (the wrapper's ${name: -3} == ".gz")
                if (gz_name) f->prefix.resize(f->prefix.size() - 3);
                f->log_name = gz_name ? "-" : x;                        // (awk's FILENAME behind `gzip -dc` is "-")
                if (dev_gz[next_file]) {
                    const hast_status gs = stripe ? hast_gz_open_multi(ctxs.data(), (int)ctxs.size(), x.c_str(), &f->gz) : hast_gz_open(ctxs[next_file % ctxs.size()], x.c_str(), &f->gz);
                    if (gs == HAST_ERR_UNSUPPORTED) {
                        f->gz = nullptr;
                        dev_gz[next_file] = 0;
                    } else if (gs != HAST_OK) die(2, ("cannot open " + x).c_str());
                }
                const size_t cap = cap_of(next_file);
                if (!f->gz) {
                    if (!f->src.open(x, cap, false)) die(2, ("cannot open " + x).c_str());
                    f->src.set_readers(std::max(4, std::min(16, t_num / (int)std::min<size_t>(read.size(), 2))));
                }
                // a stream of the first pass whose kind fits (device blocks or host blocks; on this file's context), or a new one
                const size_t want_ctx = next_file % ctxs.size();
                (void)want_ctx;
                CK(make_fq(next_file, &f->fq), "creating the FASTQ stream");
                std::vector<hast_names *> lane_tabs;
                if (stripe) lane_tabs = tabs;
                else lane_tabs.push_back(tabs[next_file % ctxs.size()]);
                CK(hast_fq_set_route(f->fq, lane_tabs.data(), (int)lane_tabs.size()), "switching the FASTQ stream to routing");
                next_file++;
                RFeed *fp = f.get();
                f->th = std::thread([fp, cap, &wake] {
                    for (;;) {
                        uint8_t *buf;
                        hast_stream fill_stream;
                        {
                            std::unique_lock<std::mutex> g(fp->mu);
                            fp->cv.wait(g, [fp] { return fp->stop || !fp->empty.empty(); });
                            if (fp->stop) return;
                            buf = fp->empty.front().first;
                            fill_stream = fp->empty.front().second;
                            fp->empty.pop_front();
                        }
                        RFeed::Filled fl{0, false, std::string()};
                        if (fp->gz) {
                            size_t n = 0;
                            if (hast_gz_read_device(fp->gz, buf, cap, &n, fill_stream) != HAST_OK) fl.err = hast_last_error();
                            else if (n < cap) {                      // (a short read: the end, or damage behind it -- the next call says which)
                                size_t more = 0;
                                if (hast_gz_read_device(fp->gz, buf + n, cap - n, &more, fill_stream) != HAST_OK) fl.err = hast_last_error();
                                n += more;
                            }
                            fl.n = n;
                        } else fl.n = fp->src.read_into(reinterpret_cast<char *>(buf), cap, fl.err);
                        fl.last = fl.n < cap || !fl.err.empty();
                        {
                            std::lock_guard<std::mutex> g(fp->mu);
                            fp->filled.push_back(fl);
                        }
                        wake();
                        if (fl.last) return;
                    }
                });
                for (int c = 0; c < 4; c++)
                    f->wth[c] = std::thread([fp, c, &wake] {            // the runs of class c to its file, in order (a thread per file: what
                        for (;;) {                                      // bounds the routing is the write, 17 GB per file at BASELINE config 2)
                            WriteJob j;
                            {
                                std::unique_lock<std::mutex> g(fp->wmu);
                                fp->wcv.wait(g, [fp, c] { return fp->wstop || !fp->wq[c].empty(); });
                                if (fp->wq[c].empty()) return;
                                j = std::move(fp->wq[c].front());
                                fp->wq[c].pop_front();
                            }
                            if (j.n[c] && !fp->write_failed.load()) {
                                if (!fp->out[c]) fp->out[c] = fopen((fp->prefix + suffix[c]).c_str(), "wb");
                                if (!fp->out[c] || fwrite(j.p[c], 1, j.n[c], fp->out[c]) != j.n[c]) fp->write_failed = true;
                            }
                            {
                                std::lock_guard<std::mutex> g(fp->wmu);
                                fp->written++;
                            }
                            wake();
                        }
                    });
                active.push_back(std::move(f));
            };
            // a block the device handed over: every record by the host's rules, in input order (quartering.h's classify)
            auto host_class = [&](std::string_view head, std::string &err) -> int {
                std::string_view f2 = hq::field(head, 1);
                if (f2.data() == nullptr || f2 == "0_0_0") return 0;
                need_cls();
                auto it = cls_of.find(std::string(f2));
                if (it != cls_of.end()) return it->second;
                err.append("ERROR : unclassify barcode : ").append(f2).append("\n");
                return -1;
            };
            auto open_block = [&](RFeed &f) {
                hast_fq_routed b;
                const double t0 = now_s();
                CK(hast_fq_next_routed(f.fq, &b), "routing a block");
                t_wait_gpu += now_s() - t0;
                WriteJob j;
                for (int c = 0; c < 4; c++) { j.p[c] = nullptr; j.n[c] = 0; }
                if (!b.host_block) {
                    for (int c = 0; c < 4; c++) {
                        j.p[c] = b.run[c];
                        j.n[c] = (size_t)b.run_bytes[c];
                        f.counts[c] += (long long)b.count[c];
                        bytes_routed += b.run_bytes[c];
                    }
                    f.counts[4] += (long long)b.n_records;
                    if (b.n_records) f.any_input = true;
                    blocks_routed++;
                } else {
                    j.own = std::make_shared<std::vector<std::string>>(4);
                    for (uint64_t i = 0; i < b.n_slots; i++) {
                        if (b.rec_class[i] == 0xFD) continue;
                        const char *r0 = reinterpret_cast<const char *>(b.bytes) + b.rec_start[i];
                        const size_t len = b.rec_len[i];
                        int c = b.rec_class[i];
                        if (c > 3) {
                            const void *nlp = memchr(r0, '\n', len);
                            c = host_class(std::string_view(r0, nlp ? (size_t)((const char *)nlp - r0) : len), f.err_lines);
                        }
                        f.counts[4]++;
                        f.any_input = true;
                        if (c >= 0) { (*j.own)[(size_t)c].append(r0, len); f.counts[c]++; }
                    }
                    for (int c = 0; c < 4; c++) { j.p[c] = reinterpret_cast<const uint8_t *>((*j.own)[(size_t)c].data()); j.n[c] = (*j.own)[(size_t)c].size(); }
                    blocks_host++;
                }
                if (b.tail_bytes) {
                    // the end of the file inside a record: awk still takes every remaining line as the record's (:21,41-49)
                    if (!j.own) {
                        j.own = std::make_shared<std::vector<std::string>>(4);
                        for (int c = 0; c < 4; c++) (*j.own)[(size_t)c].assign(reinterpret_cast<const char *>(j.p[c]), j.n[c]);
                    }
                    std::string_view rest(reinterpret_cast<const char *>(b.tail), (size_t)b.tail_bytes);
                    const size_t e = rest.find('\n');
                    const int c = host_class(rest.substr(0, e == std::string_view::npos ? rest.size() : e), f.err_lines);
                    f.counts[4]++;
                    f.any_input = true;
                    if (c >= 0) {
                        (*j.own)[(size_t)c].append(rest);
                        if (rest.back() != '\n') (*j.own)[(size_t)c].push_back('\n');
                        f.counts[c]++;
                    }
                    for (int c2 = 0; c2 < 4; c2++) { j.p[c2] = reinterpret_cast<const uint8_t *>((*j.own)[(size_t)c2].data()); j.n[c2] = (*j.own)[(size_t)c2].size(); }
                }
                {
                    std::lock_guard<std::mutex> g(f.wmu);
                    for (int c = 0; c < 4; c++) f.wq[c].push_back(j);
                    f.jobs += 4;
                }
                f.wcv.notify_all();
                f.block_open = true;
                f.opened++;
            };
            const size_t max_active = same_prefix ? 1 : (stripe ? 2 : std::max<size_t>(4, 2 * ctxs.size()));
            while (next_file < read.size() && active.size() < max_active) open_next();
            while (!active.empty()) {
                bool progress = false;
                for (size_t fi = 0; fi < active.size(); ++fi) {
                    RFeed &f = *active[fi];
                    while (!f.eof_acquired && f.held < (size_t)n_buf * (size_t)hast_fq_lanes(f.fq) && (!f.gz || f.acquired - f.submitted + 1 < (size_t)n_buf * (size_t)hast_fq_lanes(f.fq))) {
                        uint8_t *buf;
                        hast_stream fill_stream = nullptr;
                        CK(hast_fq_acquire(f.fq, &buf), "staging a block");
                        if (f.gz) CK(hast_fq_device_block(f.fq, &buf, &fill_stream), "staging a block");
                        f.held++;
                        f.acquired++;
                        std::lock_guard<std::mutex> g(f.mu);
                        f.empty.push_back({buf, fill_stream});
                        f.cv.notify_one();
                        progress = true;
                    }
                    for (;;) {
                        RFeed::Filled fl;
                        {
                            std::lock_guard<std::mutex> g(f.mu);
                            if (f.filled.empty()) break;
                            fl = f.filled.front();
                            f.filled.pop_front();
                        }
                        if (!fl.err.empty()) die(2, (f.name + ": " + fl.err).c_str());
                        CK(f.gz ? hast_fq_submit_device(f.fq, fl.n, fl.last ? 1 : 0) : hast_fq_submit(f.fq, fl.n, fl.last ? 1 : 0), "framing a block");
                        f.submitted++;
                        if (fl.last) f.eof_acquired = true;
                        progress = true;
                    }
                }
                for (size_t fi = 0; fi < active.size();) {
                    RFeed &f = *active[fi];
                    if (f.block_open) {                                // its runs written: the buffer goes back
                        bool done;
                        {
                            std::lock_guard<std::mutex> g(f.wmu);
                            done = f.written == f.jobs;
                        }
                        if (done) {
                            if (f.write_failed.load()) {
                                fprintf(stderr, "classify: cannot write %s.*.fastq\n", f.prefix.c_str());
                                fflush(stderr);
                                _exit(2);
                            }
                            CK(hast_fq_commit(f.fq), "releasing a block");
                            f.block_open = false;
                            f.held--;
                            progress = true;
                        }
                    }
                    if (!f.block_open && f.opened < f.submitted && hast_fq_poll(f.fq)) {
                        open_block(f);
                        progress = true;
                    }
                    if (f.eof_acquired && f.opened == f.submitted && !f.block_open) {
                        {
                            std::lock_guard<std::mutex> g(f.mu);
                            f.stop = true;
                        }
                        f.cv.notify_all();
                        f.th.join();
                        {
                            std::lock_guard<std::mutex> g(f.wmu);
                            f.wstop = true;
                        }
                        f.wcv.notify_all();
                        for (std::thread &w : f.wth) w.join();
                        for (FILE *&o : f.out)
                            if (o && fclose(o) != 0) { fprintf(stderr, "classify: cannot write %s.*.fastq\n", f.prefix.c_str()); fflush(stderr); _exit(2); }
                        if (f.gz) {
                            hast_gz *z = f.gz;
                            closers.emplace_back([z] { hast_gz_close(z); });
                        }
                        done_fq.push_back(f.fq);
                        const size_t idx = f.file_index;
                        finished[idx] = std::move(active[fi]);
                        active.erase(active.begin() + (long)fi);
                        if (next_file < read.size()) open_next();
                        progress = true;
                        continue;
                    }
                    ++fi;
                }
                if (!progress) {
                    const double t0 = now_s();
                    std::unique_lock<std::mutex> g(wake_mu);
                    wake_cv.wait_for(g, std::chrono::microseconds(stripe ? 10 : 100), [&] { return wake_gen != seen_gen; });
                    seen_gen = wake_gen;
                    t_wait_write += now_s() - t0;
                }
            }
            // stderr and filter_reads.log in the order of the inputs, as the wrapper's loop leaves them (awk :18-20,51-57)
            for (std::unique_ptr<RFeed> &fp : finished) {
                if (!fp) continue;
                fputs(fp->err_lines.c_str(), stderr);
                FILE *lg = fopen("filter_reads.log", "ab");
                if (lg) {
                    if (fp->any_input) fprintf(lg, "%s\n", fp->log_name.c_str());
                    fprintf(lg, "#Total reads                : %lld \n", fp->counts[4]);
                    fprintf(lg, "#Reads without barcode      : %lld \n", fp->counts[0]);
                    fprintf(lg, "#Paternal reads             : %lld \n", fp->counts[1]);
                    fprintf(lg, "#Maternal reads             : %lld \n", fp->counts[2]);
                    fprintf(lg, "#Homozygous reads           : %lld \n", fp->counts[3]);
                    fclose(lg);
                }
            }
            for (std::thread &t : closers) gz_closers.push_back(std::move(t));
            (void)spare;
            (void)own_tabs;
        } else {
        need_cls();
        // a .gz input inflated on the GPU, as a block source for the router: the bytes come back over PCIe block by block
        constexpr size_t kFrontPad = hast::BlockSource::kFrontPad;     // (room in front of a block's data: what route() expects)
        struct DevGzSource {
            hast_ctx *ctx = nullptr;
            hast_gz *gz = nullptr;
            void *d_buf = nullptr;
            size_t cap = 0;
            std::string err;
            std::vector<std::vector<char>> spare;
            ~DevGzSource() {
                if (gz) hast_gz_close(gz);
                if (d_buf) hast_dev_free(ctx, d_buf);
            }
            std::vector<char> next() {
                if (!err.empty()) return {};
                std::vector<char> blk;
                if (!spare.empty()) { blk = std::move(spare.back()); spare.pop_back(); }
                blk.resize(kFrontPad + cap);
                size_t n = 0;
                if (hast_gz_read_device(gz, static_cast<uint8_t *>(d_buf), cap, &n, nullptr) != HAST_OK ||
                    (n && hast_memcpy_d2h(ctx, blk.data() + kFrontPad, d_buf, n) != HAST_OK)) {
                    err = hast_last_error();
                    return {};
                }
                if (n == 0) return {};                       // (a short block is followed by another call: damage behind it is reported then)
                blk.resize(kFrontPad + n);
                return blk;
            }
            void recycle(std::vector<char> &&b) { spare.push_back(std::move(b)); }
            const std::string &error() const { return err; }
        };
        for (size_t fi = 0; fi < read.size(); fi++) {
            const std::string &x = read[fi];
            std::string name = x.substr(x.find_last_of('/') == std::string::npos ? 0 : x.find_last_of('/') + 1);
            const bool gz_name = name.size() >= 3 && name.compare(name.size() - 3, 3, ".gz") == 0;      // (the wrapper's ${name: -3} == ".gz")
            if (gz_name) name.resize(name.size() - 3);
            int rc;
            if (gz_name && dev_gz[fi]) {
                DevGzSource src;
                src.ctx = ctx;
                src.cap = 64u << 20;
                if (hast_gz_open(ctx, x.c_str(), &src.gz) != HAST_OK || hast_dev_alloc(ctx, src.cap, &src.d_buf) != HAST_OK) {
                    src.gz = nullptr;                        // (no room on the device, ...: the host inflates)
                    hast::BlockSource hsrc;
                    if (!hsrc.open(x, 64u << 20)) die(2, ("cannot open " + x).c_str());
                    rc = hq::route(name, cls_of, hsrc, "-", t_num, "classify");
                } else rc = hq::route(name, cls_of, src, "-", t_num, "classify");
            } else {
                hast::BlockSource hsrc;
                if (!hsrc.open(x, 64u << 20)) die(2, ("cannot open " + x).c_str());
                rc = hq::route(name, cls_of, hsrc, gz_name ? "-" : x, t_num, "classify");       // (awk's FILENAME behind `gzip -dc` is "-")
            }
            if (rc) {
                fprintf(stderr, "classify: ERROR: routing the reads of %s failed\n", x.c_str());
                fflush(stderr);
                _exit(rc);
            }
        }
        }
        for (const char *marker : {"step_10_done", "step_11_done"}) {
            FILE *mf = fopen(marker, "ab");                   // (the wrapper appends `date` to them and only tests that they exist)
            time_t now = time(0);
            if (!mf || fprintf(mf, "%s", ctime(&now)) < 0 || fclose(mf) != 0) die(2, (std::string("cannot write ") + marker).c_str());
        }
        if (stats)
            stat_line("__stats_phase_reads__ lists_and_routing_s=%.3f lists_s=%.3f routing_s=%.3f route=%s inputs=%zu blocks_routed_on_device=%llu blocks_routed_by_host=%llu "
                            "bytes_routed_on_device=%llu waiting_for_gpu_s=%.3f idle_s=%.3f\n",
                    now_s() - t_ph0, t_lists - t_ph0, now_s() - t_lists, route_on_device ? "device" : "host", read.size(), (unsigned long long)blocks_routed,
                    (unsigned long long)blocks_host, (unsigned long long)bytes_routed, t_wait_gpu, t_wait_write);
    }
    if (past_int)
        fprintf(stderr, " WARN : a barcode has more than INT_MAX hits: the reference's `int` counters overflow on this input; the exact counts were printed\n");
    logtime();
    if (stats) {
        double dt = t_classified - t_loaded;
        stat_line("__stats__ K=%zu set0=%llu set1=%llu reads=%llu bases=%llu barcodes=%zu load_s=%.3f classify_s=%.3f Mbp_per_s=%.1f\n",
                K, (unsigned long long)n_set[0], (unsigned long long)n_set[1], (unsigned long long)total_reads,
                (unsigned long long)total_bases, names.size(), t_loaded - t_start, dt, dt > 0 ? total_bases / dt / 1e6 : 0.0);
    }
    if (stats && naming.device_dict)
        stat_line("__stats_dictionary__ on=device dictionaries=%zu ids_from_device=%zu ids_from_host=%zu merge_by_text_s=%.3f\n", naming.groups.size(), n_dev_names, n_host_names, naming.merge_s);
    if (stats) stat_line("__stats_setup__ waited_for_stream_setup_s=%.3f (inside scrub_sizes_clone_s: .gz inputs opened, FASTQ streams created while the table was built)\n", t_pre_waited);
    // a context that could not get room for its filter probes the table directly (the round-1 kernel: 1.6 x the HBM requests per read):
    // same results, never silently
    for (size_t i = 0; i < ctxs.size(); i++) {
        char sw[512] = "";
        (void)hast_ctx_options(ctxs[i], sw, sizeof(sw));
        if (strstr(sw, "filter_fallback"))
            fprintf(stderr, " WARN : GPU %d (context %zu) had no room for the k-mer filter and probed the table directly (%s)\n", devices[i], i, sw);
    }
    if (stats) {
        char sw[512] = "";
        (void)hast_ctx_options(ctx, sw, sizeof(sw));          // measurement switches this context was created with (none by default)
        stat_line("__stats_switches__ %s\n", sw[0] ? sw : "none");
        int f_on = 0, f_m = 0, f_t = 0, f_kp = 0;
        uint64_t f_bytes = 0;
        (void)hast_filter_info(ctx, &f_on, &f_m, &f_t, &f_kp, &f_bytes);
        stat_line("__stats_filter__ mode=%s m=%d t=%d kp=%d bytes=%llu\n", f_on == 2 ? "exact_entries" : f_on == 1 ? "prints" : "off_table_only", f_m, f_t, f_kp,
                  (unsigned long long)f_bytes);
    }
    if (hbm_thread.joinable()) {
        hbm_stop = true;
        hbm_thread.join();
        size_t parked = 0;
        (void)hast_dev_mem_info(ctx, nullptr, nullptr, &parked);
        const size_t fm = hbm_free_min.load(), tot = hbm_total.load();
        if (tot) stat_line("__stats_hbm__ total_bytes=%zu free_min_bytes=%zu in_use_peak_bytes=%zu parked_bytes_at_end=%zu\n", tot, fm, tot - std::min(fm, tot), parked);
    }
    fprintf(stderr, "__END__\n");
    const double t_printed = now_s();
    // The output is complete.  Unpinning and freeing hundreds of MB of staging memory, the table and the streams takes ~0.1 s that the
    // operating system does anyway when the process ends: leave at once (HAST_TEARDOWN=1 runs the destructors, for leak checks) --
    // unless a profiler is listening (rocprofv3 preloads its tool library and writes its files when the process ends in an orderly way).
    const char *preload = getenv("LD_PRELOAD");
    const bool profiled = getenv("ROCP_TOOL_LIBRARIES") || (preload && strstr(preload, "rocprofiler"));
    if (!getenv("HAST_TEARDOWN") && !profiled) {
        if (stats)
            stat_line("__stats_phases__ gpu_context_s=%.3f load_kmers_s=%.3f scrub_sizes_clone_s=%.3f read_phase_s=%.3f counters_back_s=%.3f sort_print_s=%.3f teardown_s=skipped total_s=%.3f\n",
                    t_ctx - t_start, t_loaded - t_ctx, t_scrubbed - t_loaded, t_read_done - t_scrubbed, t_classified - t_read_done, t_printed - t_classified, now_s() - t_start);
        if (!stats_json.empty() && !write_stats_json(stats_json)) fprintf(stderr, "classify: cannot write %s\n", stats_json.c_str());
        if (fflush(stdout) != 0) die_output();
        fflush(stderr);
        _exit(0);
    }
    for (std::thread &t : gz_closers) t.join();
    for (hast_fq *f : done_fq) hast_fq_destroy(f);
    for (hast_names *nm : own_caches) hast_names_destroy(nm);
    for (hast_ctx *c : ctxs) hast_ctx_destroy(c);
    if (stats)                 // where a run's wall time goes, phase by phase (sums to the process's own lifetime from main() on)
        stat_line("__stats_phases__ gpu_context_s=%.3f load_kmers_s=%.3f scrub_sizes_clone_s=%.3f read_phase_s=%.3f counters_back_s=%.3f sort_print_s=%.3f teardown_s=%.3f total_s=%.3f\n",
                t_ctx - t_start, t_loaded - t_ctx, t_scrubbed - t_loaded, t_read_done - t_scrubbed, t_classified - t_read_done,
                t_printed - t_classified, now_s() - t_printed, now_s() - t_start);
    if (!stats_json.empty() && !write_stats_json(stats_json)) fprintf(stderr, "classify: cannot write %s\n", stats_json.c_str());
    return 0;
}

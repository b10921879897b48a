// classify_read_main.cpp -- drop-in for the per-read classifier of HAST stage 03 on MI355X
// (reference: /root/reference/03.mkoutput_by_fabulous2.0/src_main/classify.cpp, cited s03:N; usage name
// "classify_read", s03:305).  Same flags (--hap F --hap F --read F [--read F] [--thread N] [--format fasta|fastq],
// s03:316-358) and the same stdout rows "name \t haplotypeN|ambiguous \t density" in read order (s03:104-135).
// K-mer lookups run on the GPU through include/hast.h (hast_classify_perread); this file does what the
// reference does on the host around them: flag parsing, FASTA/FASTQ framing (s03:248-302), the density and
// the call (s03:110-133, 215-216).
//
// Additive flags: --device N, --devices A,B,... (several GPUs of this node: the table is built on the first one and copied to
// the others over xGMI; every batch of reads is cut into one contiguous share per GPU, balanced by bases, classified at the same
// time and printed in read order -- per-read rows need no reduction, BASELINE config 5's "barcode-free per-read assignment").
//
// Requirement (deviation): k-mer lines must be upper-case A/C/G/T of one length K <= 32 -- what jellyfish/meryl
// dumps are.  The reference would also store other bytes literally (s03:59-65); we stop with exit 3 instead.
#include <getopt.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <string_view>
#include <vector>

#include "../../include/hast.h"
#include "ingest.h"

namespace {

[[noreturn]] void die(int code, const std::string &what) {
    fprintf(stderr, "classify_read: ERROR: %s", what.c_str());
    const char *e = hast_last_error();
    if (e && *e) fprintf(stderr, " (%s)", e);
    fputc('\n', stderr);
    // (no destructors: a batch may be on the GPU in the background thread, and exit() would take the HIP runtime down under it)
    fflush(stdout);
    fflush(stderr);
    _exit(code);
}

void print_usage() {   // same flags as the reference (s03:316-324); stderr is free-form
    fputs("classify_read (MI355X) -- per-read haplotype assignment\n"
          "  classify_read --hap HAP0.mer --hap HAP1.mer --read READS [--read ...] [--format fasta|fastq] [--thread N]\n"
          "  reads may be gzip files when the name ends in .gz; --format defaults to fasta\n"
          "  --device N / --devices A,B,..   GPU(s); with several, every batch of reads is shared out between them\n",
          stderr);
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

}  // namespace

int main(int argc, char **argv) {
    static struct option long_options[] = {{"hap", required_argument, NULL, 'p'},    {"read", required_argument, NULL, 'r'},
                                           {"format", required_argument, NULL, 'f'}, {"thread", required_argument, NULL, 't'},
                                           {"help", no_argument, NULL, 'h'},         {"device", required_argument, NULL, 1001},
                                           {"devices", required_argument, NULL, 1002}, {0, 0, 0, 0}};
    std::vector<std::string> haps, read;
    std::string format = "fasta";
    int t_num = 8, device = 0;
    std::vector<int> devices;
    for (;;) {
        int c = getopt_long(argc, argv, "p:r:t:f:h", long_options, NULL);   // s03:324
        if (c < 0) break;
        switch (c) {
        case 'p': haps.push_back(optarg); break;
        case 'r': read.push_back(optarg); break;
        case 'f': format = optarg; break;
        case 't': t_num = atoi(optarg); break;
        case 1001: device = atoi(optarg); break;
        case 1002:
            for (const char *q = optarg; *q;) {
                char *end;
                const long v = strtol(q, &end, 10);
                if (end == q || v < 0 || (*end && *end != ',')) { print_usage(); return -1; }
                devices.push_back((int)v);
                q = *end == ',' ? end + 1 : end;
            }
            break;
        case 'h':
        default: print_usage(); return -1;
        }
    }
    if (haps.size() != 2 || read.empty() || t_num < 1) {                     // s03:351-354
        print_usage();
        return -1;
    }
    if (devices.empty()) devices.push_back(device);
    device = devices[0];
    if (format != "fasta" && format != "fastq") {
        fprintf(stderr, " ERROR : invalid format : [%s] . exit ...\n", format.c_str());
        return -1;
    }
    fprintf(stderr, "__START__\n");
    // ---- load_kmers (s03:51-70) --------------------------------------------------------------------
    // K = length of the first line of the first file (s03:57-58); regular files are streamed into the table by the library,
    // which also checks on the device that every line is K upper-case A/C/G/T bytes; anything else is read into memory first
    // (a pipe can be read only once: it is read whole now and its first bytes serve as the head)
    std::vector<char> txt[2], head;
    bool streamed[2] = {false, false};
    size_t text_bytes[2] = {0, 0};
    for (int h = 0; h < 2; h++) {
        struct stat sb;
        if (stat(haps[h].c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) {
            streamed[h] = true;
            text_bytes[h] = (size_t)sb.st_size;
        } else {
            if (!slurp(haps[h], txt[h])) die(2, "cannot read " + haps[h]);
            text_bytes[h] = txt[h].size();
        }
    }
    if (streamed[0]) {
        FILE *f = fopen(haps[0].c_str(), "rb");
        if (!f) die(2, "cannot read " + haps[0]);
        head.resize(4096);
        head.resize(fread(head.data(), 1, head.size(), f));
        fclose(f);
    } else head.assign(txt[0].begin(), txt[0].begin() + (long)std::min<size_t>(txt[0].size(), 4096));
    const void *nl0 = memchr(head.data(), '\n', head.size());
    const size_t K = nl0 ? (size_t)((const char *)nl0 - head.data()) : head.size();   // s03:57-58
    if (K < 1 || K > 32) die(3, "K (length of the first k-mer line) must be in [1,32]");
    hast_ctx *ctx = nullptr;
    if (hast_ctx_create(device, (int)K, &ctx) != HAST_OK) die(4, "cannot create GPU context");
    if (hast_ctx_set_text_check(ctx, 1) != HAST_OK) die(4, "cannot create GPU context");
    if (hast_table_reserve(ctx, text_bytes[0] / (K + 1) + text_bytes[1] / (K + 1) + 2, 0.0) != HAST_OK) die(4, "allocating the k-mer table");
    int total_kmers[2] = {0, 0};                                             // s03:50,68 (an `int` in the reference)
    for (int h = 0; h < 2; h++) {
        fprintf(stderr, "__load hap%d kmers__\n", h);
        uint64_t lines = 0;
        hast_status st;
        if (streamed[h]) st = hast_table_insert_text_file(ctx, h, haps[h].c_str(), &lines);
        else {
            // complete lines only: a trailing piece without '\n' is dropped (s03:63) -- except a lone first line
            size_t usable = txt[h].size();
            while (usable && txt[h][usable - 1] != '\n') usable--;
            st = hast_table_insert_text(ctx, h, txt[h].data(), usable, &lines);
        }
        if (st == HAST_ERR_FORMAT) die(3, std::string(hast_last_error()) + " (k-mer files must hold one K-mer of upper-case A/C/G/T per line)");
        if (st == HAST_ERR_IO) die(2, "cannot read " + haps[h]);
        if (st != HAST_OK) die(4, "building the k-mer table");
        if (h == 0 && !nl0 && text_bytes[0] == K) {
            for (size_t i = 0; i < K; i++)
                if (!strchr("ACGT", head[i])) die(3, "k-mer files must hold upper-case A/C/G/T lines only");
            uint64_t key = hast_canon_kmer(head.data(), (int)K);
            if (hast_table_insert_keys(ctx, 0, &key, 1) != HAST_OK) die(4, "building the k-mer table");
            lines = 1;
        }
        total_kmers[h] = (int)lines;
        fprintf(stderr, "Recorded %d haplotype %d specific %zu-mers\n", total_kmers[h], h, K);
        std::vector<char>().swap(txt[h]);
    }
    // the other GPUs get a copy of the finished table, peer to peer
    std::vector<hast_ctx *> ctxs{ctx};
    for (size_t i = 1; i < devices.size(); i++) {
        hast_ctx *c2 = nullptr;
        if (hast_ctx_create(devices[i], (int)K, &c2) != HAST_OK) die(4, "cannot create GPU context");
        ctxs.push_back(c2);
    }
    if (ctxs.size() > 1) {
        // (all at once: every GPU pulls its copy over its own xGMI link; the source's filter is built first, the clones only read the source)
        if (hast_filter_build(ctx) != HAST_OK) die(4, "building the k-mer filter");
        std::vector<std::thread> cloners;
        std::vector<int> failed(ctxs.size(), 0);
        for (size_t i = 1; i < ctxs.size(); i++)
            cloners.emplace_back([&, i] { failed[i] = hast_table_clone(ctxs[i], ctx) != HAST_OK; });
        for (std::thread &t : cloners) t.join();
        for (int f : failed)
            if (f) die(4, "copying the k-mer table to another GPU");
    }
    // ---- reads: block ingest with t_num parser threads (ingest.h), one GPU batch per block ------------------------
    // Framing as in the reference: FASTQ = 4 getlines per record, the header must be newline-terminated (s03:255-263);
    // FASTA = '>' lines start a record, other non-empty lines are appended, a last line without '\n' is dropped
    // (s03:279-296).  Rows of a file are printed when the file is done, like PrintOutput after the loop (s03:269,301).
    hast::WorkerPool pool(t_num);
    const int T = pool.size();
    std::vector<std::vector<uint32_t>> nl(T);
    std::vector<uint32_t> allnl;
    std::vector<uint8_t> bases;
    std::vector<uint64_t> offsets;
    std::vector<std::string> names;
    std::vector<uint64_t> part(T + 1);
    std::vector<int> bad(T);
    std::string out_rows;
    const bool fastq = format == "fastq";
    // Classify the parsed batch (names/offsets/bases) and append its rows.  The batch is handed to ONE background thread (rows
    // stay in order) so that the upload, the kernel and the formatting of block i run while block i+1 is indexed and copied.
    struct Batch {
        std::vector<uint8_t> bases;
        std::vector<uint64_t> offsets;
        std::vector<std::string> names;
    };
    std::thread bg;
    struct Joiner {                                         // every way out of main() waits for the batch in flight
        std::thread &t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{bg};
    Batch bt[2];                                            // the batch in flight and the one before it (its buffers are reused)
    int cur = 0;
    std::string bg_error;
    auto wait_bg = [&]() {
        if (bg.joinable()) bg.join();
        if (!bg_error.empty()) die(4, bg_error);
    };
    auto run_batch = [&](Batch &b) {
        if (b.names.empty()) return;
        const size_t n = b.names.size(), G = ctxs.size();
        std::vector<uint32_t> v(n * 2, 0);
        if (G == 1) {
            if (hast_classify_perread(ctx, b.bases.data(), b.offsets.data(), n, v.data()) != HAST_OK) {
                bg_error = std::string("classifying a batch (") + hast_last_error() + ")";
                return;
            }
        } else {
            // one contiguous share of the reads per GPU, cut where the bases divide evenly (reads differ in length by orders of
            // magnitude); the shares run at the same time, every one on a thread of its own (a context is driven by one thread)
            std::vector<size_t> cut(G + 1, n);
            cut[0] = 0;
            const uint64_t total = b.offsets[n] - b.offsets[0];
            for (size_t g = 1; g < G; g++) {
                const uint64_t want = b.offsets[0] + total * g / G;
                cut[g] = (size_t)(std::lower_bound(b.offsets.begin(), b.offsets.begin() + (long)n, want) - b.offsets.begin());
                if (cut[g] < cut[g - 1]) cut[g] = cut[g - 1];
            }
            std::vector<std::string> errs(G);
            std::vector<std::thread> th;
            for (size_t g = 0; g < G; g++)
                th.emplace_back([&, g] {
                    const size_t lo = cut[g], hi = cut[g + 1];
                    if (hi > lo && hast_classify_perread(ctxs[g], b.bases.data(), b.offsets.data() + lo, hi - lo, v.data() + 2 * lo) != HAST_OK)
                        errs[g] = hast_last_error();
                });
            for (std::thread &t : th) t.join();
            for (const std::string &e : errs)
                if (!e.empty()) {
                    bg_error = "classifying a batch (" + e + ")";
                    return;
                }
        }
        char row[96];
        for (size_t i = 0; i < b.names.size(); i++) {
            const double d0 = (double)v[2 * i] / total_kmers[0], d1 = (double)v[2 * i + 1] / total_kmers[1];   // s03:215-216
            // s03:110-133 for two haplotypes: both zero -> ambiguous 0.0; else the larger density, ties to haplotype0
            if (!(d0 > 0) && !(d1 > 0)) snprintf(row, sizeof(row), "\tambiguous\t0.0\n");
            else if (d1 > d0) snprintf(row, sizeof(row), "\thaplotype1\t%0.6f\n", d1);
            else snprintf(row, sizeof(row), "\thaplotype0\t%0.6f\n", d0);
            out_rows += b.names[i];
            out_rows += row;
        }
    };
    auto classify_batch = [&]() {
        wait_bg();                                          // one batch in flight; its rows are in out_rows now
        if (names.empty()) return;
        Batch &b = bt[cur];
        b.bases.swap(bases);
        b.offsets.swap(offsets);
        b.names.swap(names);
        bg = std::thread([&run_batch, &b] { run_batch(b); });
        cur ^= 1;
        Batch &o = bt[cur];                                 // idle since the wait above: its vectors (and their capacity) are the next work area
        bases.swap(o.bases);
        offsets.swap(o.offsets);
        names.swap(o.names);
    };
    // records = [first, last) pairs of line indices: header line, then sequence lines; fills names/offsets/bases in parallel
    struct RecSpan { uint32_t head, seq_first, seq_end; };             // line indices; sequence lines [seq_first, seq_end)
    std::vector<RecSpan> recs;
    auto line_start = [&](size_t i) -> size_t { return i ? (size_t)allnl[i - 1] + 1 : 0; };
    auto build_batch = [&](const char *data) {
        const size_t n = recs.size();
        names.assign(n, std::string());
        offsets.assign(n + 1, 0);
        if (n == 0) { bases.clear(); return; }
        std::vector<uint64_t> len(n);
        pool.run([&](int t) {
            uint64_t sum = 0;
            for (size_t i = n * (size_t)t / T; i < n * (size_t)(t + 1) / T; i++) {
                const RecSpan &r = recs[i];
                uint64_t l = 0;
                if (r.seq_end > r.seq_first) l = ((uint64_t)allnl[r.seq_end - 1] - line_start(r.seq_first)) - (r.seq_end - r.seq_first - 1);
                len[i] = l;
                sum += l;
            }
            part[t + 1] = sum;
        });
        part[0] = 0;
        for (int t = 0; t < T; t++) part[t + 1] += part[t];
        bases.resize(part[T] + 16);
        pool.run([&](int t) {
            uint64_t off = part[t];
            for (size_t i = n * (size_t)t / T; i < n * (size_t)(t + 1) / T; i++) {
                const RecSpan &r = recs[i];
                const size_t hs = line_start(r.head), he = allnl[r.head];
                names[i].assign(data + hs + (he > hs ? 1 : 0), data + he);               // head.substr(1), s03:207
                offsets[i] = off;
                for (uint32_t li = r.seq_first; li < r.seq_end; li++) {
                    const size_t ls = line_start(li), le = allnl[li];
                    memcpy(bases.data() + off, data + ls, le - ls);
                    off += le - ls;
                }
            }
        });
        offsets[n] = part[T];
    };
    const size_t block_bytes = (size_t)std::max(1L, getenv("HAST_READ_BLOCK_BYTES") ? atol(getenv("HAST_READ_BLOCK_BYTES")) : (256L << 20));
    bool output_failed = false;
    for (const auto &r : read) {
        fprintf(stderr, "__process read: %s\n", r.c_str());
        hast::BlockSource src;
        if (!src.open(r, block_bytes)) die(2, "cannot open " + r);
        std::vector<char> carry;
        wait_bg();
        out_rows.clear();
        bool seen_header = false;                       // FASTA: lines before the first '>' belong to no record (s03:286-292)
        for (;;) {
            std::vector<char> blk = src.next();
            const bool last = blk.empty();
            if (last && !src.error().empty()) die(2, r + ": " + src.error());
            constexpr size_t kPad = hast::BlockSource::kFrontPad;
            const char *data;
            size_t len;
            if (last) { data = carry.data(); len = carry.size(); }
            else if (carry.size() <= kPad) {
                if (!carry.empty()) memcpy(blk.data() + kPad - carry.size(), carry.data(), carry.size());
                data = blk.data() + kPad - carry.size();
                len = blk.size() - kPad + carry.size();
            } else {
                carry.insert(carry.end(), blk.begin() + kPad, blk.end());
                data = carry.data();
                len = carry.size();
            }
            if (len >= (1ull << 32)) die(3, "a single record spans more than 4 GB");
            pool.run([&](int t) {
                auto &v = nl[t];
                v.clear();
                const char *p = data + len * (size_t)t / T, *e = data + len * (size_t)(t + 1) / T;
                while (p < e && (p = (const char *)memchr(p, '\n', (size_t)(e - p)))) { v.push_back((uint32_t)(p - data)); ++p; }
            });
            allnl.clear();
            for (int t = 0; t < T; t++) allnl.insert(allnl.end(), nl[t].begin(), nl[t].end());
            const size_t n_lines = allnl.size();                 // complete ('\n'-terminated) lines in the work area
            recs.clear();
            size_t consumed = 0;
            if (fastq) {
                const size_t n_rec = n_lines / 4;
                for (size_t i = 0; i < n_rec; i++) recs.push_back({(uint32_t)(4 * i), (uint32_t)(4 * i + 1), (uint32_t)(4 * i + 2)});
                consumed = n_rec ? (size_t)allnl[4 * n_rec - 1] + 1 : 0;
                if (last && n_lines % 4 >= 1) {
                    // tail: header terminated; bases = next piece, terminated or not (s03:260); make the tail a full record
                    std::string tail(data + consumed, len - consumed);
                    size_t he = tail.find('\n');
                    std::string head = tail.substr(0, he), seq;
                    size_t s0 = he + 1, se = tail.find('\n', s0);
                    seq = tail.substr(s0, se == std::string::npos ? std::string::npos : se - s0);
                    // handled after the block's records, as a one-record batch below
                    for (const RecSpan &x : recs)
                        if (data[line_start(x.head)] == '>') { fprintf(stderr, "fasta detected . ERROR . please use \"--format fasta\". exit ... \n"); return 1; }
                    build_batch(data);
                    classify_batch();
                    if (!head.empty() && head[0] == '>') { fprintf(stderr, "fasta detected . ERROR . please use \"--format fasta\". exit ... \n"); return 1; }
                    names.assign(1, head.empty() ? head : head.substr(1));
                    bases.assign(seq.begin(), seq.end());
                    bases.resize(bases.size() + 16);
                    offsets.assign({0, (uint64_t)seq.size()});
                    classify_batch();
                    break;
                }
                for (const RecSpan &x : recs)
                    if (len && data[line_start(x.head)] == '>') { fprintf(stderr, "fasta detected . ERROR . please use \"--format fasta\". exit ... \n"); return 1; }
            } else {
                // header lines (first byte '>'); '@'/'+' at the start of a non-empty line is the reference's format error
                std::vector<std::vector<uint32_t>> hl(T);
                pool.run([&](int t) {
                    bad[t] = 0;
                    hl[t].clear();
                    for (size_t i = n_lines * (size_t)t / T; i < n_lines * (size_t)(t + 1) / T; i++) {
                        const size_t ls = line_start(i);
                        if (allnl[i] == ls) continue;                                       // empty line: skipped (s03:280)
                        const char c = data[ls];
                        if (c == '>') hl[t].push_back((uint32_t)i);
                        else if (c == '@' || c == '+') bad[t] = 1;
                    }
                });
                std::vector<uint32_t> heads;
                for (int t = 0; t < T; t++) {
                    if (bad[t]) { fprintf(stderr, "fasta detected . ERROR . please use \"--format fastq\". exit ... \n"); return 1; }
                    heads.insert(heads.end(), hl[t].begin(), hl[t].end());
                }
                // complete records: header j .. header j+1; at EOF the last header's record ends at the last complete line
                for (size_t j = 0; j + 1 < heads.size(); j++) recs.push_back({heads[j], heads[j] + 1, heads[j + 1]});
                if (last && !heads.empty()) recs.push_back({heads.back(), heads.back() + 1, (uint32_t)n_lines});
                if (!heads.empty()) seen_header = true;
                if (last) consumed = len;
                else if (!heads.empty()) consumed = line_start(heads.back());      // carry the (possibly incomplete) last record
                else consumed = seen_header ? 0 : (n_lines ? (size_t)allnl[n_lines - 1] + 1 : 0);   // no header yet: drop complete lines
            }
            build_batch(data);
            classify_batch();
            if (last) break;
            std::vector<char> keep(data + consumed, data + len);
            carry.swap(keep);
            src.recycle(std::move(blk));
        }
        wait_bg();
        if (fwrite(out_rows.data(), 1, out_rows.size(), stdout) != out_rows.size()) output_failed = true;
        fprintf(stderr, "__process read done__\n");
    }
    // (mkoutput_by_fabulous2.0.sh redirects stdout into a file and goes on: a full disk or a closed pipe must not pass as exit 0)
    if (fflush(stdout) != 0) output_failed = true;
    if (output_failed) {
        fprintf(stderr, "classify: ERROR: writing the result to stdout failed (%s)\n", strerror(errno));
        return 2;
    }
    fprintf(stderr, "__END__\n");
    for (hast_ctx *c : ctxs) hast_ctx_destroy(c);
    return 0;
}

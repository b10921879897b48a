// classify_read_main.cpp -- drop-in for the per-read classifier of HAST stage 03 on MI355X
// (reference: /root/reference/03.mkoutput_by_fabulous2.0/src_main/classify.cpp, cited s03:N; usage name
// "classify_read", s03:305).  Same flags (--hap F --hap F --read F [--read F] [--thread N] [--format fasta|fastq],
// s03:316-358) and the same stdout rows "name \t haplotypeN|ambiguous \t density" in read order (s03:104-135).
// K-mer lookups run on the GPU through include/hast.h (hast_classify_perread); this file does what the
// reference does on the host around them: flag parsing, FASTA/FASTQ framing (s03:248-302), the density and
// the call (s03:110-133, 215-216).
//
// Requirement (deviation): k-mer lines must be upper-case A/C/G/T of one length K <= 31 -- what jellyfish/meryl
// dumps are.  The reference would also store other bytes literally (s03:59-65); we stop with exit 3 instead.
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <vector>

#include "../../include/hast.h"
#include "fastq_reader.h"

namespace {

[[noreturn]] void die(int code, const std::string &what) {
    fprintf(stderr, "classify_read: ERROR: %s", what.c_str());
    const char *e = hast_last_error();
    if (e && *e) fprintf(stderr, " (%s)", e);
    fputc('\n', stderr);
    exit(code);
}

void print_usage() {   // same flags as the reference (s03:316-324); stderr is free-form
    fputs("classify_read (MI355X) -- per-read haplotype assignment\n"
          "  classify_read --hap HAP0.mer --hap HAP1.mer --read READS [--read ...] [--format fasta|fastq] [--thread N]\n"
          "  reads may be gzip files when the name ends in .gz; --format defaults to fasta\n",
          stderr);
}

bool slurp(const std::string &path, std::vector<char> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(sz > 0 ? (size_t)sz : 0);
    bool ok = sz <= 0 || fread(out.data(), 1, (size_t)sz, f) == (size_t)sz;
    fclose(f);
    return ok;
}

struct Batch {
    std::vector<std::string> names;
    std::vector<uint8_t> bases;
    std::vector<uint64_t> offsets{0};
    void add(std::string_view head, std::string_view seq) {
        names.emplace_back(head.empty() ? head : head.substr(1));          // s03:207 head.substr(1)
        bases.insert(bases.end(), seq.begin(), seq.end());
        offsets.push_back(bases.size());
    }
    void clear() {
        names.clear();
        bases.clear();
        offsets.assign(1, 0);
    }
};

// One output row from a read's integer hits.  Densities are hits / line count of the file (s03:68,215-216).  The
// reference's selection loop (s03:110-133) reduces, for two haplotypes, to: both zero -> "ambiguous 0.0"; otherwise
// the larger density wins and a tie goes to haplotype0 (its strict comparisons never replace the first maximum).
void print_row(const std::string &name, uint32_t h0, uint32_t h1, const int total_kmers[2]) {
    const double d0 = (double)h0 / total_kmers[0], d1 = (double)h1 / total_kmers[1];
    if (!(d0 > 0) && !(d1 > 0)) printf("%s\tambiguous\t0.0\n", name.c_str());
    else if (d1 > d0) printf("%s\thaplotype1\t%0.6f\n", name.c_str(), d1);
    else printf("%s\thaplotype0\t%0.6f\n", name.c_str(), d0);
}

}  // namespace

int main(int argc, char **argv) {
    static struct option long_options[] = {{"hap", required_argument, NULL, 'p'},    {"read", required_argument, NULL, 'r'},
                                           {"format", required_argument, NULL, 'f'}, {"thread", required_argument, NULL, 't'},
                                           {"help", no_argument, NULL, 'h'},         {"device", required_argument, NULL, 1001},
                                           {0, 0, 0, 0}};
    std::vector<std::string> haps, read;
    std::string format = "fasta";
    int t_num = 8, device = 0;
    for (;;) {
        int c = getopt_long(argc, argv, "p:r:t:f:h", long_options, NULL);   // s03:324
        if (c < 0) break;
        switch (c) {
        case 'p': haps.push_back(optarg); break;
        case 'r': read.push_back(optarg); break;
        case 'f': format = optarg; break;
        case 't': t_num = atoi(optarg); break;
        case 1001: device = atoi(optarg); break;
        case 'h':
        default: print_usage(); return -1;
        }
    }
    if (haps.size() != 2 || read.empty() || t_num < 1) {                     // s03:351-354
        print_usage();
        return -1;
    }
    if (format != "fasta" && format != "fastq") {
        fprintf(stderr, " ERROR : invalid format : [%s] . exit ...\n", format.c_str());
        return -1;
    }
    fprintf(stderr, "__START__\n");
    // ---- load_kmers (s03:51-70) --------------------------------------------------------------------
    std::vector<char> txt[2];
    for (int h = 0; h < 2; h++)
        if (!slurp(haps[h], txt[h])) die(2, "cannot read " + haps[h]);
    const void *nl0 = memchr(txt[0].data(), '\n', txt[0].size());
    const size_t K = nl0 ? (size_t)((const char *)nl0 - txt[0].data()) : txt[0].size();   // s03:57-58
    if (K < 1 || K > 31) die(3, "K (length of the first k-mer line) must be in [1,31]");
    for (int h = 0; h < 2; h++)
        for (char ch : txt[h])
            if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T' && ch != '\n') die(3, "k-mer files must hold upper-case A/C/G/T lines only");
    hast_ctx *ctx = nullptr;
    if (hast_ctx_create(device, (int)K, &ctx) != HAST_OK) die(4, "cannot create GPU context");
    if (hast_table_reserve(ctx, txt[0].size() / (K + 1) + txt[1].size() / (K + 1) + 2, 0.0) != HAST_OK) die(4, "allocating the k-mer table");
    int total_kmers[2] = {0, 0};                                             // s03:50,68 (an `int` in the reference)
    for (int h = 0; h < 2; h++) {
        fprintf(stderr, "__load hap%d kmers__\n", h);
        uint64_t lines = 0;
        // complete lines only: a trailing piece without '\n' is dropped (s03:63) -- except a lone first line
        size_t usable = txt[h].size();
        while (usable && txt[h][usable - 1] != '\n') usable--;
        hast_status st = hast_table_insert_text(ctx, h, txt[h].data(), usable, &lines);
        if (st == HAST_ERR_FORMAT) die(3, "k-mer file is not one K-mer per line");
        if (st != HAST_OK) die(4, "building the k-mer table");
        if (h == 0 && !nl0 && txt[0].size() == K) {
            uint64_t key = hast_canon_kmer(txt[0].data(), (int)K);
            if (hast_table_insert_keys(ctx, 0, &key, 1) != HAST_OK) die(4, "building the k-mer table");
            lines = 1;
        }
        total_kmers[h] = (int)lines;
        fprintf(stderr, "Recorded %d haplotype %d specific %zu-mers\n", total_kmers[h], h, K);
    }
    // ---- reads ------------------------------------------------------------------------------------------
    Batch batch;
    std::vector<uint32_t> votes;
    auto flush = [&]() {
        if (batch.names.empty()) return;
        votes.assign(batch.names.size() * 2, 0);
        if (hast_classify_perread(ctx, batch.bases.data(), batch.offsets.data(), batch.names.size(), votes.data()) != HAST_OK)
            die(4, "classifying a batch");
        for (size_t i = 0; i < batch.names.size(); i++) print_row(batch.names[i], votes[2 * i], votes[2 * i + 1], total_kmers);
        batch.clear();
    };
    const size_t kBatchBytes = 256u << 20;
    for (const auto &r : read) {
        fprintf(stderr, "__process read: %s\n", r.c_str());
        hast::LineSource in;
        if (!in.open(r)) die(2, "cannot open " + r);
        bool eof;
        if (format == "fastq") {                                             // s03:248-270
            for (;;) {
                std::string head(in.getline(eof));
                if (eof) break;
                if (!head.empty() && head[0] == '>') {
                    fprintf(stderr, "fasta detected . ERROR . please use \"--format fasta\". exit ... \n");
                    return 1;
                }
                std::string_view seq = in.getline(eof);
                batch.add(head, seq);
                in.getline(eof);
                in.getline(eof);
                if (batch.bases.size() >= kBatchBytes) flush();
            }
        } else {                                                             // s03:272-302
            std::string head, seq;
            long long id = 0;
            for (;;) {
                std::string_view tmp = in.getline(eof);
                if (eof) break;
                if (tmp.empty()) continue;
                if (tmp[0] == '@' || tmp[0] == '+') {
                    fprintf(stderr, "fasta detected . ERROR . please use \"--format fastq\". exit ... \n");
                    return 1;
                }
                if (tmp[0] == '>') {
                    if (id > 0) {
                        batch.add(head, seq);
                        if (batch.bases.size() >= kBatchBytes) flush();
                    }
                    head.assign(tmp);
                    seq.clear();
                    id++;
                } else {
                    seq.append(tmp);
                }
            }
            if (id > 0) batch.add(head, seq);                                // s03:297 (the reference crashes on an empty file)
        }
        flush();                                                             // rows of one file are printed before the next (s03:269,301)
        fprintf(stderr, "__process read done__\n");
    }
    fflush(stdout);
    fprintf(stderr, "__END__\n");
    hast_ctx_destroy(ctx);
    return 0;
}

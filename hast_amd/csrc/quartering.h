// quartering.h -- the routing step right after `classify` in HAST stage 01, shared by the stand-alone program
// (quartering_main.cpp) and by `classify` itself (HAST_PHASE_READS=1: steps 10 and 11 of the wrapper done by the program that has the
// barcodes' classes in memory and the GPU's inflate at hand): every FASTQ record to <prefix>.{paternal,maternal,homozygous,nobarcode}.fastq
// by the barcode lists, which the reference does with single-threaded awk
// (/root/reference/01.classify_stlfr_reads/quartering_fastq.awk, invoked at classify_stlfr_reads.sh:176-185).
//
// Same outputs byte for byte: the four FASTQ files (created only when something is routed to them), the
// "ERROR : unclassify barcode" lines on stderr (those records are dropped, awk :31-34) and the lines appended
// to filter_reads.log (awk :18-20, :51-57).  Semantics restated from the awk program: fields are split at '#'
// or '/' (-F '#|/'); a list line contributes its first field (:12-16); a header line (every 4th line from the
// first, :21) with more than one field and a second field other than "0_0_0" is looked up in the paternal, then
// maternal, then homozygous list (:23-35), otherwise it is a no-barcode read (:36-39); all four lines follow the
// header's class (:41-49).  Host code; the bytes come from a block source (ingest.h BlockSource: plain files, .gz inflated by the
// host decoders, "-"; or anything else with next() / recycle() / error(), e.g. a .gz inflated on the GPU).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "ingest.h"

namespace hast {
namespace quartering {

inline std::string_view field(std::string_view line, int idx) {      // idx-th field under -F '#|/' (0-based); npos data() if absent
    size_t start = 0;
    for (int f = 0;; ++f) {
        size_t e = start;
        while (e < line.size() && line[e] != '#' && line[e] != '/') ++e;
        if (f == idx) return line.substr(start, e - start);
        if (e >= line.size()) return std::string_view(nullptr, 0);
        start = e + 1;
    }
}

using ClassMap = std::unordered_map<std::string, uint8_t>;           // first field of a list line -> 1 paternal, 2 maternal, 3 homozygous

inline bool load_list(const std::string &path, uint8_t cls, ClassMap &map) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::string data;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) data.append(buf, n);
    fclose(f);
    size_t pos = 0;
    while (pos < data.size()) {
        size_t e = data.find('\n', pos);
        if (e == std::string::npos) e = data.size();
        std::string_view line(data.data() + pos, e - pos);
        map.emplace(std::string(field(line, 0)), cls);             // first list wins, as the awk's if/else chain does
        pos = e + 1;
    }
    return true;
}

// Routes the records of one input.  src: next() -> a block with kFrontPad bytes of room in front of its data (empty = end of input),
// recycle(), error().  log_name: what awk's FILENAME would be (the path as given, "-" behind `gzip -dc`).  who: the program's name in
// messages.  Returns 0, or the exit code of a failure (2: I/O, 3: a record larger than 4 GB).
template <class Source>
int route(const std::string &prefix, const ClassMap &cls_of, Source &src, const std::string &log_name, int t_num, const char *who) {
    hast::WorkerPool pool(t_num);
    const int T = pool.size();
    const char *suffix[4] = {".nobarcode.fastq", ".paternal.fastq", ".maternal.fastq", ".homozygous.fastq"};
    FILE *out[4] = {nullptr, nullptr, nullptr, nullptr};
    long long counts[5] = {0, 0, 0, 0, 0};                         // no, pa, ma, ho, total
    bool any_input = false;
    std::vector<std::vector<uint32_t>> nl(T);
    std::vector<uint32_t> allnl;
    struct Local {
        std::string buf[4], err;
        long long n[4] = {0, 0, 0, 0};
    };
    std::vector<Local> loc(T);

    // class of a header line: 0 nobarcode, 1..3 lists, -1 unclassified (dropped with an ERROR line)
    auto classify = [&](std::string_view head, std::string &err) -> int {
        std::string_view f2 = field(head, 1);
        if (f2.data() == nullptr || f2 == "0_0_0") return 0;        // NF <= 1, or the no-barcode marker (awk :22,36-39)
        auto it = cls_of.find(std::string(f2));
        if (it != cls_of.end()) return it->second;
        err.append("ERROR : unclassify barcode : ").append(f2).append("\n");
        return -1;
    };
    bool write_failed = false;
    auto emit = [&]() {                                             // worker buffers -> files, in input order
        for (int t = 0; t < T; t++) {
            fputs(loc[t].err.c_str(), stderr);
            for (int c = 0; c < 4; c++) {
                counts[c] += loc[t].n[c];
                if (loc[t].buf[c].empty()) continue;
                if (!out[c]) out[c] = fopen((prefix + suffix[c]).c_str(), "wb");
                if (!out[c] || fwrite(loc[t].buf[c].data(), 1, loc[t].buf[c].size(), out[c]) != loc[t].buf[c].size()) {
                    fprintf(stderr, "%s: cannot write %s%s\n", who, prefix.c_str(), suffix[c]);
                    write_failed = true;
                    return;
                }
            }
        }
    };
    std::vector<char> carry;
    for (;;) {
        std::vector<char> blk = src.next();
        const bool last = blk.empty();
        if (last && !src.error().empty()) {
            fprintf(stderr, "%s: %s\n", who, src.error().c_str());
            for (FILE *f : out) if (f) fclose(f);
            return 2;
        }
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
        if (len) any_input = true;
        if (len >= (1ull << 32)) { fprintf(stderr, "%s: record larger than 4 GB\n", who); return 3; }
        pool.run([&](int t) {
            auto &v = nl[t];
            v.clear();
            const char *p = data + len * (size_t)t / T, *e = data + len * (size_t)(t + 1) / T;
            while (p < e && (p = (const char *)memchr(p, '\n', (size_t)(e - p)))) { v.push_back((uint32_t)(p - data)); ++p; }
        });
        allnl.clear();
        for (int t = 0; t < T; t++) allnl.insert(allnl.end(), nl[t].begin(), nl[t].end());
        const size_t n_rec = allnl.size() / 4;
        pool.run([&](int t) {
            Local &L = loc[t];
            for (auto &b : L.buf) b.clear();
            L.err.clear();
            for (auto &x : L.n) x = 0;
            for (size_t i = n_rec * (size_t)t / T; i < n_rec * (size_t)(t + 1) / T; i++) {
                const size_t r0 = i ? (size_t)allnl[4 * i - 1] + 1 : 0, h1 = allnl[4 * i], r1 = (size_t)allnl[4 * i + 3] + 1;
                const int c = classify(std::string_view(data + r0, h1 - r0), L.err);
                if (c >= 0) { L.buf[c].append(data + r0, r1 - r0); L.n[c]++; }
            }
        });
        counts[4] += (long long)n_rec;
        emit();
        if (write_failed) break;
        const size_t consumed = n_rec ? (size_t)allnl[4 * n_rec - 1] + 1 : 0;
        if (last) {
            // trailing partial record: awk still treats every remaining line (terminated or not) as a record (:21,41-49)
            std::string_view rest(data + consumed, len - consumed);
            if (!rest.empty()) {
                Local &L = loc[0];
                for (int t = 0; t < T; t++) { for (auto &b : loc[t].buf) b.clear(); loc[t].err.clear(); for (auto &x : loc[t].n) x = 0; }
                size_t e = rest.find('\n');
                const int c = classify(rest.substr(0, e == std::string_view::npos ? rest.size() : e), L.err);
                counts[4]++;
                if (c >= 0) {
                    L.buf[c].append(rest);
                    if (rest.back() != '\n') L.buf[c].push_back('\n');
                    L.n[c]++;
                }
                emit();
            }
            break;
        }
        std::vector<char> keep(data + consumed, data + len);
        carry.swap(keep);
        src.recycle(std::move(blk));
    }
    for (FILE *f : out)
        if (f && fclose(f) != 0) write_failed = true;
    if (write_failed) return 2;
    FILE *lg = fopen("filter_reads.log", "ab");
    if (lg) {
        if (any_input) fprintf(lg, "%s\n", log_name.c_str());           // awk :18-20 (FNR==1 of the reads file)
        fprintf(lg, "#Total reads                : %lld \n", counts[4]);
        fprintf(lg, "#Reads without barcode      : %lld \n", counts[0]);
        fprintf(lg, "#Paternal reads             : %lld \n", counts[1]);
        fprintf(lg, "#Maternal reads             : %lld \n", counts[2]);
        fprintf(lg, "#Homozygous reads           : %lld \n", counts[3]);
        fclose(lg);
    }
    return 0;
}

}  // namespace quartering
}  // namespace hast

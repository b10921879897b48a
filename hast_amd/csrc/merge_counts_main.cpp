// merge_counts_main.cpp -- shard merge of `classify` outputs: what the reference's unbuilt and
// buggy mergeResult.cpp (01.classify_stlfr_reads/mergeResult.cpp:28-29 adds hap1 counts into key 0) set out to be.
// Several lanes / nodes classify disjoint read sets against the SAME k-mer sets; their TSVs
// (barcode \t hap \t c0 \t c1, classify.cpp:93-102) are summed per barcode and the call is recomputed with
// getHap (classify.cpp:66-86), which needs the two set sizes |S_0|,|S_1| (after adaptor scrub) and the weights:
//
//   merge_counts --set0 N0 --set1 N1 [--weight0 W] [--weight1 W] shard1.tsv shard2.tsv ... > merged.tsv
//
// (`classify --stats` prints set0=/set1= on stderr.)  Output rows are byte-wise sorted like the reference's.
// Inside one node the same merge happens on the GPUs with one RCCL all-reduce; this tool is for results that
// only exist as files.  Host-only.
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>

#include "../../include/hast.h"

int main(int argc, char **argv) {
    static struct option lo[] = {{"set0", required_argument, NULL, 'a'},    {"set1", required_argument, NULL, 'b'},
                                 {"weight0", required_argument, NULL, 'w'}, {"weight1", required_argument, NULL, 'u'},
                                 {0, 0, 0, 0}};
    unsigned long long n0 = 0, n1 = 0;
    double w0 = 1.0, w1 = 1.0;
    for (;;) {
        int c = getopt_long(argc, argv, "a:b:w:u:", lo, NULL);
        if (c < 0) break;
        switch (c) {
        case 'a': n0 = strtoull(optarg, 0, 10); break;
        case 'b': n1 = strtoull(optarg, 0, 10); break;
        case 'w': w0 = atof(optarg); break;
        case 'u': w1 = atof(optarg); break;
        default: return 2;
        }
    }
    if (!n0 || !n1 || optind >= argc) {
        fprintf(stderr, "usage: merge_counts --set0 N0 --set1 N1 [--weight0 W] [--weight1 W] shard.tsv...\n");
        return 2;
    }
    std::map<std::string, std::pair<long long, long long>> sum;      // std::map<std::string>: the reference's row order
    for (int i = optind; i < argc; i++) {
        std::ifstream in(argv[i]);
        if (!in) { fprintf(stderr, "merge_counts: cannot read %s\n", argv[i]); return 2; }
        std::string line;
        while (std::getline(in, line)) {
            // columns from the right: the barcode itself may contain tabs (parseName keeps them, classify.cpp:112-119)
            size_t t3 = line.rfind('\t');
            size_t t2 = t3 == std::string::npos || t3 == 0 ? std::string::npos : line.rfind('\t', t3 - 1);
            size_t t1 = t2 == std::string::npos || t2 == 0 ? std::string::npos : line.rfind('\t', t2 - 1);
            if (t1 == std::string::npos) { fprintf(stderr, "merge_counts: bad row in %s: %s\n", argv[i], line.c_str()); return 3; }
            auto &s = sum[line.substr(0, t1)];
            s.first += atoll(line.c_str() + t2 + 1);
            s.second += atoll(line.c_str() + t3 + 1);
        }
    }
    for (const auto &kv : sum) {
        // 64-bit sums: shards whose totals pass INT_MAX (classify.cpp:51 counts in `int`) print the exact count, as classify does
        int hap = hast_get_hap(kv.first.data(), kv.first.size(), (uint64_t)kv.second.first, (uint64_t)kv.second.second, n0, n1, w0, w1);
        printf("%s\t%d\t%lld\t%lld\n", kv.first.c_str(), hap, kv.second.first, kv.second.second);
    }
    return 0;
}

// gen_fastq.cpp -- measurement tool: writes the SURVEY 8(d) synthetic workload as FILES (k-mer text files +
// paired stLFR-style FASTQ) so the drop-in `classify` CLI and the real reference binary can be run and
// timed on the same inputs.   gen_fastq <out_dir> <n_pairs> <keys_per_hap> <n_barcodes> [K=21] [L=150] [threads=8] [clustered=0]
// Header shape: @V300R%09d#<barcode>/<mate>\t<id>\t1 ; barcode id 0 -> 0_0_0, else a_b_c with a,b,c in [1,1536].
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../hast_amd/csrc/hast_common.h"

using namespace hast;

static std::string barcode_name(uint32_t id) {
    if (id == 0) return "0_0_0";
    char buf[64];
    snprintf(buf, sizeof(buf), "%u_%u_%u", id % 1536 + 1, (id / 1536) % 1536 + 1, id / (1536 * 1536) + 1);
    return buf;
}

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: gen_fastq out_dir n_pairs keys_per_hap n_barcodes [K] [L] [threads]\n");
        return 1;
    }
    std::string dir = argv[1];
    const uint64_t n_pairs = strtoull(argv[2], 0, 10), n_keys = strtoull(argv[3], 0, 10);
    SynthParams p{0x4841535401ull, 0x4841535402ull, 0x4841535403ull, n_keys, (uint32_t)strtoul(argv[4], 0, 10),
                  argc > 6 ? (uint32_t)atoi(argv[6]) : 150u, argc > 5 ? (uint32_t)atoi(argv[5]) : 21u, 0};
    const int threads = argc > 7 ? atoi(argv[7]) : 8;
    // what this run would write: two key files of K+1 bytes per line, two FASTQ files of ~2L+40 bytes per record.  Refused above
    // 64 GB (GEN_FASTQ_MAX_GB): a swapped argument once asked for 100 000 reads of 5 Mbp and took the machine down with it.
    {
        const double bytes = 2.0 * (double)n_keys * (p.k + 1) + 2.0 * (double)n_pairs * (2.0 * p.read_len + 40);
        const double cap = (getenv("GEN_FASTQ_MAX_GB") ? atof(getenv("GEN_FASTQ_MAX_GB")) : 64.0) * 1e9;
        if (bytes > cap || p.k < 1 || p.k > 32 || p.read_len < p.k) {
            fprintf(stderr, "gen_fastq: %llu pairs of %u bp + %llu %u-mers per haplotype = %.1f GB: refused (limit %.0f GB, GEN_FASTQ_MAX_GB)\n",
                    (unsigned long long)n_pairs, p.read_len, (unsigned long long)n_keys, p.k, bytes / 1e9, cap / 1e9);
            return 1;
        }
    }
    if (argc > 8 && atoi(argv[8])) p.reserved = 1;            // clustered keys (runs of K windows around variant sites)
    for (int h = 0; h < 2; h++) {
        FILE *f = fopen((dir + "/hap" + std::to_string(h) + ".mer").c_str(), "wb");
        if (!f) return 2;
        std::vector<char> buf;
        buf.reserve(64 << 20);
        for (uint64_t j = 0; j < n_keys; j++) {
            uint64_t key = synth_key(p, h, j);
            for (uint32_t i = 0; i < p.k; i++) buf.push_back("ACTG"[(key >> (2 * (p.k - 1 - i))) & 3]);
            buf.push_back('\n');
            if (buf.size() > (60u << 20)) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
        }
        fwrite(buf.data(), 1, buf.size(), f);
        fclose(f);
    }
    for (int mate = 0; mate < 2; mate++) {
        FILE *f = fopen((dir + "/r" + std::to_string(mate + 1) + ".fq").c_str(), "wb");
        if (!f) return 2;
        const uint64_t chunk = 1 << 18;
        for (uint64_t c0 = 0; c0 < n_pairs; c0 += chunk * threads) {
            std::vector<std::string> out(threads);
            std::vector<std::thread> th;
            for (int t = 0; t < threads; t++)
                th.emplace_back([&, t] {
                    std::string &o = out[t];
                    std::vector<uint8_t> seq(p.read_len);
                    std::string qual(p.read_len, 'F');
                    // GEN_FASTQ_QUAL=noisy: quality strings like a sequencer's (MGI / Illumina binned scores: mostly 'F', some ':' ',' '#'
                    // at random places) instead of the constant one -- the classification does not look at them, a gzip stream does:
                    // constant lines become 150-byte matches and 300-KB deflate blocks, noisy ones literals and ~40-KB blocks
                    static const bool noisy = getenv("GEN_FASTQ_QUAL") && !strcmp(getenv("GEN_FASTQ_QUAL"), "noisy");
                    char head[128];
                    for (uint64_t i = c0 + chunk * t; i < std::min(n_pairs, c0 + chunk * (t + 1)); i++) {
                        const uint32_t bc = synth_barcode(p, i);
                        synth_read_bc(p, 2 * i + mate, bc, seq.data());
                        int n = snprintf(head, sizeof(head), "@V300R%09llu#%s/%d\t%llu\t1\n", (unsigned long long)i,
                                         barcode_name(bc).c_str(), mate + 1, (unsigned long long)i);
                        o.append(head, n);
                        o.append((const char *)seq.data(), seq.size());
                        o.append("\n+\n");
                        if (noisy) {
                            uint64_t h = splitmix64((2 * i + mate) * 0x9E3779B97F4A7C15ull + 77);
                            for (uint32_t q = 0; q < p.read_len; q++) {
                                if ((q & 7) == 0) h = splitmix64(h + q);
                                const uint32_t r = (uint32_t)(h >> (8 * (q & 7))) & 0xFF;
                                qual[q] = r < 218 ? 'F' : r < 238 ? ':' : r < 251 ? ',' : '#';
                            }
                        }
                        o.append(qual);
                        o.push_back('\n');
                    }
                });
            for (auto &t : th) t.join();
            for (auto &o : out) fwrite(o.data(), 1, o.size(), f);
        }
        fclose(f);
    }
    return 0;
}

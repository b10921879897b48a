// gen_trio.cpp -- measurement tool: writes the stage-00 synthetic trio (hast_amd/csrc/kc_common.h: one random genome,
// parent-specific SNPs, reads with substitution errors) as FASTQ FILES, so that the drop-in `unshared_kmers`, the
// oracle's front end and (in the build container) the real reference script can be run and timed on the same inputs.
//   gen_trio <out_dir> <genome_len> <coverage> [L=150] [files_per_parent=2] [threads=8]
// Files: maternal_<i>.fq, paternal_<i>.fq (read j of a parent goes to file j % files).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "../hast_amd/csrc/kc_common.h"

using namespace hast;

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: gen_trio out_dir genome_len coverage [L] [files_per_parent] [threads]\n");
        return 1;
    }
    const std::string dir = argv[1];
    KcSynth g{0x4841535400ull, strtoull(argv[2], 0, 10), argc > 4 ? (uint32_t)atoi(argv[4]) : 150u, 1, 20, 20};
    const double cov = atof(argv[3]);
    const int n_files = argc > 5 ? atoi(argv[5]) : 2, threads = argc > 6 ? atoi(argv[6]) : 8;
    const uint64_t n_reads = (uint64_t)((double)g.genome_len * cov / g.read_len);
    const char *names[2] = {"paternal", "maternal"};
    for (int parent = 0; parent < 2; ++parent)
        for (int fi = 0; fi < n_files; ++fi) {
            FILE *f = fopen((dir + "/" + names[parent] + "_" + std::to_string(fi) + ".fq").c_str(), "wb");
            if (!f) return 2;
            const uint64_t chunk = 1 << 16;
            std::vector<std::string> parts(threads);
            for (uint64_t r0 = fi; r0 < n_reads; r0 += chunk * threads * n_files) {
                std::vector<std::thread> th;
                for (int t = 0; t < threads; ++t)
                    th.emplace_back([&, t] {
                        std::string &s = parts[t];
                        s.clear();
                        char head[64];
                        for (uint64_t j = 0; j < chunk; ++j) {
                            const uint64_t r = r0 + ((uint64_t)t * chunk + j) * n_files;
                            if (r >= n_reads) break;
                            s.append(head, (size_t)snprintf(head, sizeof head, "@%c%010llu/%d\n", names[parent][0], (unsigned long long)r, 1 + (int)(r & 1)));
                            const size_t at = s.size();
                            s.resize(at + g.read_len);
                            for (uint32_t q = 0; q < g.read_len; ++q) s[at + q] = (char)kc_synth_byte(g, parent, r, q);
                            s.append("\n+\n");
                            s.append(g.read_len, 'F');
                            s.push_back('\n');
                        }
                    });
                for (auto &t : th) t.join();
                for (auto &s : parts) fwrite(s.data(), 1, s.size(), f);
            }
            fclose(f);
        }
    fprintf(stderr, "gen_trio: %llu reads per parent, %d files each\n", (unsigned long long)n_reads, n_files);
    return 0;
}

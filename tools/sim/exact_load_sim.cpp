// exact_load_sim.cpp -- design tool, not product code: load of the EXACT-entry filter (hast_common.h) for random and for clustered
// keys (synth_key, reserved = 1: runs of K windows around variant sites), one slice of the blocks: sub-bucket histogram and the
// share of random read windows that land in a FULL sub-bucket (those ask the exact table).   exact_load_sim keys_per_hap clustered(0/1)
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "../../hast_amd/csrc/hast_common.h"
using namespace hast;
int main(int argc, char **argv) {
    const uint64_t n_per_hap = strtoull(argv[1], 0, 10);
    const int clustered = atoi(argv[2]), threads = 8;
    const uint32_t SLICE = 64;
    const int K = 21;
    FilterGeom g = filter_geom_for(K, 2 * n_per_hap, 0, 0);
    SynthParams p{0x4841535401ull, 0x4841535402ull, 0x4841535403ull, n_per_hap, 1000, 150, (uint32_t)K, (uint32_t)(clustered ? 1 : 0)};
    const uint64_t nb = filter_nblocks(g), nslice = nb / SLICE;
    std::vector<std::atomic<uint8_t>> cnt(nslice * 8);
    for (auto &c : cnt) c = 0;
    std::atomic<uint64_t> filed{0}, lost{0};
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back([&, t] {
        for (int h = 0; h < 2; h++)
            for (uint64_t j = t; j < n_per_hap; j += threads) {
                const uint64_t key = synth_key(p, h, j);
                for (int o = 0; o < 2; o++) {
                    const uint64_t s = o ? kmer_revcomp(key, K) : key;
                    if (o && s == key) break;
                    const uint32_t pm = filter_sample_pos(s, g);
                    const uint32_t b = filter_block_of((uint32_t)(s >> (2 * (K - g.m - (int)pm))) & (uint32_t)kmer_mask(g.m), g.m);
                    if (b % SLICE) continue;
                    const uint32_t c17 = filter_exact_code(s, pm, g);
                    std::atomic<uint8_t> &c = cnt[(uint64_t)(b / SLICE) * 8 + filter_exact_sub(c17)];
                    // (duplicates -- the same string from both haplotypes -- would be one entry; ignored here)
                    uint8_t v = c.load();
                    if (v >= 8) { lost++; continue; }
                    c++;
                    filed++;
                }
            }
    });
    for (auto &x : th) x.join();
    uint64_t hist[10] = {0};
    for (auto &c : cnt) hist[c > 8 ? 8 : (int)c]++;
    // random windows
    std::atomic<uint64_t> in_slice{0}, full{0};
    th.clear();
    for (int t = 0; t < threads; t++) th.emplace_back([&, t] {
        for (uint64_t i = t; i < 40000000ull; i += threads) {
            const uint64_t s = synth_rand(99, i, 7) & kmer_mask(K);
            const uint32_t pm = filter_sample_pos(s, g);
            const uint32_t b = filter_block_of((uint32_t)(s >> (2 * (K - g.m - (int)pm))) & (uint32_t)kmer_mask(g.m), g.m);
            if (b % SLICE) continue;
            in_slice++;
            if (cnt[(uint64_t)(b / SLICE) * 8 + filter_exact_sub(filter_exact_code(s, pm, g))] >= 8) full++;
        }
    });
    for (auto &x : th) x.join();
    printf("clustered=%d m=%d filed/block %.2f lost %.5f hist", clustered, g.m, (double)filed / nslice, (double)lost / (double)(lost + filed));
    for (int i = 0; i <= 8; i++) printf(" %llu", (unsigned long long)hist[i]);
    printf("  windows in a full sub-bucket: %.5f = %.2f per 130-window read\n", (double)full / in_slice, 130.0 * full / in_slice);
}

// filter_load_sim.cpp -- design tool, not product code: faithful (unscaled) simulation of the fingerprint filter's
// per-block load for a slice of the address space.  N random canonical K-mers are filed under the block of both
// orientations (forward schemes); only blocks with id % SLICE == 0 are tallied.  Then random reads are pushed through:
// blocks per read (distinct per 64-window instruction), and how many windows land in a FULL sub-bucket (forced verify).
//   filter_load_sim N K m t choices [threads] [slice]
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../../hast_amd/csrc/hast_common.h"
using namespace hast;
int main(int argc, char **argv) {
    const uint64_t N = strtoull(argv[1], 0, 10);
    const int K = atoi(argv[2]), m = atoi(argv[3]), t = atoi(argv[4]), choices = atoi(argv[5]);
    const int threads = argc > 6 ? atoi(argv[6]) : 8;
    const uint32_t SLICE = argc > 7 ? atoi(argv[7]) : 256;
    // the sampling scheme only looks at the first kp bases of a window (kp = argv[8], default K)
    const int kp = argc > 8 ? atoi(argv[8]) : K;
    FilterGeom g = filter_geom_for(kp, 0, m, t);
    auto block_of = [&](uint64_t s) { return filter_block_of_string(s >> (2 * (K - kp)), g); };
    const uint64_t nb = filter_nblocks(g);
    const uint64_t nslice = nb / SLICE;
    std::vector<std::atomic<uint8_t>> cnt(nslice * 8);
    for (auto &c : cnt) c = 0;
    std::atomic<uint64_t> lost{0}, filed{0};
    auto sub2 = [](uint32_t h) { return (h >> 26) & 7u; };
    std::vector<std::thread> th;
    for (int ti = 0; ti < threads; ti++) th.emplace_back([&, ti] {
        for (uint64_t i = ti; i < N; i += threads) {
            const uint64_t w = synth_rand(77, 1, i) & kmer_mask(K);
            const uint64_t key = kmer_canon(w, K);
            for (int o = 0; o < 2; o++) {
                const uint64_t s = o ? kmer_revcomp(key, K) : key;
                if (o && s == key) break;
                const uint32_t b = block_of(s);
                const uint32_t h = filter_keyhash(s);
                if (b % SLICE) continue;
                const uint64_t base = (uint64_t)(b / SLICE) * 8;
                uint32_t s1 = filter_sub_of(h), s2 = sub2(h);
                uint32_t pick = s1;
                if (choices == 2 && s2 != s1 && cnt[base + s2] < cnt[base + s1]) pick = s2;
                if (cnt[base + pick] >= 8) { lost++; continue; }
                cnt[base + pick]++;
                filed++;
            }
        }
    });
    for (auto &x : th) x.join();
    uint64_t hist[10] = {0};
    for (auto &c : cnt) hist[std::min<int>(c, 8)]++;
    // reads
    const int L = argc > 9 ? atoi(argv[9]) : 150, nreads = 400000 * 150 / L;
    std::atomic<uint64_t> tot_w{0}, in_slice{0}, forced{0}, blocks64{0};
    th.clear();
    for (int ti = 0; ti < threads; ti++) th.emplace_back([&, ti] {
        std::vector<uint32_t> blk(L);
        for (int r = ti; r < nreads; r += threads) {
            std::vector<uint8_t> code(L);
            for (int j = 0; j < L; j += 32) { uint64_t x = synth_rand(99, r, j); for (int q = 0; q < 32 && j + q < L; q++) code[j + q] = (x >> (2 * q)) & 3; }
            int nw = L - K + 1;
            for (int p = 0; p < nw; p++) {
                uint64_t fwd = 0;
                for (int i = 0; i < K; i++) fwd = (fwd << 2) | code[p + i];
                const uint32_t b = block_of(fwd);
                blk[p] = b;
                tot_w++;
                if (b % SLICE == 0) {
                    in_slice++;
                    const uint32_t h = filter_keyhash(fwd);
                    const uint64_t base = (uint64_t)(b / SLICE) * 8;
                    bool f = cnt[base + filter_sub_of(h)] >= 8;
                    if (choices == 2) f = f && cnt[base + sub2(h)] >= 8;      // a key may sit in either: only "both full" hides one
                    if (f) forced++;
                }
            }
            for (int i0 = 0; i0 < nw; i0 += 64) { std::vector<uint32_t> v(blk.begin() + i0, blk.begin() + std::min(nw, i0 + 64)); std::sort(v.begin(), v.end()); blocks64 += std::unique(v.begin(), v.end()) - v.begin(); }
        }
    });
    for (auto &x : th) x.join();
    printf("{\"N\": %llu, \"K\": %d, \"m\": %d, \"t\": %d, \"W\": %u, \"choices\": %d, \"blocks_M\": %.1f, \"filed_per_block\": %.2f, \"lost_frac\": %.5f, "
           "\"sub_hist_0..8+\": [%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu], \"blocks64_per_read\": %.2f, \"forced_verify_per_read\": %.3f}\n",
           (unsigned long long)N, K, g.m, g.t, filter_w(g), choices, nb / 1e6, (double)filed / nslice, (double)lost / (double)(lost + filed),
           (unsigned long long)hist[0], (unsigned long long)hist[1], (unsigned long long)hist[2], (unsigned long long)hist[3], (unsigned long long)hist[4], (unsigned long long)hist[5], (unsigned long long)hist[6], (unsigned long long)hist[7], (unsigned long long)hist[8],
           (double)blocks64 / (double)tot_w, (double)forced / (double)in_slice);
    return 0;
}

// Host-only test of the fingerprint filter's placement arithmetic (hast_amd/csrc/hast_common.h), the part shared by the
// build kernel (one key at a time) and the classify kernel (sliding over a read):
//   1. x mod W by multiply-shift is exact for every W <= 32 and x < 64;
//   2. a host model of the kernel's sliding selection (position-tagged t-mer orders, first-level minima over g entries,
//      chunks at 0, g, 2g, ... and K-t+1-g) names the same m-mer as filter_sample_pos on the window's own string, for every
//      window of random reads, over many (K, m, t);
//   3. the block a read window asks for is one of the (at most two) blocks its canonical key was filed under;
//   4. the geometry picked for the BASELINE sizes.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../hast_amd/csrc/hast_common.h"

using namespace hast;

static uint64_t rng_state = 0x1234567ull;
static uint64_t rnd() { return rng_state = splitmix64(rng_state); }

int main() {
    for (uint32_t w = 1; w <= 32; ++w) {
        const uint32_t magic = 65536u / w + 1u;
        for (uint32_t x = 0; x < 64; ++x)
            if (x - ((x * magic) >> 16) * w != x % w) { printf("mod: W=%u x=%u\n", w, x); return 1; }
    }
    long windows = 0;
    for (int it = 0; it < 400; ++it) {
        const int k = 1 + (int)(rnd() % 32);
        int m = 1 + (int)(rnd() % (uint64_t)(k < kFilterMaxM ? k : kFilterMaxM));
        int t = (rnd() & 1) ? 0 : 1 + (int)(rnd() % (uint64_t)m);
        const int kpo = (rnd() & 1) ? 0 : m + (int)(rnd() % (uint64_t)(k - m + 1));
        const FilterGeom g = filter_geom_for(k, 0, m, t, kpo);
        if (g.m != m || (t && g.t != t) || g.t > g.m || g.t < 1) { printf("geom: k=%d m=%d t=%d -> m=%d t=%d\n", k, m, t, g.m, g.t); return 1; }
        const uint32_t W = filter_w(g), NT = filter_nt(g), G = (uint32_t)g.g;
        const uint32_t L = (uint32_t)k + (uint32_t)(rnd() % 200);
        std::vector<uint8_t> code(L);
        for (auto &c : code) c = (uint8_t)(rnd() & 3);
        // kernel model: e[q] for every t-mer position of the read, L1[q] = min(e[q..q+G-1])
        const uint32_t n_t = L - (uint32_t)g.t + 1;
        std::vector<uint32_t> e(n_t + 8, 0xFFFFFFFFu), l1(n_t + 8, 0xFFFFFFFFu);
        for (uint32_t q = 0; q < n_t; ++q) {
            uint32_t tm = 0;
            for (int i = 0; i < g.t; ++i) tm = (tm << 2) | code[q + i];
            e[q] = tmer_order(tm, q);
        }
        for (uint32_t q = 0; q < n_t; ++q) {
            uint32_t mn = e[q];
            for (uint32_t i = 1; i < G; ++i) mn = mn < e[q + i] ? mn : e[q + i];
            l1[q] = mn;
        }
        const uint32_t ntc = (NT + G - 1) / G;
        for (uint32_t p = 0; p + k <= L; ++p, ++windows) {
            uint64_t fwd = 0;
            for (int i = 0; i < k; ++i) fwd = (fwd << 2) | code[p + i];
            uint32_t x = l1[p];
            for (uint32_t c = 1; c + 1 < ntc; ++c) x = x < l1[p + c * G] ? x : l1[p + c * G];
            if (ntc > 1) x = x < l1[p + NT - G] ? x : l1[p + NT - G];
            const uint32_t xr = ((x & 0xFFFu) - p) & 63u;
            const uint32_t pm = xr - ((xr * g.wdiv) >> 16) * W;
            if (pm != filter_sample_pos(fwd, g) || pm >= W) { printf("slide: k=%d m=%d t=%d p=%u: %u vs %u\n", k, g.m, g.t, p, pm, filter_sample_pos(fwd, g)); return 1; }
            // strand: the window's block is one of the blocks its canonical key is filed under
            const uint64_t rc = kmer_revcomp(fwd, k), key = fwd < rc ? fwd : rc;
            const uint32_t mine = filter_block_of_string(fwd, g);
            if (mine != filter_block_of_string(key, g) && mine != filter_block_of_string(kmer_revcomp(key, k), g)) { printf("strand\n"); return 1; }
            if (mine >= filter_nblocks(g)) { printf("range\n"); return 1; }
            const uint32_t h = filter_keyhash(key);
            if (filter_sub_of(h) >= (uint32_t)kFilterSubs || filter_print_of(h) == 0 || filter_print_of(h) > 0xFFFF) { printf("print\n"); return 1; }
        }
    }
    struct { int k; uint64_t n; int m, t, kp; } want[] = {{21, 400000000ull, 14, 6, 21}, {21, 100000000ull, 13, 4, 21}, {31, 800000000ull, 14, 5, 22},
                                                          {21, 40000ull, 8, 8, 16}, {5, 10ull, 5, 4, 5}, {32, 2000000000ull, 14, 5, 22}};
    for (auto &w : want) {
        const FilterGeom g = filter_geom_for(w.k, w.n, 0, 0);
        if (g.m != w.m || g.t != w.t || g.kp != w.kp) { printf("pick: K=%d n=%llu -> m=%d t=%d kp=%d (want %d %d %d)\n", w.k, (unsigned long long)w.n, g.m, g.t, g.kp, w.m, w.t, w.kp); return 1; }
    }
    printf("ok %ld windows\n", windows);
    return 0;
}

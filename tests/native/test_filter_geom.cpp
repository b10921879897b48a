// Host-only test of the fingerprint filter's placement arithmetic (hast_amd/csrc/hast_common.h), the part shared by the
// build kernel (one key at a time) and the classify kernel (sliding over a read):
//   1. x mod W by multiply-shift is exact for every W <= 32 and x < 64;
//   2. a host model of the kernel's sliding selection (position-tagged t-mer orders, first-level minima over g entries,
//      chunks at 0, g, 2g, ... and K-t+1-g) names the same m-mer as filter_sample_pos on the window's own string, for every
//      window of random reads, over many (K, m, t);
//   3. the block a read window asks for is one of the (at most two) blocks its canonical key was filed under;
//   4. exact entries (FilterGeom::exact): block + sub-bucket + the 14 stored bits give the filed string back, bit for bit;
//   5. the geometry picked for the BASELINE sizes.
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
    long windows = 0, exact_windows = 0;
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
            if (g.exact) {
                // the string from (block, code): the scramble of the m-mer is an involution, the code's multiplier has an inverse mod 2^17
                static uint32_t inv = 0;
                if (!inv) { inv = 1; for (int i = 0; i < 5; ++i) inv *= 2u - 0x1D2C5u * inv; inv &= 0x1FFFFu; }
                const uint32_t c17 = filter_exact_code(fwd, pm, g);
                const uint32_t entry = filter_exact_entry(c17, 1 + (uint32_t)(rnd() % 3));
                if (c17 > 0x1FFFFu || filter_exact_sub(c17) >= (uint32_t)kFilterSubs || entry == 0 || entry > 0xFFFFu) { printf("exact: range\n"); return 1; }
                const uint32_t back17 = (filter_exact_sub(c17) << 14) | (entry >> 2);
                const uint32_t code = (back17 * inv) & 0x1FFFFu;
                const int rb = 2 * (k - g.m);
                const uint32_t pm2 = code >> rb, rest = code & ((1u << rb) - 1u);
                const uint64_t mm = (uint64_t)(mine ^ (mine >> g.m)) & kmer_mask(g.m);
                const uint64_t prefix = rest & ((1ull << (2 * pm2)) - 1), suffix = rest >> (2 * pm2);
                const uint64_t again = (pm2 ? prefix << (2 * (k - (int)pm2)) : 0) | (mm << (2 * (k - g.m - (int)pm2))) | suffix;
                if (pm2 != pm || again != fwd) { printf("exact: k=%d m=%d pm=%u: %llx -> %llx\n", k, g.m, pm, (unsigned long long)fwd, (unsigned long long)again); return 1; }
                ++exact_windows;
            }
        }
    }
    // (K = 21 from 16M keys on: m = 14 so that the entries are exact; with prints forced, the smallest m that holds the keys)
    struct { int k; uint64_t n; int mode, m, t, kp, exact; } want[] = {{21, 400000000ull, -1, 14, 6, 21, 1}, {21, 100000000ull, -1, 14, 6, 21, 1}, {21, 100000000ull, 0, 13, 4, 21, 0},
                                                                       {21, 400000000ull, 0, 14, 6, 21, 0}, {31, 800000000ull, -1, 15, 6, 23, 0}, {21, 40000ull, -1, 8, 8, 16, 0},
                                                                       {5, 10ull, -1, 5, 4, 5, 1}, {32, 2000000000ull, -1, 15, 6, 23, 0}, {17, 400000000ull, -1, 14, 6, 17, 1},
                                                                       // long windows get the m that files prints once (W = 9 either way)
                                                                       {31, 400000000ull, -1, 15, 6, 23, 0}, {31, 100000000ull, -1, 14, 5, 22, 0}, {23, 400000000ull, -1, 15, 6, 23, 0}};
    for (auto &w : want) {
        const FilterGeom g = filter_geom_for(w.k, w.n, 0, 0, 0, w.mode);
        // one choice per print exactly where a block holds <= 2.2 strings on average
        if (!g.exact && g.choices != ((2.0 * (double)w.n <= 2.2 * (double)(1ull << (2 * g.m))) ? 1 : 2)) { printf("pick: choices\n"); return 1; }
        if (w.k == 31 && w.n == 800000000ull && g.choices != 1) { printf("pick: config 5 should file prints once\n"); return 1; }
        if (g.exact != w.exact) { printf("pick: K=%d n=%llu mode %d -> exact=%d\n", w.k, (unsigned long long)w.n, w.mode, g.exact); return 1; }
        if (g.m != w.m || g.t != w.t || g.kp != w.kp) { printf("pick: K=%d n=%llu -> m=%d t=%d kp=%d (want %d %d %d)\n", w.k, (unsigned long long)w.n, g.m, g.t, g.kp, w.m, w.t, w.kp); return 1; }
    }
    if (exact_windows < windows / 20) { printf("too few exact geometries tried: %ld of %ld\n", exact_windows, windows); return 1; }
    printf("ok %ld windows, %ld with exact entries\n", windows, exact_windows);
    return 0;
}

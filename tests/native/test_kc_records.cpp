// Host model of stage 00's partitioned counting (hast_amd/csrc/kc_common.h "records", "PLACEMENT"; kernels in kc_kernels.hip): windows
// of a byte stream -> records of minimizer runs (as k_kc_count<EMIT> cuts them) -> every record placed by ONE m-mer hash
// (kc_rec_minhash) -> its windows expanded again (as k_kc_apply does).  Checked against the per-window arithmetic of the direct path:
//   * the windows that come back out of the records are exactly the valid windows of the stream, each once, with its canonical key;
//   * a record's placement hash is the minimizer hash of EVERY window it holds (so the slice it is sent to is the one the direct
//     kernel's bucket_of_minhash would pick for each of them, and a k-mer has one slice wherever and on whichever strand it occurs);
//   * the record fits 64 bits for every K the path accepts (kc_run_max), the offset field never overlaps the bases;
//   * kc_key_bucket stays inside the slice.
//   * k_kc_emit4's lanes (K = 17 .. 21: four positions, then four windows per lane, steps of 256 windows) restated with plain loops over
//     the shared per-lane arithmetic (kc_common.h kc_e4_*): the hashes of phase M equal kc_mmer_hash of the canonical 16-mers, and the
//     descriptors of phase B are exactly the runs of the definition above, cut at the steps' borders.
//   * k_kc_part's tile (8192 records: counts per bin, a scan, the records grouped by bin with the bin's low byte beside them, then a lane
//     per record in bin order that finds its bin from that byte and the first slots of bins 256 / 512 / 768) restated with plain loops:
//     every record arrives in its bin's region, the regions hold exactly the tile's records, in the order the kernel's ranks give.
// TEST INFRASTRUCTURE: no product code is called besides the shared integer header.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "../../hast_amd/csrc/kc_common.h"

using namespace hast;

static bool is_base(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'a' || c == 'c' || c == 'g' || c == 't'; }


// k_kc_emit4's phases M and B over one "tile" = the whole stream (a multiple of 256 windows is not required: the last step is partial)
template <int WT>
static int emit4_model(const std::string &s, int k, const std::vector<char> &valid, const std::vector<uint32_t> &off, std::mt19937_64 &rng, size_t *n_desc) {
    const size_t n = s.size();
    // the packed tile: 2 bits per byte whatever it is (the validity mask is separate), padded with junk
    std::vector<uint8_t> code(n + 96);
    for (size_t i = 0; i < code.size(); ++i) code[i] = i < n ? (uint8_t)base_code((uint8_t)s[i]) : (uint8_t)(rng() & 3);
    auto bits64 = [&](size_t q) { uint64_t x = 0; for (int i = 0; i < 32; ++i) x = (x << 2) | code[q + (size_t)i]; return x; };
    // phase M: a "lane" per four positions
    std::vector<uint32_t> H(n + 64);
    for (size_t q0 = 0; q0 < n + 60; q0 += 4) {
        uint32_t h[4];
        kc_e4_mmer_hashes(bits64(q0), h);
        for (int i = 0; i < 4; ++i) H[q0 + (size_t)i] = h[i];
    }
    for (size_t q = 0; q + 16 <= n; ++q) {                       // == the definition, wherever the 16 bytes are bases
        bool ok = true;
        uint64_t f = 0;
        for (int i = 0; i < 16; ++i) { ok = ok && is_base((uint8_t)s[q + (size_t)i]); f = (f << 2) | base_code((uint8_t)s[q + (size_t)i]); }
        if (ok && H[q] != kc_mmer_hash(kmer_canon(f, 16))) { printf("K=%d: phase M's hash of position %zu differs from kc_mmer_hash\n", k, q); return 1; }
        if (H[q] & 15u) { printf("K=%d: a hash with low bits\n", k); return 1; }
    }
    // phase B: steps of 256 windows, 64 lanes of four
    std::vector<uint32_t> desc;
    for (size_t base = 0; base < n; base += 256) {
        uint32_t vm[64], cn[64], at[64][4];
        for (uint32_t lane = 0; lane < 64; ++lane) {
            const size_t p0 = base + 4 * lane;
            uint32_t c[9], w[4];
            for (int j = 0; j < 9; ++j) c[j] = (p0 + (size_t)j < H.size() ? H[p0 + (size_t)j] : (uint32_t)rng() & ~15u) | (uint32_t)j;
            kc_e4_minima<WT>(c, w);
            vm[lane] = 0;
            for (int q = 0; q < 4; ++q) {
                at[lane][q] = w[q] & 15u;
                if (p0 + (size_t)q < n && valid[p0 + (size_t)q]) vm[lane] |= 1u << q;
            }
        }
        for (uint32_t lane = 0; lane < 64; ++lane)
            cn[lane] = kc_e4_cont(vm[lane], at[lane], lane > 0, lane ? at[lane - 1][3] | ((vm[lane - 1] >> 3) << 4) : 0u);
        for (uint32_t lane = 0; lane < 64; ++lane) {
            const uint32_t C = cn[lane] | (lane < 63 ? cn[lane + 1] << 4 : 0u) | (lane < 62 ? cn[lane + 2] << 8 : 0u);
            const uint32_t sn = vm[lane] & ~cn[lane];
            for (int q = 0; q < 4; ++q)
                if ((sn >> q) & 1u) desc.push_back(kc_e4_desc((uint32_t)(base + 4 * lane) + (uint32_t)q, kc_e4_run_minus_1(C, q), at[lane][q] - (uint32_t)q));
        }
    }
    // the definition: runs of consecutive valid windows that name the same minimizer occurrence, cut where a step of 256 windows ends
    std::vector<uint32_t> want;
    for (size_t p = 0; p < n;) {
        if (!valid[p]) { ++p; continue; }
        size_t q = p + 1;
        while (q < n && valid[q] && (q & 255) != 0 && q + off[q] == p + off[p]) ++q;
        want.push_back(kc_e4_desc((uint32_t)p, (uint32_t)(q - p - 1), off[p]));
        p = q;
    }
    if (desc != want) {
        printf("K=%d: k_kc_emit4's lanes cut %zu runs, the definition %zu", k, desc.size(), want.size());
        for (size_t i = 0; i < desc.size() && i < want.size(); ++i)
            if (desc[i] != want[i]) { printf("; first difference at run %zu: window %u run %u offset %u against window %u run %u offset %u", i, desc[i] & 0x3FFF, ((desc[i] >> 14) & 15) + 1, desc[i] >> 18, want[i] & 0x3FFF, ((want[i] >> 14) & 15) + 1, want[i] >> 18); break; }
        printf("\n");
        return 1;
    }
    *n_desc += desc.size();
    return 0;
}

// k_kc_part's steps on one tile: bins[i] = the bin of record i (or -1: a null record).  Returns false and prints on a mismatch.
static bool part_tile_model(const std::vector<int> &bins, uint32_t n_bins, std::mt19937_64 &rng) {
    const uint32_t n = (uint32_t)bins.size();
    std::vector<uint32_t> cnt(1024, 0), rank(n, 0), off(1024, 0);
    // (1) counts; the LDS add hands every record its rank in its bin -- in whatever order the lanes get there: a random one here
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    for (uint32_t i = n; i > 1; --i) std::swap(order[i - 1], order[rng() % i]);
    for (uint32_t i : order)
        if (bins[i] >= 0) rank[i] = cnt[(size_t)bins[i]]++;
    // (2) exclusive scan, one bin per "thread"; the first slots of bins 256, 512, 768
    uint32_t run = 0, first[3] = {0, 0, 0};
    for (uint32_t b = 0; b < 1024; ++b) {
        if (b && (b & 255) == 0) first[(b >> 8) - 1] = run;
        off[b] = run;
        run += cnt[b];
    }
    const uint32_t n_valid = run;
    // (3) scatter: record and the low byte of its bin
    std::vector<uint64_t> rec(n_valid, ~0ull);
    std::vector<uint8_t> bin8(n_valid, 0);
    for (uint32_t i = 0; i < n; ++i)
        if (bins[i] >= 0) {
            const uint32_t at = off[(size_t)bins[i]] + rank[i];
            if (at >= n_valid || rec[at] != ~0ull) { printf("k_kc_part model: two records in one slot\n"); return false; }
            rec[at] = i;
            bin8[at] = (uint8_t)bins[i];
        }
    // (4) a lane per slot: the bin from the byte and the three thresholds; the place in the region = slot - off[bin] (the kernel adds
    //     the region's reservation, which is the same for all records of the bin)
    std::vector<uint32_t> seen(1024, 0);
    for (uint32_t i = 0; i < n_valid; ++i) {
        const uint32_t b = (uint32_t)bin8[i] + ((i >= first[0] ? 256u : 0u) + (i >= first[1] ? 256u : 0u) + (i >= first[2] ? 256u : 0u));
        const uint64_t r = rec[i];
        if (b >= n_bins || bins[(size_t)r] != (int)b) { printf("k_kc_part model: slot %u holds a record of bin %d, the lane says bin %u\n", i, bins[(size_t)r], b); return false; }
        if (i - off[b] != seen[b]++) { printf("k_kc_part model: a bin's records do not leave in slot order\n"); return false; }
    }
    for (uint32_t b = 0; b < 1024; ++b)
        if (seen[b] != cnt[b]) { printf("k_kc_part model: bin %u lost records\n", b); return false; }
    return true;
}

int main() {
    std::mt19937_64 rng(20260404);
    size_t n_records = 0, n_windows = 0, n_cases = 0, n_e4 = 0;
    for (int k = 1; k <= 32; ++k) {
        for (int m : {k <= 16 ? k : (k - 8 > 16 ? k - 8 : 16), k > 4 ? k - 3 : k, k}) {
            if (m < 1 || m > k) continue;
            const uint32_t rmax = kc_run_max(k, m), ob = kc_rec_off_bits(k, m), W = (uint32_t)(k - m + 1);
            if (rmax == 0) {                                   // no room in 64 bits: the path must say so, not pack anyway
                if (2 * k + 6 + (int)ob <= 64) { printf("k=%d m=%d: refused although one window fits\n", k, m); return 1; }
                continue;
            }
            if (2 * (k + (int)rmax - 1) + 6 + (int)ob > 64 || rmax > W) { printf("k=%d m=%d: run of %u does not fit\n", k, m, rmax); return 1; }
            ++n_cases;
            // a stream with every byte class: runs of bases (low-complexity stretches included), separators, N, lower case
            std::string s;
            for (int piece = 0; piece < 60; ++piece) {
                const int len = (int)(rng() % 200);
                const int mode = (int)(rng() % 5);
                for (int i = 0; i < len; ++i) {
                    uint8_t c = "ACGT"[rng() & 3];
                    if (mode == 1) c = "AAAC"[rng() & 3];                    // homopolymers: repeated m-mers, ties of the minimum
                    if (mode == 2) c = "AC"[i & 1];                          // period 2
                    if (mode == 3 && (rng() & 7) == 0) c = (uint8_t)(c | 0x20);
                    s.push_back((char)c);
                }
                s.push_back("\n\nN@+"[rng() % 5]);
            }
            const size_t n = s.size();
            // per-window truth: valid, forward bits, canonical key, minimizer hash and the place of its leftmost smallest m-mer
            std::vector<char> valid(n, 0);
            std::vector<uint64_t> fwd(n, 0);
            std::vector<uint32_t> mn(n, 0), off(n, 0);
            std::multimap<uint64_t, size_t> truth;                           // canonical key -> window start
            for (size_t p = 0; p + (size_t)k <= n; ++p) {
                bool ok = true;
                uint64_t f = 0;
                for (int i = 0; i < k; ++i) {
                    const uint8_t c = (uint8_t)s[p + (size_t)i];
                    ok = ok && is_base(c);
                    f = (f << 2) | base_code(c);
                }
                if (!ok) continue;
                valid[p] = 1;
                fwd[p] = f & kmer_mask(k);
                uint32_t best = 0xFFFFFFFFu, bo = 0;
                for (uint32_t j = 0; j < W; ++j) {
                    const uint32_t h = kc_mmer_hash(kmer_canon((fwd[p] >> (2 * ((uint32_t)(k - m) - j))) & kmer_mask(m), m));
                    if (h < best) { best = h; bo = j; }
                }
                mn[p] = best;
                off[p] = bo;
                if (best != kc_minimizer_hash(fwd[p], k, m)) { printf("k=%d m=%d: window minimum differs from kc_minimizer_hash\n", k, m); return 1; }
                truth.insert({kmer_canon(fwd[p], k), p});
                ++n_windows;
            }
            // records as the emit kernel cuts them: consecutive valid windows whose minimizer is the same m-mer OCCURRENCE, at most rmax
            std::multimap<uint64_t, size_t> back;
            for (size_t p = 0; p < n;) {
                if (!valid[p]) { ++p; continue; }
                size_t q = p + 1;
                while (q < n && valid[q] && q - p < rmax && q + off[q] == p + off[p]) ++q;
                const uint32_t run = (uint32_t)(q - p);
                uint64_t bases = 0;
                for (size_t i = p; i < p + (size_t)k + run - 1; ++i) bases = (bases << 2) | base_code((uint8_t)s[i]);
                const uint64_t rec = (ob ? (uint64_t)off[p] << (64 - ob) : 0ull) | (bases << 6) | ((uint64_t)(run - 1) << 1) | (uint64_t)(p & 1);
                if (rec == ~0ull) { printf("k=%d m=%d: a record equals the null record\n", k, m); return 1; }
                if (ob && (bases >> (58 - ob)) != 0) { printf("k=%d m=%d: bases reach into the offset field\n", k, m); return 1; }
                ++n_records;
                const uint32_t h = kc_rec_minhash(rec, k, m, ob);
                // expand (k_kc_apply): window j of the record
                const uint32_t r2 = (uint32_t)((rec >> 1) & 31) + 1;
                if (r2 != run || (rec & 1) != (p & 1)) { printf("k=%d m=%d: run / parent do not come back\n", k, m); return 1; }
                for (uint32_t j = 0; j < run; ++j) {
                    const uint64_t raw = ((rec >> 6) >> (2 * (run - 1 - j))) & kmer_mask(k);
                    if (raw != fwd[p + j]) { printf("k=%d m=%d: window %u of a record differs from the stream\n", k, m, j); return 1; }
                    if (h != mn[p + j]) { printf("k=%d m=%d: the record's placement hash is not window %u's minimizer hash\n", k, m, j); return 1; }
                    // the record of ONE window a full bucket chain spills (k_kc_apply): the minimizer lies j bases nearer to its start
                    const uint64_t one = (ob ? (uint64_t)(off[p] - j) << (64 - ob) : 0ull) | (raw << 6) | (rec & 1);
                    if (off[p] < j || kc_rec_minhash(one, k, m, ob) != h) { printf("k=%d m=%d: a spilled window names another m-mer\n", k, m); return 1; }
                    back.insert({kmer_canon(raw, k), p + j});
                }
                p = q;
            }
            if (back != truth) { printf("k=%d m=%d: the windows out of the records are not the windows of the stream\n", k, m); return 1; }
            if (m == 16 && W >= 2 && W <= 6 && rmax >= W && n < (1u << 14)) {
                int bad = 0;
                switch (W) {
                case 2: bad = emit4_model<2>(s, k, valid, off, rng, &n_e4); break;
                case 3: bad = emit4_model<3>(s, k, valid, off, rng, &n_e4); break;
                case 4: bad = emit4_model<4>(s, k, valid, off, rng, &n_e4); break;
                case 5: bad = emit4_model<5>(s, k, valid, off, rng, &n_e4); break;
                default: bad = emit4_model<6>(s, k, valid, off, rng, &n_e4); break;
                }
                if (bad) return 1;
            }
        }
    }
    // k_kc_part's tile: few and many bins, empty bins (also at the thresholds 256 / 512 / 768), all records in one bin, null records
    for (int it = 0; it < 200; ++it) {
        const uint32_t n_bins = it < 5 ? 1u + (uint32_t)it : (it % 3 == 0 ? 1024u : 1u + (uint32_t)(rng() % 1024));
        const uint32_t n = it % 7 == 0 ? 8192u : (uint32_t)(rng() % 8193);
        const int mode = it % 5;
        std::vector<int> bins(n);
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t b = (uint32_t)(rng() % n_bins);
            if (mode == 1) b = b / 2 * 2 % n_bins;                                   // half the bins stay empty
            if (mode == 2) b = n_bins - 1;                                           // one bin takes everything
            if (mode == 3 && n_bins > 300) b = 250u + (uint32_t)(rng() % 20);        // around a threshold: 250 .. 269
            bins[i] = (mode == 4 && (rng() & 3) == 0) ? -1 : (int)b;                 // null records (a chunk's unused end)
        }
        if (!part_tile_model(bins, n_bins, rng)) return 1;
    }
    // placement inside a slice
    for (int i = 0; i < 200000; ++i) {
        const uint64_t key = rng();
        const uint32_t nh = 1 + (uint32_t)(rng() % 1024);
        if (kc_key_bucket(key, nh) >= nh) { printf("kc_key_bucket leaves the slice\n"); return 1; }
    }
    if (n_e4 == 0) { printf("k_kc_emit4's model has not run\n"); return 1; }
    printf("ok %zu (K, m) cases, %zu windows in %zu records; %zu runs through the model of k_kc_emit4\n", n_cases, n_windows, n_records, n_e4);
    return 0;
}

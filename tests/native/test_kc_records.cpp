// Host model of stage 00's partitioned counting (hast_amd/csrc/kc_common.h "records", "PLACEMENT"; kernels in kc_kernels.hip): windows
// of a byte stream -> records of minimizer runs (as k_kc_count<EMIT> cuts them) -> every record placed by ONE m-mer hash
// (kc_rec_minhash) -> its windows expanded again (as k_kc_apply does).  Checked against the per-window arithmetic of the direct path:
//   * the windows that come back out of the records are exactly the valid windows of the stream, each once, with its canonical key;
//   * a record's placement hash is the minimizer hash of EVERY window it holds (so the slice it is sent to is the one the direct
//     kernel's bucket_of_minhash would pick for each of them, and a k-mer has one slice wherever and on whichever strand it occurs);
//   * the record fits 64 bits for every K the path accepts (kc_run_max), the offset field never overlaps the bases;
//   * kc_key_bucket stays inside the slice.
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

int main() {
    std::mt19937_64 rng(20260404);
    size_t n_records = 0, n_windows = 0, n_cases = 0;
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
        }
    }
    // placement inside a slice
    for (int i = 0; i < 200000; ++i) {
        const uint64_t key = rng();
        const uint32_t nh = 1 + (uint32_t)(rng() % 1024);
        if (kc_key_bucket(key, nh) >= nh) { printf("kc_key_bucket leaves the slice\n"); return 1; }
    }
    printf("ok %zu (K, m) cases, %zu windows in %zu records\n", n_cases, n_windows, n_records);
    return 0;
}

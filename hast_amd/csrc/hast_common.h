// hast_common.h -- integer primitives shared by host C++ and gfx950 device code.
//
// K-mer arithmetic follows the reference's kmer/kmer.h semantics (A0 C1 T2 G3 via (c&6)>>1,
// first base most significant, canonical = min(fwd, revcomp)); the bit tricks are our own.
// Table hashing and the synthetic-workload generator are ours (nothing in the reference
// corresponds to them).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HAST_HD __host__ __device__ __forceinline__
#else
#define HAST_HD inline
#endif

namespace hast {

// ---- 2-bit k-mers (kmer.h:11-13, 129-166, 196-210) -----------------------------------------
HAST_HD uint32_t base_code(uint32_t c) { return (c & 6u) >> 1; }              // kmer.h:11
HAST_HD uint64_t kmer_mask(int k) { return k >= 32 ? ~0ull : (1ull << (2 * k)) - 1; }   // k <= 32

// reverse complement of a 2K-bit value: complement = code^2 (kmer.h:13), then reverse the order
// of the 2-bit groups and right-align (kmer.h:196-210).
HAST_HD uint64_t kmer_revcomp(uint64_t x, int k) {
    x ^= 0xAAAAAAAAAAAAAAAAull;
#if defined(__HIP_DEVICE_COMPILE__)
    x = __brevll(x);                                         // full bit reversal (2 x v_bfrev_b32)
    x = ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);  // un-swap pairs
#else
    x = ((x & 0x3333333333333333ull) << 2) | ((x >> 2) & 0x3333333333333333ull);
    x = ((x & 0x0F0F0F0F0F0F0F0Full) << 4) | ((x >> 4) & 0x0F0F0F0F0F0F0F0Full);
    x = __builtin_bswap64(x);
#endif
    return x >> (64 - 2 * k);
}

HAST_HD uint64_t kmer_canon(uint64_t fwd, int k) {
    uint64_t rc = kmer_revcomp(fwd, k);
    return fwd < rc ? fwd : rc;                                                // kmer.h:161-165
}

HAST_HD uint64_t kmer_pack(const char *s, int k) {                             // kmer.h:156-160
    uint64_t w = 0;
    for (int i = 0; i < k; i++) w = (w << 2) | base_code((uint8_t)s[i]);
    return w;
}

// ---- table geometry ---------------------------------------------------------------------------
#ifndef HAST_SLOTS
#define HAST_SLOTS 8
#endif
constexpr int      kSlotsPerBucket = HAST_SLOTS;   // 8 x 8 B = one 64-B line (16: one 128-B line)
constexpr int      kPieces = kSlotsPerBucket / 2;  // 16-B pieces per bucket
constexpr uint64_t kEmptySlot = ~0ull;             // (key<<2|tags) can never be all ones: the
                                                   // all-G k-mer is never canonical (all-C is smaller)
// K = 32 ("wide"): a key needs all 64 bits, so there is no room for tag bits.  The table is then TWO tables back to
// back, one per haplotype, whose slots hold the bare key; empty = all ones (G^32, never canonical) and an erased slot
// becomes kTombSlot = G^31 T, never canonical either (its reverse complement starts with A).
constexpr uint64_t kTombSlot = ~0ull - 1;

// Home bucket of a canonical key = f(minimizer of the key), so that the K-m+1.. consecutive windows of a
// read that share a minimizer probe the SAME 64-B line (the memory system merges them: one HBM request
// instead of several).  minimizer hash = min over the key's m-mers of hash32(canonical m-mer); it is
// strand-independent because an m-mer and its reverse complement have the same canonical form.
// m == k degenerates to plain hashing of the key.
HAST_HD uint32_t mmer_hash32(uint64_t canon_mmer) {
    return (uint32_t)((canon_mmer * 0x9E3779B97F4A7C15ull) >> 32);
}
HAST_HD uint32_t minimizer_hash(uint64_t kmer, int k, int m) {
    const uint64_t mm_mask = kmer_mask(m);
    uint32_t best = 0xFFFFFFFFu;
    for (int j = 0; j + m <= k; ++j) {
        uint64_t mm = (kmer >> (2 * (k - m - j))) & mm_mask;
        uint32_t h = mmer_hash32(kmer_canon(mm, m));
        best = h < best ? h : best;
    }
    return best;
}
// the minimum of several hashes is biased towards 0: re-spread it before range reduction
HAST_HD uint32_t bucket_of_minhash(uint32_t minh, uint32_t nbuckets) {
    return (uint32_t)(((uint64_t)(minh * 0x9E3779B1u) * nbuckets) >> 32);
}
// A minimizer owns a PAIR of adjacent buckets (one aligned 128-B block) and the key's last bit picks the half.  Real
// parent-specific k-mers come in runs of K windows around a variant site, for both alleles, so up to a dozen keys share
// a minimizer; spreading them over 16 slots keeps buckets from filling up, while every probe still loads only its own
// 64 B.  The windows of a read that share a minimizer now touch two lines of one 128-B block instead of one line, which
// costs no extra HBM requests (measured: TCC_EA0_RDREQ unchanged, same kernel time on random keys, -10 % time on
// clustered keys).  nbuckets is even.
HAST_HD uint32_t bucket_of(uint32_t minh, uint64_t key, uint32_t nbuckets) {
    return (bucket_of_minhash(minh, nbuckets >> 1) << 1) | (uint32_t)(key & 1);
}
HAST_HD uint32_t home_bucket(uint64_t key, int k, int m, uint32_t nbuckets) {
    return bucket_of(minimizer_hash(key, k, m), key, nbuckets);
}
// Probe sequence of a key: its home bucket (by minimizer), then -- only when that one is full -- a bucket chosen by the
// KEY's own hash, then the buckets after that one.  The jump matters for minimizers that very many keys share (in real
// genomes: poly-A and other low-complexity m-mers): walking on from the home bucket would put all their keys into ONE run
// of full buckets that every one of those keys has to cross (quadratic); after the jump they are spread over the whole
// table and a lookup that overflows costs one more random line.
HAST_HD uint32_t overflow_bucket(uint64_t key, uint32_t nbuckets) {
    const uint32_t h = (uint32_t)((key * 0xD6E8FEB86659FD93ull) >> 32);
    return (uint32_t)(((uint64_t)h * nbuckets) >> 32);
}
// bucket after `b` in the probe sequence of `key`; `step` = number of buckets probed so far (>= 1)
HAST_HD uint32_t next_bucket(uint32_t b, uint32_t step, uint64_t key, uint32_t nbuckets) {
#ifndef HAST_LINEAR_OVERFLOW           // (experiments only: the old walk-on-from-home behaviour)
    if (step == 1) return overflow_bucket(key, nbuckets);
#endif
    return b + 1 == nbuckets ? 0 : b + 1;
}

// ---- fingerprint filter (the structure k_classify_f probes; the exact table above is only consulted for its positives) ----
//
// Why: the probe kernel runs at the HBM random-request rate of the part, so the only way to get faster is
// fewer requests per read.  A request brings a 128-B block whatever is used of it; the exact table spends 8 B per key,
// so a block can only answer for ~3 keys and the placement function needs m = 16 (W = K-m+1 = 6 consecutive windows
// share a block at best).  The filter spends 2 B per key, so a block answers for dozens of keys, m can be 14 and the
// sampling function can be any FORWARD-strand scheme, because every key is filed twice -- under the sampled m-mer of
// its own string and under that of its reverse complement -- and a read window is looked up under the m-mer sampled
// from the window as it stands (no canonical m-mers anywhere):
//     block(window)      = scramble(sampled m-mer of the window's forward string)            (4^m blocks of 128 B)
//     sub-bucket, print  = from a hash of the window's K-mer as it stands                     (8 x 16 B, 8 prints each)
//   sampling = mod-minimizer (Groot Koerkamp & Pibiri 2024): the window's smallest t-mer (leftmost on ties), position x
//   among its K-t+1 t-mers, names the m-mer at position x mod W.  With t = r + (m-r) mod W, r = 4, the sampled m-mer stays
//   put for W consecutive windows and then jumps by W: density ~0.18 at K=21, m=14 (W=8, t=6) against 0.287 for the
//   exact table's random minimizers (m=16, W=6): 24 instead of 39 blocks per 150-bp read (tools/sim/filter_load_sim.cpp).
//   When the formula gives t = m the scheme IS the plain forward minimizer (large W).
// With PRINTS, a window whose print is in one of its sub-buckets, or whose sub-buckets are all full (then a key may not have
// found room), is a POSITIVE and is looked up in the exact table, which alone decides hits and tag bits; everything else is a
// proven miss.  So the filter can only cost time, never change a result.  Where a filed string fits a 16-bit entry exactly
// ("exact entries" below) a match is the hit itself and only full sub-buckets send a window to the table.
struct FilterGeom {
    int k, m, t;            // k-mer, sampled m-mer (m <= 15, 4^m blocks), ordering t-mer (t <= m)
    int kp;                 // the sampling only looks at the first kp bases of a window (kp <= k): W = kp-m+1 candidates
    int g;                  // entries per first-level minimum of the kernel's sliding window: min(4, kp-t+1)
    uint32_t wdiv;          // floor(2^16 / W) + 1: x / W for x < 64 by multiply-shift
    int exact;              // 1: the 16-bit entries are EXACT codes of the filed strings (below), one sub-bucket per string
    int choices;            // prints: sub-buckets a print may sit in (2, or 1 where the blocks are lightly loaded)
};
constexpr int kFilterSubs = 8;                    // 16-B sub-buckets per 128-B block
constexpr int kFilterPrints = 8;                  // 16-bit prints per sub-bucket
constexpr int kFilterMaxM = 15;                   // 4^15 blocks = 137 GB (4^14 = 34 GB)
HAST_HD uint32_t filter_w(const FilterGeom &g) { return (uint32_t)(g.kp - g.m + 1); }
HAST_HD uint32_t filter_nt(const FilterGeom &g) { return (uint32_t)(g.kp - g.t + 1); }
HAST_HD uint64_t filter_nblocks(const FilterGeom &g) { return 1ull << (2 * g.m); }
// low 32 bits of the product of the operands' low 24 bits.  On the device this is v_mul_u32_u24, a full-rate instruction;
// v_mul_lo_u32 / v_mul_hi_u32 run at a quarter of the rate, and the probe kernel is bound by VALU issue as much as by
// HBM requests, so the per-window arithmetic uses 24-bit products wherever the operands allow.
HAST_HD uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (a & 0xFFFFFFu) * (b & 0xFFFFFFu);
#endif
}
// as mul24, for call sites where the compiler proves both operands below 2^24, rewrites the product as a plain 32-bit multiply
// and then fails to select the 24-bit instruction for it (v_mul_lo_u32, a quarter of the rate): the instruction by name
HAST_HD uint32_t mul24_forced(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return (a & 0xFFFFFFu) * (b & 0xFFFFFFu);
#endif
}
// order of a t-mer: 20 hash bits above 12 position bits; smaller wins, equal t-mers (or equal hashes) -> the leftmost.
// (t-mers are at most 12 bases in every geometry filter_geom_for picks, so the 24-bit product sees the whole t-mer; a
// longer t-mer forced by an override is ordered by its last 12 bases -- still one order shared by build and probe.)
HAST_HD uint32_t tmer_order(uint32_t tmer, uint32_t pos) { return ((mul24(tmer + 1u, 0x9E3779u) >> 12) << 12) | pos; }
// position (0 .. W-1) of the m-mer that names the block of the K-mer string `fwd` (2K bits, first base most significant)
HAST_HD uint32_t filter_sample_pos(uint64_t fwd, const FilterGeom &g) {
    const uint32_t nt = filter_nt(g), tmask = (uint32_t)kmer_mask(g.t);
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t j = 0; j < nt; ++j) {
        const uint32_t e = tmer_order((uint32_t)(fwd >> (2 * (g.k - g.t - (int)j))) & tmask, j);
        best = e < best ? e : best;
    }
    const uint32_t x = best & 0xFFFu;
    return x - ((x * g.wdiv) >> 16) * filter_w(g);
}
// block of an m-mer: a bijective scramble of its 2m bits, so no two m-mers share a block.  The first half of the m-mer is
// folded into the second: the busiest blocks are those whose m-mer ENDS in one of the lowest-ordered t-mers, and the low
// address bits (which pick channel and bank) must not be a function of those last bases alone.
HAST_HD uint32_t filter_block_of(uint32_t mmer, int m) { return (mmer ^ (mmer >> m)) & (uint32_t)kmer_mask(m); }
HAST_HD uint32_t filter_block_of_string(uint64_t fwd, const FilterGeom &g) {
    const uint32_t p = filter_sample_pos(fwd, g);
    return filter_block_of((uint32_t)(fwd >> (2 * (g.k - g.m - (int)p))) & (uint32_t)kmer_mask(g.m), g.m);
}
// hash of a K-mer string (one strand of a key): sub-buckets = top bits, print = 16 bits from the middle, never 0 (0 = free
// slot).  One 32-bit multiply (a quarter-rate instruction on the device) over a fold of the K-mer's two halves; the xor-shift
// behind it carries the well-mixed top bits down into the print.
HAST_HD uint32_t filter_keyhash(uint64_t kmer) {
    const uint32_t lo = (uint32_t)kmer, hi = (uint32_t)(kmer >> 32);
    uint32_t h = (lo ^ ((hi << 15) | (hi >> 17))) * 0x9E3779B1u;          // rotate: one v_alignbit
    return h ^ (h >> 15);
}
HAST_HD uint32_t filter_sub_of(uint32_t keyhash) { return keyhash >> 29; }
// Two-choice filing: a print may sit in either of two sub-buckets of its block (the less loaded one at build time; both
// are in the same 128-B block, so a look-up still costs one request).  Only a window whose TWO sub-buckets are full has to
// ask the table.  At 400M keys in 4^14 blocks this takes the forced look-ups from 0.54 per read to none, at 800M 31-mers from
// 11 % of the windows to 2.8 % (tools/sim/filter_load_sim.cpp).  Lightly loaded filters file a print once (FilterGeom::choices).
HAST_HD uint32_t filter_sub2_of(uint32_t keyhash) { return (keyhash >> 26) & 7u; }
HAST_HD uint32_t filter_print_of(uint32_t keyhash) {
    const uint32_t f = (keyhash >> 8) & 0xFFFFu;
    return f ? f : 1u;
}
// ---- exact entries (FilterGeom::exact) -----------------------------------------------------------------------------------
// A block is named by a BIJECTION of the sampled m-mer, so a filed string is known completely by its block, the position
// pm of the m-mer inside it and the K-m bases outside the m-mer: 2(K-m) + log2(W) bits.  When those fit 17 bits (K = 21 with
// m = 14, W = 8: 14 + 3) a 16-bit entry can hold them EXACTLY instead of a 16-bit print: the code goes through a bijection of
// its 17 bits, whose top 3 bits pick THE sub-bucket (one choice: the sub-bucket is part of the code) and whose low 14 bits
// are stored above the key's 2 tag bits.  An entry that matches a window then IS the window's string with its tags: a hit
// needs no look-up in the exact table (with prints every hit costs a second HBM request -- 1.3 per read on random keys, 11 per
// read on keys with real-data structure), and a window whose sub-bucket holds no match and is not full is a proven miss.
// Only a window that lands in a FULL sub-bucket without a match still asks the table (a key may have found no room).
HAST_HD int filter_pos_bits(const FilterGeom &g) {
    int b = 0;
    while ((1u << b) < filter_w(g)) ++b;
    return b;
}
// (m <= 14: block * 8 + sub-bucket then fits 32 bits, which the probe's address arithmetic relies on)
HAST_HD bool filter_exact_fits(const FilterGeom &g) { return g.k <= 32 && g.m <= 14 && 2 * (g.k - g.m) + filter_pos_bits(g) <= 17; }
// the 17-bit code of a string whose sampled m-mer sits at pm: (pm, the bases behind the m-mer, the bases in front of it), scrambled
HAST_HD uint32_t filter_exact_code(uint64_t fwd, uint32_t pm, const FilterGeom &g) {
    const int rb = 2 * (g.k - g.m);
    // rotate the m-mer out: only the low rb bits of either term matter (fwd >> 2(K-pm) = the pm bases in front of the m-mer)
    const uint32_t rest = ((uint32_t)fwd << (2 * pm)) | (uint32_t)(fwd >> (2 * ((uint32_t)g.k - pm)));
    const uint32_t code = (rest & ((1u << rb) - 1u)) | (pm << rb);
    return mul24_forced(code, 0x1D2C5u) & 0x1FFFFu;                   // odd multiplier: a bijection of the 17 bits
}
HAST_HD uint32_t filter_exact_sub(uint32_t code17) { return code17 >> 14; }
HAST_HD uint32_t filter_exact_entry(uint32_t code17, uint32_t tags) { return ((code17 & 0x3FFFu) << 2) | tags; }   // tags 1..3: never 0
// Geometry for K and a key count (m = 15, a 137-GB filter, only where the key count asks for it: above 537M keys).
// Geometry for K and a key count.  Measured with tools/sim/filter_load_sim.cpp (unscaled: 400M keys, both strands filed):
// what limits m from below is not the average load of a block but the skew of the sampling -- the sampled m-mers all hold
// one of the window's lowest-ordered t-mers, so a fraction of the blocks takes most of the keys.  4^m >= 0.67 N keeps the
// windows that land in a full sub-bucket (and must ask the table) under ~0.5 per 150-bp read:
//     N = 400M: m = 14, kp = 21 -> W = 8, t = 6: 25.5 blocks + 0.5 forced look-ups per read (m = 13, t = 4: 23 + 77)
//     N = 100M: m = 13, kp = 21 -> W = 9, t = 4: 23.4 + 0.9          (prints; with exact entries K = 21 always gets m = 14)
//     N = 800M: m = 15, kp = 23 -> W = 9, t = 6 (K = 31): 1.5 strings per block, prints filed once, nothing forced
// kp = min(K, m + 8): longer windows (K = 31) are sampled on their first kp bases only -- more candidates would lower the
// density further but pile the keys on even fewer blocks.
// exact_mode: -1 = exact entries where they fit (and, for tables of 16M keys and more, the m that makes them fit: m = K-7,
// i.e. 14 at K = 21 -- a 34-GB filter next to 288 GB of HBM), 0 = prints always.
HAST_HD FilterGeom filter_geom_for(int k, uint64_t n_keys, int m_override, int t_override, int kp_override = 0, int exact_mode = -1) {
    FilterGeom g;
    g.k = k;
    int m = k < 8 ? k : 8;
    while (m < k && m < kFilterMaxM && (1ull << (2 * m)) * 3 < 2 * n_keys) ++m;
    if (exact_mode && n_keys >= (16ull << 20) && k - 7 > m && k - 7 <= 14) m = k - 7;
    // long windows (K >= m + 9: the sampling keeps W = 9 either way) get the m that lets prints be filed ONCE (<= 2.2 strings per
    // block, `choices` below): the probe then loads and compares one sub-bucket per window, and nothing is forced to the table
    while (m < kFilterMaxM && k >= m + 9 && 2.0 * (double)n_keys > 2.2 * (double)(1ull << (2 * m))) ++m;
    if (m_override >= 1 && m_override <= k && m_override <= kFilterMaxM) m = m_override;
    g.m = m;
    int kp = k < m + 8 ? k : m + 8;
    if (kp_override >= m && kp_override <= k && kp_override - m < 32) kp = kp_override;
    g.kp = kp;
    const int w = kp - m + 1, r = m < 4 ? m : 4;
    int t = r + (m - r) % w;
    if (t_override >= 1 && t_override <= m) t = t_override;
    g.t = t;
    const int nt = kp - t + 1;
    g.g = nt < 4 ? nt : 4;
    g.wdiv = (65536u / (uint32_t)w) + 1u;
    g.exact = (exact_mode && filter_exact_fits(g)) ? 1 : 0;
    // One choice where a block holds 2.2 strings or fewer on average (both strands filed): 800M 31-mers in 4^15 blocks (1.5 per
    // block) leave no window to a full sub-bucket with one choice (tools/sim/filter_load_sim.cpp), and the probe saves its
    // second load and compare -- config 5 is bound by VALU issue.  At 3 per block (400M keys, m = 14) one choice sends 0.5
    // windows per read to the table and 6 per block (800M keys, m = 14) 11 % of them: two choices there.
    g.choices = (2.0 * (double)n_keys <= 2.2 * (double)(1ull << (2 * m))) ? 1 : 2;
    return g;
}

// ---- synthetic workload ----------------------------------------------------------
HAST_HD uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
HAST_HD uint64_t synth_rand(uint64_t seed, uint64_t a, uint64_t b) {
    return splitmix64(splitmix64(seed ^ splitmix64(a)) + b);
}

struct SynthParams {          // mirrors hast_synth_params with defaults resolved
    uint64_t seed_k, seed_r, seed_b;
    uint64_t n_keys_per_hap;
    uint32_t n_barcodes, read_len, k, reserved;
};

// key j of haplotype h.
//   reserved == 0: canonical(random 2K bits); duplicates / keys in both sets occur by chance.
//   reserved == 1 ("clustered", the structure of REAL parent-specific k-mers): keys come in runs of K overlapping
//     windows around a variant site.  Site s = j / K has a random context of 2K-1 bases whose middle base is the
//     haplotype's allele (hap 1 = hap 0's allele + 1 mod 4); key j is window j % K of that context.  Neighbouring
//     keys then share minimizers, within and across the two haplotypes -- the hard case for bucket placement.
HAST_HD uint64_t synth_context_window(const SynthParams &p, int hap, uint64_t site, uint32_t t) {
    // context bases c[0 .. 2K-2], 2 bits each, drawn from two 64-bit words (62 + 62 bits hold up to 61 bases)
    const uint32_t K = p.k;
    const uint64_t w0 = synth_rand(p.seed_k, 0x5173ull, 2 * site), w1 = synth_rand(p.seed_k, 0x5173ull, 2 * site + 1);
    uint64_t v = 0;
    for (uint32_t i = 0; i < K; ++i) {
        const uint32_t pos = t + i;                                   // 0 .. 2K-2
        uint32_t base = (uint32_t)(((pos < 31) ? (w0 >> (2 * pos)) : (w1 >> (2 * (pos - 31)))) & 3);
        if (pos == K - 1) base = (base + (uint32_t)hap) & 3;          // the variant: different allele per haplotype
        v = (v << 2) | base;
    }
    return v;
}
HAST_HD uint64_t synth_key(const SynthParams &p, int hap, uint64_t j) {
    if (p.reserved & 1) return kmer_canon(synth_context_window(p, hap, j / p.k, (uint32_t)(j % p.k)), (int)p.k);
    return kmer_canon(synth_rand(p.seed_k, (uint64_t)hap, j) & kmer_mask((int)p.k), (int)p.k);
}

HAST_HD uint32_t synth_barcode(const SynthParams &p, uint64_t read) {
    return (uint32_t)(synth_rand(p.seed_b, read, 0) % p.n_barcodes);
}

// One read: uniform random ACGT, then n in {0..3} planted parental k-mers (80 % from the
// barcode's true haplotype = id&1, random strand, random offset), then with p = 1/200 one 'N'.
// Writes read_len ASCII bytes.
HAST_HD void synth_read_bc(const SynthParams &p, uint64_t read, uint32_t bc, uint8_t *out) {
    const uint32_t L = p.read_len, K = p.k;
    for (uint32_t j0 = 0; j0 < L; j0 += 32) {
        uint64_t w = synth_rand(p.seed_r, read, j0 >> 5);
        uint32_t n = (L - j0 < 32) ? (L - j0) : 32;
        for (uint32_t j = 0; j < n; j++) out[j0 + j] = (uint8_t)"ACGT"[(w >> (2 * j)) & 3];
    }
    if (L >= K && p.n_keys_per_hap > 0 && !(p.reserved & 2)) {        // reserved bit 1: reads without planted k-mers
        const uint32_t nplant = (uint32_t)(synth_rand(p.seed_r, read, 1000) & 3);
        for (uint32_t t = 0; t < nplant; t++) {
            uint64_t w = synth_rand(p.seed_r, read, 1001 + t);
            int truth = (int)(bc & 1u);
            int hap = ((w & 0xFF) < 205) ? truth : 1 - truth;
            if ((p.reserved & 1) && L >= 2 * K - 1) {
                // clustered mode: plant a site's whole context with this haplotype's allele (up to K consecutive hits)
                const uint64_t site = ((w >> 32) % p.n_keys_per_hap) / K;
                const uint32_t off = (uint32_t)((w >> 16) & 0xFFFF) % (L - (2 * K - 1) + 1);
                for (uint32_t tt = 0; tt < K; tt += K - 1) {           // two windows cover the 2K-1 bases: t=0 and t=K-1
                    const uint64_t v = synth_context_window(p, hap, site, tt);
                    for (uint32_t j = 0; j < K; j++) out[off + tt + j] = (uint8_t)"ACTG"[(v >> (2 * (K - 1 - j))) & 3];
                }
                continue;
            }
            uint32_t off = (uint32_t)((w >> 16) & 0xFFFF) % (L - K + 1);
            uint64_t key = synth_key(p, hap, (w >> 32) % p.n_keys_per_hap);
            if ((w >> 8) & 1) key = kmer_revcomp(key, (int)K);
            for (uint32_t j = 0; j < K; j++)
                out[off + j] = (uint8_t)"ACTG"[(key >> (2 * (K - 1 - j))) & 3];   // kmer.h:12 int2base
        }
    }
    uint64_t w = synth_rand(p.seed_r, read, 2000);
    if (w % 200 == 0) out[(uint32_t)(w >> 16) % L] = 'N';
}
HAST_HD void synth_read(const SynthParams &p, uint64_t read, uint8_t *out, uint32_t *barcode_out) {
    const uint32_t bc = synth_barcode(p, read);
    *barcode_out = bc;
    synth_read_bc(p, read, bc, out);
}

}  // namespace hast

// kc_common.h -- stage 00 (parent-unique k-mer sets): integer helpers shared by host C++ and gfx950 device code.
//
// The count table works on the SAME canonical keys as the stage-01 table (hast_common.h: A0 C1 T2 G3, canonical =
// numeric min of k-mer and reverse complement).  The reference's stage 00 prints what jellyfish prints
// (build_unshared_kmers.sh:283-284): the representative that is smaller in the order A<C<G<T.  The set {k-mer,
// reverse complement} is the same under both conventions, only the printed member may differ, so keys are converted
// when they leave the table (kc_to_print_key).
#pragma once
#include "hast_common.h"

namespace hast {

constexpr int kKcSlots = 8;                       // keys per bucket
constexpr int kKcBucketWords = 16;                // 8 keys (64 B) + 8 x u32 paternal (32 B) + 8 x u32 maternal (32 B) = 128 B
constexpr uint32_t kKcHistoHigh = 10000;          // jellyfish histo default --high (analysis_kmercount.sh:7-9)

// 2-bit codes A0 C1 T2 G3 -> A0 C1 G2 T3 (swap codes 2 and 3: low bit ^= high bit)
HAST_HD uint64_t kc_recode(uint64_t x) { return x ^ ((x >> 1) & 0x5555555555555555ull); }
// table key -> the key jellyfish would print: min over both strands in the order A<C<G<T, in that coding
HAST_HD uint64_t kc_to_print_key(uint64_t key, int k) {
    const uint64_t a = kc_recode(key), b = kc_recode(kmer_revcomp(key, k));
    return a < b ? a : b;
}
// and back: printed key -> table key
HAST_HD uint64_t kc_from_print_key(uint64_t pk, int k) { return kmer_canon(kc_recode(pk), k); }   // the recoding is an involution

// ---- records of the partitioned counting path (kc_kernels.hip) -----------------------------------------------------------------
// A record = the bases of a run of consecutive windows that share their minimizer (the SAME m-mer occurrence, so that a run is at
// most W = K - m + 1 windows), packed into 64 bits:
//   [offset of that m-mer in the run's first window : ob bits at the top][bases: 2 (K + run - 1) bits, first base most
//   significant, ending at bit 6][run - 1 : 5 bits][parent : 1 bit]
// The offset is what makes a record cheap to place: its bucket follows from ONE canonical m-mer hash instead of the minimum over W
// of them (the partition passes and the LDS pass each place every record once; recomputing the minimum took most of their time).
HAST_HD uint32_t kc_rec_off_bits(int k, int m) {
    uint32_t b = 0;
    while ((1u << b) < (uint32_t)(k - m + 1)) ++b;
    return b;
}
// longest run a record has room for (0: none -- K too large for this path)
HAST_HD uint32_t kc_run_max(int k, int m) {
    const int r = (58 - (int)kc_rec_off_bits(k, m)) / 2 - k + 1, w = k - m + 1;
    const int c = r < w ? r : w;
    return c < 1 ? 0u : (c > 16 ? 16u : (uint32_t)c);
}
// THE ORDER OF M-MERS (which of a window's m-mers is its minimizer) and what names a record's slice of the table: 28 hash bits above
// four zero bits.  The emit kernel (k_kc_emit4) puts a position into those four bits and takes ONE unsigned minimum per m-mer for
// "smallest hash, leftmost on ties"; everything placed by a minimizer is placed by this VALUE, never by the m-mer itself, so two
// different m-mers with the same value in one window -- each strand would pick its own leftmost -- still name the same slice.
// 24-bit multiplies (full-rate instructions, kc_mul24 below): the 64-bit multiply-shift of stage 01's table (mmer_hash32) is two
// quarter-rate multiplies per position on a kernel that is bound by VALU issue.
HAST_HD uint32_t kc_mul24(uint32_t a, uint32_t b) { return mul24_forced(a, b); }      // (hast_common.h: the instruction by name -- where the
                                                                                         // compiler proves an operand below 2^24 it rewrites the product as a
                                                                                         // plain 32-bit multiply and selects the quarter-rate instruction)
HAST_HD uint32_t kc_mmer_hash32(uint32_t canon_mmer) {                                  // m <= 16
    uint32_t h = kc_mul24(canon_mmer, 0x9E3779u) ^ kc_mul24(canon_mmer >> 8, 0x85EBCBu);
    h ^= h >> 15;
    return h & ~15u;
}
HAST_HD uint32_t kc_mmer_hash(uint64_t canon_mmer) {                                    // any m; == kc_mmer_hash32 where the m-mer fits 32 bits
    const uint32_t a = (uint32_t)canon_mmer, b = (uint32_t)(canon_mmer >> 32);
    uint32_t h = kc_mul24(a, 0x9E3779u) ^ kc_mul24(a >> 8, 0x85EBCBu) ^ kc_mul24(b, 0xC2B2AFu) ^ (kc_mul24(b >> 8, 0x27D4EBu) << 7);
    h ^= h >> 15;
    return h & ~15u;
}
// bits 24 .. 47 of the 48-bit product of two 24-bit numbers, i.e. floor(a x b / 2^24) for a, b < 2^24: two full-rate instructions
// + a funnel shift (a 32 x 32 -> 64-bit product is two quarter-rate ones)
HAST_HD uint32_t kc_mul24_shr24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t lo, hi;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    return __builtin_amdgcn_alignbit(hi, lo, 24);
#else
    return (uint32_t)(((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu)) >> 24);
#endif
}
// the SLICE (fine bin) of the table a minimizer value names, of n_fine < 2^24 slices.  The minimum of W hashes leans towards zero: the
// value's 24 low bits (they do not) are spread by an odd multiplier first, then range-reduced.  Every slice must hold the same
// number of buckets for this to load them evenly: hast_kc_create rounds the table down to whole slices.
HAST_HD uint32_t kc_fine_of_hash(uint32_t minh, uint32_t n_fine) { return kc_mul24_shr24(kc_mul24(minh >> 4, 0x9E3779u) & 0xFFFFFFu, n_fine); }
// the minimizer value of a k-mer (any strand): the smallest kc_mmer_hash over its canonical m-mers
HAST_HD uint32_t kc_minimizer_hash(uint64_t kmer, int k, int m) {
    uint32_t best = 0xFFFFFFFFu;
    for (int j = 0; j + m <= k; ++j) {
        const uint32_t h = kc_mmer_hash(kmer_canon((kmer >> (2 * (k - m - j))) & kmer_mask(m), m));
        best = h < best ? h : best;
    }
    return best;
}
HAST_HD uint32_t kc_rec_minhash(uint64_t rec, int k, int m, uint32_t ob) {
    const uint32_t run = (uint32_t)((rec >> 1) & 31) + 1, off = ob ? (uint32_t)(rec >> (64 - ob)) : 0u;
    const uint64_t first = ((rec >> 6) >> (2 * (run - 1))) & kmer_mask(k);
    return kc_mmer_hash(kmer_canon((first >> (2 * ((uint32_t)(k - m) - off))) & kmer_mask(m), m));
}

// ---- the per-lane arithmetic of k_kc_emit4 (kc_kernels.hip), shared with its host model (tests/native/test_kc_records.cpp) -----------
// the m-mer hashes of positions q .. q + 3 (m = 16) from x = the 32 bases behind q, first base most significant
HAST_HD uint32_t kc_e4_funnel(uint32_t hi, uint32_t lo, uint32_t sh) {           // the low dword of (hi:lo) >> sh, 0 < sh < 32
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
}
HAST_HD void kc_e4_mmer_hashes(uint64_t x, uint32_t h[4]) {
    uint64_t y = x ^ 0xAAAAAAAAAAAAAAAAull;                                       // the reverse complement of all 32 bases (kmer_revcomp without its shift)
#if defined(__HIP_DEVICE_COMPILE__)
    y = __brevll(y);
    y = ((y & 0x5555555555555555ull) << 1) | ((y >> 1) & 0x5555555555555555ull);
#else
    y = ((y & 0x3333333333333333ull) << 2) | ((y >> 2) & 0x3333333333333333ull);
    y = ((y & 0x0F0F0F0F0F0F0F0Full) << 4) | ((y >> 4) & 0x0F0F0F0F0F0F0F0Full);
    y = __builtin_bswap64(y);
#endif
    const uint32_t xh = (uint32_t)(x >> 32), xl = (uint32_t)x, yh = (uint32_t)(y >> 32), yl = (uint32_t)y;
    uint32_t f = xh, r = yl;
    h[0] = kc_mmer_hash32(f < r ? f : r);
    f = kc_e4_funnel(xh, xl, 30), r = kc_e4_funnel(yh, yl, 2);
    h[1] = kc_mmer_hash32(f < r ? f : r);
    f = kc_e4_funnel(xh, xl, 28), r = kc_e4_funnel(yh, yl, 4);
    h[2] = kc_mmer_hash32(f < r ? f : r);
    f = kc_e4_funnel(xh, xl, 26), r = kc_e4_funnel(yh, yl, 6);
    h[3] = kc_mmer_hash32(f < r ? f : r);
}
// the minima of the four windows of W hashes that start at c[0] .. c[3]; c[j] = hash | j: the minimum names its place in the block
template <int WT>
HAST_HD void kc_e4_minima(const uint32_t c[9], uint32_t w[4]) {
    static_assert(WT >= 2 && WT <= 6, "nine values");
    auto mn = [](uint32_t a, uint32_t b) { return a < b ? a : b; };
    if (WT == 2) {
        for (int k = 0; k < 4; ++k) w[k] = mn(c[k], c[k + 1]);
        return;
    }
    uint32_t mid = 0xFFFFFFFFu;                                                  // positions 3 .. W - 1: in all four windows
    for (int j = 3; j < WT; ++j) mid = mn(mid, c[j]);
    const uint32_t s2 = c[2], s1 = mn(c[1], s2), s0 = mn(c[0], s1);
    const uint32_t q1 = c[WT], q2 = mn(q1, c[WT + 1]), q3 = mn(q2, c[WT + 2]);
    w[0] = mn(s0, mid);
    w[1] = mn(mn(s1, mid), q1);
    w[2] = mn(mn(s2, mid), q2);
    w[3] = mn(mid, q3);
}
// bit k: window k goes on with the run of the window in front of it (both valid, the same minimizer position).  vm = the lane's four
// validity bits, at[k] = where in the lane's block window k's minimizer starts, prev = (at[3] | valid[3] << 4) of the lane in front
HAST_HD uint32_t kc_e4_cont(uint32_t vm, const uint32_t at[4], bool has_prev, uint32_t prev) {
    uint32_t cn = 0;
    if (has_prev && (vm & 1u) && (prev >> 4) && (prev & 15u) == at[0] + 4u) cn = 1u;
    for (int k = 1; k < 4; ++k)
        if (((vm >> k) & 1u) && ((vm >> (k - 1)) & 1u) && at[k] == at[k - 1]) cn |= 1u << k;
    return cn;
}
// windows that go on behind window k of a lane: C = the continuation bits of this lane | the next << 4 | the one after << 8
HAST_HD uint32_t kc_e4_run_minus_1(uint32_t C, int k) {
    const uint32_t x = ~(C >> (k + 1));
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_ctz(x);
#else
    return (uint32_t)__builtin_ctz(x);
#endif
}
// a run's descriptor: window (tile-relative, < 2^14) | run - 1 << 14 | offset of the minimizer in the run's first window << 18
HAST_HD uint32_t kc_e4_desc(uint32_t p, uint32_t runm1, uint32_t off) { return p | (runm1 << 14) | (off << 18); }

// PLACEMENT.  Direct counting (one atomic per minimizer run) files a key in its minimizer's bucket, then the next three, then from a
// bucket of the key's own hash on (kc_kernels.hip kc_probe): the windows of a read that share a minimizer touch one 128-B line.
// Partitioned counting only needs the minimizer to pick the SLICE of 2^fine_shift buckets a record goes to -- the slice is in LDS
// while it is counted -- so inside the slice a key is filed by its OWN hash: bucket kc_key_bucket, then the next three (wrapping
// inside the slice), then the same overflow walk.  A minimizer's keys (~8.5 per occupied minimizer at 30x, error k-mers included)
// then no longer pile up in one bucket while seven in eight stay empty, and a wave's 64 windows rarely need a second bucket.
// A table is filled by one of the two ways only (hast_kc_create decides), and its readers are scans.
// The hash: 24-bit multiplies only.  k_kc_apply is bound by VALU issue (2.99e10 VALU instructions x 4 cycles = its 50 ms), and a
// 32 x 32-bit multiply takes a quarter-rate instruction on this part where a 24-bit one (v_mul_u32_u24) takes a full-rate one: the
// 64-bit product + remix + range reduction of round 4 were five quarter-rate multiplies per window, 80 of its ~570 cycles.
HAST_HD uint32_t kc_key_hash(uint64_t key) {
    const uint32_t a = (uint32_t)key, b = (uint32_t)(key >> 32);
    uint32_t h = kc_mul24(a, 0x9E3779u) ^ kc_mul24(a >> 8, 0x85EBCBu) ^ kc_mul24(b, 0xC2B2AFu) ^ (kc_mul24(b >> 8, 0x27D4EBu) << 7);
    return h ^ (h >> 15);         // (a second round of multiplies, round 5a, bought nothing: 12 instead of 8 windows of 10 G find their four buckets full without it)
}
HAST_HD uint32_t kc_bucket_of_hash(uint32_t h, uint32_t n_here) { return kc_mul24(h >> 16, n_here) >> 16; }      // n_here <= 2^16: slices of 512 or 1024 buckets
HAST_HD uint32_t kc_key_bucket(uint64_t key, uint32_t n_here) { return kc_bucket_of_hash(kc_key_hash(key), n_here); }
// the tag of a key in a slice's LDS copy (k_kc_apply): 7 other bits of the same hash under a set top bit (0 = the slot is empty)
HAST_HD uint32_t kc_tag_of_hash(uint32_t h) { return ((h >> 3) & 0x7Fu) | 0x80u; }

// slice of the key space a window belongs to (decided by its minimizer, so a bucket never mixes slices)
HAST_HD uint32_t kc_slice_of(uint32_t minh, uint32_t n_slices) {
    uint32_t h = minh * 0x85EBCA6Bu;
    h ^= h >> 15;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return (uint32_t)(((uint64_t)h * n_slices) >> 32);
}

// ---- synthetic trio (bench + tests): two parental genomes that differ from a common random genome by SNPs -------
// Reads are sampled at random positions of the parent's genome, random strand, with substitution errors; the byte
// stream holds the reads back to back, each followed by '\n'.
struct KcSynth {
    uint64_t seed;
    uint64_t genome_len;      // bases
    uint32_t read_len;
    uint32_t snp_per_1024;    // parent-specific SNP rate (per 1024 bases)
    uint32_t err_per_4096;    // sequencing substitution errors (per 4096 bases)
    uint32_t n_per_4096;      // reads that get one 'N' (per 4096 reads)
};
HAST_HD uint32_t kc_genome_base(const KcSynth &g, int parent, uint64_t x) {
    uint32_t b = (uint32_t)(synth_rand(g.seed, 0x67656E6Full, x >> 5) >> (2 * (x & 31))) & 3;
    const uint64_t h = synth_rand(g.seed, 0x736E7000ull + (uint64_t)parent, x);
    if ((h & 1023) < g.snp_per_1024) b = (b + 1 + (uint32_t)((h >> 10) % 3)) & 3;       // a different base
    return b;
}
// byte j (0 .. read_len) of read i of `parent`; byte read_len is the separator
HAST_HD uint8_t kc_synth_byte(const KcSynth &g, int parent, uint64_t i, uint32_t j) {
    if (j >= g.read_len) return (uint8_t)'\n';
    const uint64_t r = synth_rand(g.seed, 0x72656164ull + (uint64_t)parent, i);
    const uint64_t start = (r >> 1) % (g.genome_len - g.read_len + 1);
    const bool rev = r & 1;
    uint32_t b = rev ? 3 - kc_genome_base(g, parent, start + (g.read_len - 1 - j)) : kc_genome_base(g, parent, start + j);
    const uint64_t e = synth_rand(g.seed, 0x65727200ull + (uint64_t)parent, i * 4096 + j);
    if ((e & 4095) < g.err_per_4096) b = (b + 1 + (uint32_t)((e >> 12) % 3)) & 3;
    const uint64_t nn = synth_rand(g.seed, 0x6E6E6E00ull + (uint64_t)parent, i);
    if ((nn & 4095) < g.n_per_4096 && (uint32_t)((nn >> 12) % g.read_len) == j) return (uint8_t)'N';
    return (uint8_t)"ACGT"[b];
}

}  // namespace hast

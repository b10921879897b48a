// gz_core.h -- deflate (RFC 1951) decoding of ONE chunk of a gzip stream, written once for the device and for the host
// (tests/native/test_gz_core.cpp runs these functions chunk by chunk against zlib, so the arithmetic and the rules for where
// a chunk ends are checked on the CPU before they cost GPU time).  On the device (gz_kernels.hip) the search's strict header
// parse, every block's header parse and table construction are these functions run by one lane; the symbol loop of
// k_gz_decode is decode_huffman re-shaped for a whole wave (uniform bit buffer, tables in LDS, matches copied by all lanes).
//
// What it is for (reference: gzstream.h:47, classify.cpp:245-254 -- one zlib stream per input file read
// through a 303-byte buffer): the drop-in CLI's .fq.gz inputs are inflated ON THE GPU, so that the compressed bytes cross
// PCIe (5.5 x fewer) and the FASTQ framer reads the inflated bytes where they lie in HBM.  The method is the one
// par_inflate.h uses on host threads (pugz: Kerbiriou & Chikhi 2019; rapidgzip: Knespel & Brunst 2023), re-shaped for
// 10^4 .. 10^5 lanes in flight instead of 16 threads:
//   search   every chunk of the compressed bytes looks for its first dynamic-block header (gz_candidate + a strict parse);
//   decode   from there a lane decodes blocks into 16-bit symbols with the 32 KB in front of the chunk UNKNOWN: a symbol is
//            a literal byte, or 0x8000 + i = "byte i of the window in front of this chunk"; it stops at the first block
//            boundary at or behind its stop position, at a member's end, or where room or input run out;
//   chain    (host) a chunk is accepted iff the chunk in front of it ended exactly where it started -- by induction from
//            the stream's first block every accepted chunk starts at a true boundary; holes are decoded by follow-up jobs
//            with a known start;
//   windows, translate, CRC-32: gz_kernels.hip.
// Table entries are those of fast_inflate.h (bits 0-7 code bits to drop, 8-12 extra-bit count / sub-table bits, 13 literal,
// 14 end of block, 15 sub-table pointer, 16-31 value).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GZ_HD __host__ __device__ inline
#else
#define GZ_HD inline
#endif

namespace hast {
namespace gz {

constexpr uint32_t kLit = 1u << 13, kEob = 1u << 14, kSub = 1u << 15;
// first-level table sizes as zlib's (9 / 6 bits): the tables of a block live in LDS on the device, 7 KB per wave, and the number of
// waves a CU holds is what the decode kernel's throughput scales with
constexpr int kLitRoot = 9, kDistRoot = 6;
constexpr uint32_t kLitTabCap = 856, kDistTabCap = 592, kPreTabCap = 128;       // entries (u32): first level + sub-tables (zlib's own bounds for complete codes
                                                                                 // are 852 and 592 at these roots -- inftrees.h ENOUGH_LENS / ENOUGH_DISTS; an
                                                                                 // incomplete code is a lone one); a code that needs more is refused (kErrTableSize)
constexpr uint32_t kTabWords = kLitTabCap + kDistTabCap + kPreTabCap;           // a lane's tables, back to back
constexpr uint32_t kWindow = 32768;
constexpr uint16_t kMarker = 0x8000;

// status bits of a decoded chunk
enum : uint32_t {
    kStFound = 1,          // a start was given or found
    kStStop = 2,           // ended at a block boundary at or behind stop_bit
    kStFinal = 4,          // ended behind a final block: end_bit is the first bit behind it (the member's trailer follows, byte aligned)
    kStNoRoom = 8,         // the next block did not fit the symbol buffer: ended at the last boundary in front of it
    kStStarved = 16,       // the next block runs past the end of the input that is there
    kStError = 32,         // the next block is not valid deflate (err_code says why); ended at the last boundary in front of it
    kStNoBlock = 64        // not even one block was completed
};
enum : uint32_t {
    kErrNone = 0, kErrBlockType, kErrStoredLen, kErrCounts, kErrPreCode, kErrRepeat, kErrTooManyLens, kErrNoEob, kErrOverSub,
    kErrIncomplete, kErrTableSize, kErrLitCode, kErrDistCode, kErrTooFar
};

struct ChunkJob {               // 80 bytes; one per chunk, device memory
    uint64_t from_bit;          // in: known start (flags & 1) or where the search starts
    uint64_t stop_bit;          // in: decode until the first block boundary >= this
    uint64_t sym_off;           // in: first symbol of this chunk's buffer in the arena (u16 units); with kJobPoolSlot: the first symbol of the
                                //     pass's slot POOL -- the decode kernel takes the next slot of sym_cap symbols and writes the chunk's own offset back
    uint32_t sym_cap;           // in: symbols of room
    uint32_t flags;             // in: 1 = from_bit is a proven block start (no search), 2 = nothing in front may be copied (a member starts here)
    uint64_t start_bit;         // out
    uint64_t end_bit;           // out: the last block boundary reached
    uint32_t n_out;             // out: symbols decoded up to end_bit
    uint32_t status;            // out
    uint32_t err_code;          // out
    uint32_t search_to_lo;      // in: the search gives up at from_bit + this many bits
    // in: where the compressed bytes lie when the device holds a RING of them, not the whole file (gz_api.cpp "ring"): word i of the
    // file is w[i - in_adj_words] for this job, and bits at and behind limit_bits (0: no such limit) are not there for it
    uint64_t in_adj_words;
    uint64_t limit_bits;
};
static_assert(sizeof(ChunkJob) == 80, "ChunkJob is copied to and from the device as it is");
// kJobPoolSlot (round 6): 4 chunks in 10 of a FASTQ hold no block start and decode nothing; a pass's arena therefore holds fewer slots than
// the pass has chunks, and a chunk that HAS a start takes the next free one (one atomic per chunk).  When the pool is empty the chunk
// reports "found, no room, no block decoded", which the chain already knows: a follow-up job decodes from its start with room of its own.
constexpr uint32_t kJobKnown = 1, kJobNoHistory = 2, kJobPoolSlot = 4;

// ---- bit input: aligned 32-bit words, LSB first; the buffer holds >= 2 zero words behind its last real bit -------------------
struct Bits {
    const uint32_t *w;
    uint64_t nbits;             // real bits in the buffer
    uint64_t bb;
    uint64_t wp;                // next word to load
    uint32_t bc;                // valid bits in bb
};
GZ_HD void seek(Bits &b, uint64_t bit) {
    b.wp = bit >> 5;
    const uint32_t sh = (uint32_t)(bit & 31);
    const uint64_t nw = (b.nbits + 31) >> 5;
    b.bb = (b.wp < nw + 2 ? (uint64_t)b.w[b.wp] : 0ull) >> sh;
    b.wp++;
    b.bc = 32 - sh;
}
GZ_HD void refill(Bits &b) {    // >= 32 valid bits afterwards (zeros behind the end of the buffer)
    if (b.bc <= 32) {
        const uint64_t nw = (b.nbits + 31) >> 5;
        const uint64_t v = b.wp < nw + 2 ? (uint64_t)b.w[b.wp] : 0ull;
        b.bb |= v << b.bc;
        b.wp++;
        b.bc += 32;
    }
}
GZ_HD uint32_t take(Bits &b, uint32_t n) {
    const uint32_t v = (uint32_t)(b.bb & ((1ull << n) - 1));
    b.bb >>= n;
    b.bc -= n;
    return v;
}
GZ_HD uint64_t pos(const Bits &b) { return b.wp * 32 - b.bc; }
GZ_HD bool overran(const Bits &b) { return pos(b) > b.nbits; }

GZ_HD uint32_t rev_bits(uint32_t v, int n) {            // the low n (<= 16) bits of v, reversed
    v = ((v & 0x5555u) << 1) | ((v >> 1) & 0x5555u);
    v = ((v & 0x3333u) << 2) | ((v >> 2) & 0x3333u);
    v = ((v & 0x0F0Fu) << 4) | ((v >> 4) & 0x0F0Fu);
    v = ((v & 0x00FFu) << 8) | ((v >> 8) & 0x00FFu);
    return v >> (16 - n);
}

GZ_HD uint32_t len_base(int i) {
    const uint16_t t[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    return t[i];
}
GZ_HD uint32_t len_extra(int i) { return i < 8 || i == 28 ? 0u : (uint32_t)((i - 4) >> 2); }
GZ_HD uint32_t dist_base(int i) {
    const uint16_t t[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    return t[i];
}
GZ_HD uint32_t dist_extra(int i) { return i < 4 ? 0u : (uint32_t)((i - 2) >> 1); }

// zlib's rule (inftrees.c): a code must be complete, except (lone_ok) a single code of length 1, or no code at all
GZ_HD bool complete(const uint8_t *lens, int n, bool lone_ok) {
    int count[16];
    for (int l = 0; l < 16; ++l) count[l] = 0;
    int used = 0, maxl = 0;
    for (int i = 0; i < n; ++i)
        if (lens[i]) {
            count[lens[i]]++;
            ++used;
            if (lens[i] > maxl) maxl = lens[i];
        }
    if (used == 0) return lone_ok;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false;
    }
    return left == 0 || (lone_ok && maxl == 1);
}

// What a header parse indexes by a value it has just read (code lengths, counts per length, next code per length): on the device a
// private array indexed like that lives in scratch memory -- k_gz_decode keeps ONE of these in LDS for its lane 0 (no scratch, and
// no registers held across the symbol loop), the search's strict parse and the host keep it on their stacks.
struct HdrScratch {
    uint8_t lens[286 + 30 + 4];
    uint8_t pre[20];
    int count[16];
    uint32_t first[16], next[16];
};

// lens[0..n): code lengths (0 = unused).  kind 0: literal/length alphabet, 1: distance alphabet, 2: code-length alphabet.
// tab: cap entries; first level = 2^root entries, sub-tables behind it.  No per-symbol scratch: the canonical codes are
// regenerated in symbol order by every pass.  Returns kErrNone or what is wrong.
// (on the device a function of its own, called three times per block header by k_gz_decode's lane 0: inlined into the kernel its
// registers would be the kernel's -- 4 of the symbol loop's values spilled to scratch memory)
#if defined(__HIPCC__)
__attribute__((noinline))
#endif
GZ_HD uint32_t build_table(const uint8_t *lens, int n, int kind, uint32_t *tab, uint32_t cap, int root, HdrScratch &scr) {
    int *const count = scr.count;
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    count[0] = 0;
    int left = 1, used = 0;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return kErrOverSub;
        used += count[l];
    }
    const uint32_t nroot = 1u << root;
    for (uint32_t k = 0; k < nroot; ++k) tab[k] = 0;           // 0 = invalid code
    if (used == 0) return kErrNone;                             // e.g. a block without distance codes
    uint32_t *const first = scr.first, *const next = scr.next;
    {
        uint32_t code = 0;
        first[0] = 0;
        for (int l = 1; l <= 15; ++l) {
            code = (code + (uint32_t)count[l - 1]) << 1;
            first[l] = code;
        }
    }
    // pass 1: sub-table size per first-level prefix = longest code with that prefix (kept in the prefix's own entry: no
    // short code can own it, the code is prefix-free)
    bool any_long = false;
    for (int l = root + 1; l <= 15; ++l) any_long = any_long || count[l] != 0;
    if (any_long) {
        for (int l = 0; l < 16; ++l) next[l] = first[l];
        for (int i = 0; i < n; ++i) {
            const int l = lens[i];
            if (!l) continue;
            const uint32_t c = rev_bits(next[l]++, l);
            if (l > root) {
                const uint32_t p = c & (nroot - 1), b = (uint32_t)(l - root);
                if (tab[p] < b) tab[p] = b;
            }
        }
        uint32_t cursor = nroot;
        for (uint32_t p = 0; p < nroot; ++p)
            if (tab[p]) {
                const uint32_t b = tab[p], size = 1u << b;
                if (cursor + size > cap || cursor >= 65536u) return kErrTableSize;
                for (uint32_t k = 0; k < size; ++k) tab[cursor + k] = 0;
                tab[p] = kSub | (b << 8) | (uint32_t)root | (cursor << 16);
                cursor += size;
            }
    }
    // pass 2: the entries
    for (int l = 0; l < 16; ++l) next[l] = first[l];
    for (int i = 0; i < n; ++i) {
        const int l = lens[i];
        if (!l) continue;
        const uint32_t c = rev_bits(next[l]++, l);
        uint32_t e;
        if (kind == 0) {
            if (i < 256) e = kLit | ((uint32_t)i << 16);
            else if (i == 256) e = kEob;
            else if (i < 286) e = (len_extra(i - 257) << 8) | (len_base(i - 257) << 16);
            else continue;                                       // 286, 287 never occur in valid data: left invalid
        } else if (kind == 1) {
            if (i >= 30) continue;
            e = (dist_extra(i) << 8) | (dist_base(i) << 16);
        } else e = (uint32_t)i << 16;
        if (l <= root) {
            e |= (uint32_t)l;
            for (uint32_t k = c; k < nroot; k += 1u << l) tab[k] = e;
        } else {
            const uint32_t ptr = tab[c & (nroot - 1)];
            const uint32_t sb = (ptr >> 8) & 31, start = ptr >> 16;
            e |= (uint32_t)(l - root);
            for (uint32_t k = c >> root; k < (1u << sb); k += 1u << (l - root)) tab[start + k] = e;
        }
    }
    return kErrNone;
}

struct Tables {                 // views into one lane's kTabWords words
    uint32_t *lit, *dist, *pre;
};
GZ_HD Tables tables_at(uint32_t *words) { return Tables{words, words + kLitTabCap, words + kLitTabCap + kDistTabCap}; }

// the header of a dynamic block behind its 3 type bits: code lengths -> tables.  strict: zlib's completeness rules (what the
// search demands of a candidate).  Returns kErrNone or the reason; the caller looks at overran() for "input ended".
// build = false: only the checks (the search's strict parse of a candidate needs no literal / distance tables: building them is
// most of the work -- ~300 candidates per chunk reach this point when no block starts in it)
GZ_HD uint32_t read_dynamic(Bits &in, Tables &t, bool strict, bool build, HdrScratch &scr) {
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    refill(in);
    const int hlit = (int)take(in, 5) + 257, hdist = (int)take(in, 5) + 1, hclen = (int)take(in, 4) + 4;
    if (hlit > 286 || hdist > 30) return kErrCounts;
    uint8_t *const pre = scr.pre;
    for (int i = 0; i < 19; ++i) pre[i] = 0;
    for (int i = 0; i < hclen; ++i) {
        refill(in);
        pre[order[i]] = (uint8_t)take(in, 3);
    }
    if (strict && !complete(pre, 19, false)) return kErrPreCode;
    if (uint32_t bad = build_table(pre, 19, 2, t.pre, kPreTabCap, 7, scr)) return bad;
    uint8_t *const lens = scr.lens;
    int i = 0;
    const int total = hlit + hdist;
    while (i < total) {
        if (overran(in)) return kErrPreCode;
        refill(in);
        const uint32_t e = t.pre[in.bb & 127];
        if ((e & 0xFF) == 0) return kErrPreCode;
        take(in, e & 0xFF);
        const int sym = (int)(e >> 16);
        if (sym < 16) lens[i++] = (uint8_t)sym;
        else {
            int rep;
            uint8_t v = 0;
            if (sym == 16) {
                if (i == 0) return kErrRepeat;
                v = lens[i - 1];
                rep = 3 + (int)take(in, 2);
            } else if (sym == 17) rep = 3 + (int)take(in, 3);
            else rep = 11 + (int)take(in, 7);
            if (i + rep > total) return kErrTooManyLens;
            for (int k = 0; k < rep; ++k) lens[i + k] = v;
            i += rep;
        }
    }
    if (lens[256] == 0) return kErrNoEob;
    if (strict && (!complete(lens, hlit, true) || !complete(lens + hlit, hdist, true))) return kErrIncomplete;
    if (!build) return kErrNone;
    if (uint32_t bad = build_table(lens, hlit, 0, t.lit, kLitTabCap, kLitRoot, scr)) return bad;
    if (uint32_t bad = build_table(lens + hlit, hdist, 1, t.dist, kDistTabCap, kDistRoot, scr)) return bad;
    return kErrNone;
}
GZ_HD void fixed_tables(Tables &t, HdrScratch &scr) {
    uint8_t *const l = scr.lens;
    for (int i = 0; i < 144; ++i) l[i] = 8;
    for (int i = 144; i < 256; ++i) l[i] = 9;
    for (int i = 256; i < 280; ++i) l[i] = 7;
    for (int i = 280; i < 288; ++i) l[i] = 8;
    (void)build_table(l, 288, 0, t.lit, kLitTabCap, kLitRoot, scr);
    for (int i = 0; i < 30; ++i) l[i] = 5;
    (void)build_table(l, 30, 1, t.dist, kDistTabCap, kDistRoot, scr);
}

// Cheap test of bit position `bit` as the header of a non-final dynamic block: type bits, code counts, and a code-length code
// that is complete (sum of 2^(7-len) == 128).  v = the 64 bits at `bit` (>= 57 valid), v2 = the 64 bits at bit + 56.
GZ_HD bool candidate(uint64_t v, uint64_t v2) {
    if ((v & 7) != 4) return false;                                             // BFINAL = 0, BTYPE = 10
    if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) return false;             // HLIT, HDIST
    const uint32_t hclen = (uint32_t)((v >> 13) & 15) + 4;
    uint64_t w = v >> 17;                                                       // 39 bits = 13 lengths
    uint32_t sum = 0, i = 0;
    for (; i < hclen && i < 13; ++i, w >>= 3) sum += (w & 7) ? (128u >> (w & 7)) : 0u;
    w = v2;                                                                     // the lengths from the 14th on start at bit + 17 + 39 = bit + 56
    for (; i < hclen; ++i, w >>= 3) sum += (w & 7) ? (128u >> (w & 7)) : 0u;
    return sum == 128;
}
GZ_HD uint64_t bits_at(const uint32_t *w, uint64_t bit) {                       // the 64 bits at `bit` (the buffer is padded)
    const uint64_t wi = bit >> 5;
    const uint32_t sh = (uint32_t)(bit & 31);
    const uint64_t lo = (uint64_t)w[wi] | ((uint64_t)w[wi + 1] << 32);
    return sh ? (lo >> sh) | ((uint64_t)w[wi + 2] << (64 - sh)) : lo;
}

// The symbols of one Huffman block.  sym = the chunk's own buffer (symbol 0 = the chunk's first output symbol); a copy
// from in front of it yields markers, or (no_history) is an error.  Returns 0 = end of block reached, else a status bit.
GZ_HD uint32_t decode_huffman(Bits &in, const Tables &t, uint16_t *sym, uint32_t &n_out, uint32_t cap, bool no_history, uint32_t &err) {
    constexpr uint32_t LM = (1u << kLitRoot) - 1, DM = (1u << kDistRoot) - 1;
    uint32_t n = n_out;
    for (;;) {
        if (n + 260 > cap) { n_out = n; return kStNoRoom; }
        if (overran(in)) { n_out = n; return kStStarved; }
        refill(in);
        uint32_t e = t.lit[in.bb & LM];
        if (e & kSub) {
            in.bb >>= kLitRoot;
            in.bc -= kLitRoot;
            e = t.lit[(e >> 16) + (uint32_t)(in.bb & ((1u << ((e >> 8) & 31)) - 1))];
        }
        in.bb >>= (e & 0xFF);
        in.bc -= (e & 0xFF);
        if (e & kLit) {
            sym[n++] = (uint16_t)(e >> 16);
            continue;
        }
        if ((e & 0xFF) == 0) { err = kErrLitCode; n_out = n; return kStError; }
        if (e & kEob) { n_out = n; return 0; }
        const uint32_t leb = (e >> 8) & 31;
        const uint32_t len = (e >> 16) + (uint32_t)(in.bb & ((1u << leb) - 1));
        in.bb >>= leb;
        in.bc -= leb;
        refill(in);
        uint32_t d = t.dist[in.bb & DM];
        if (d & kSub) {
            in.bb >>= kDistRoot;
            in.bc -= kDistRoot;
            d = t.dist[(d >> 16) + (uint32_t)(in.bb & ((1u << ((d >> 8) & 31)) - 1))];
        }
        if ((d & 0xFF) == 0) { err = kErrDistCode; n_out = n; return kStError; }
        in.bb >>= (d & 0xFF);
        in.bc -= (d & 0xFF);
        const uint32_t deb = (d >> 8) & 31;
        const uint32_t distance = (d >> 16) + (uint32_t)(in.bb & ((1u << deb) - 1));
        in.bb >>= deb;
        in.bc -= deb;
        if (distance > n) {
            if (no_history || distance > kWindow) { err = kErrTooFar; n_out = n; return kStError; }
            // the copy starts in the (unknown) window in front of this chunk: byte j of that window = marker kMarker + j
            uint32_t k = 0;
            const uint32_t from_window = distance - n < len ? distance - n : len;
            for (; k < from_window; ++k) sym[n + k] = (uint16_t)(kMarker + (kWindow - (distance - n) + k));
            for (; k < len; ++k) sym[n + k] = sym[n + k - distance];
        } else {
            const uint16_t *src = sym + n - distance;
            for (uint32_t k = 0; k < len; ++k) sym[n + k] = src[k];
        }
        n += len;
    }
}

// The ring of a chunk's most recent symbols (k_gz_decode keeps it in LDS; tests/native/test_gz_core.cpp models it): kRing
// symbols, position p in slot p & (kRing - 1).  A round of <= 64 symbols reads its sources before it writes: everything from
// kRing positions back is there.  A long match is written 64 symbols a step, each step reading before it writes: symbol k comes
// from k - distance when the match does not overlap itself (distance >= length: there while distance <= kRing), from the first
// `distance` symbols over and over when it does (those must survive the up to 256 symbols written before the last step).
constexpr uint32_t kRing = 512;
GZ_HD bool ring_holds_long_match(uint32_t distance, uint32_t len) { return distance >= len ? distance <= kRing : distance <= kRing - 256; }

// ---- token-parallel decoding (k_gz_decode): what ONE LANE does with the 64 bits of the stream at its own bit offset ----------------
// A token = one or two literals, one match (length code + extra bits + distance code + extra bits: at most 15 + 5 + 15 + 13 = 48
// bits) or the end-of-block code.  The wave parses a token at each of 64 consecutive bit offsets at once -- most of them are not
// token starts -- and then walks the chain offset 0 -> 0 + its token's bits -> ... with one scalar read per token, instead of one
// dependent table look-up per symbol.  info: bits 0-6 = the token's bits (1 .. 48), bits 7.. = kind (0 = literal(s) or a match
// of at most 64 symbols: the chain walks on; others stop it).
// TWO literals per look-up where both codes fit the root bits (pair_entry: bit 8 of a literal's first-level entry says "a second
// literal in bits 24..31", the length is that of both codes): the bases of a FASTQ record are literals of ~2.2 bits each, quality
// values mostly fit in pairs too -- half the tokens for the chain to walk.
// parse_token_fast looks at the FIRST-LEVEL tables only (two dependent look-ups instead of four): a code longer than the root bits
// -- rare by construction -- makes a kTokSlow token, which stops the chain; the step that starts at it parses in full.
constexpr uint32_t kTokEob = 1, kTokLong = 2, kTokErrLit = 3, kTokErrDist = 4, kTokSlow = 5;
struct Token {
    uint32_t info;              // bits | kind << 7
    uint32_t olen;              // symbols it stands for (literals 1 or 2, match 3 .. 258, end of block 0)
    uint32_t val;               // the literal byte(s): first | second << 8
    uint32_t dist;              // match: distance (1 .. 32768), else 0
};
// what first-level entry idx of a literal/length table becomes: itself, or itself and the literal behind it (computed from the table
// as build_table left it: every entry is worked out before any is replaced)
GZ_HD uint32_t pair_entry(const uint32_t *lit, uint32_t idx) {
    const uint32_t e1 = lit[idx], l1 = e1 & 0xFF;
    if ((e1 & (kLit | kSub)) == kLit && l1 > 0 && l1 < (uint32_t)kLitRoot) {
        const uint32_t e2 = lit[idx >> l1], l2 = e2 & 0xFF;     // (the bits behind the first code, zeros above them: an entry whose
        // code is no longer than the bits that are really there does not depend on those zeros)
        if ((e2 & (kLit | kSub)) == kLit && l2 > 0 && l1 + l2 <= (uint32_t)kLitRoot)
            return kLit | (1u << 8) | (l1 + l2) | (e1 & 0x00FF0000u) | ((e2 & 0x00FF0000u) << 8);
    }
    return e1;
}
template <bool kFull>
GZ_HD Token parse_token_t(uint64_t v, const uint32_t *lit, const uint32_t *dst) {
    constexpr uint32_t LM = (1u << kLitRoot) - 1, DM = (1u << kDistRoot) - 1;
    uint32_t e = lit[(uint32_t)v & LM], used = 0;
    if (e & kSub) {
        if (!kFull) return Token{1u | (kTokSlow << 7), 0u, 0u, 0u};
        used = (uint32_t)kLitRoot;
        e = lit[(e >> 16) + ((uint32_t)(v >> kLitRoot) & ((1u << ((e >> 8) & 31)) - 1))];
    }
    const uint32_t cl = e & 0xFF;
    used += cl;
    if (e & kLit) return Token{used, 1u + ((e >> 8) & 1u), e >> 16, 0u};
    if (cl == 0) return Token{1u | (kTokErrLit << 7), 0u, 0u, 0u};
    if (e & kEob) return Token{used | (kTokEob << 7), 0u, 0u, 0u};
    const uint32_t leb = (e >> 8) & 31;
    const uint32_t len = (e >> 16) + ((uint32_t)(v >> used) & ((1u << leb) - 1));
    used += leb;
    uint32_t d = dst[(uint32_t)(v >> used) & DM];
    if (d & kSub) {
        if (!kFull) return Token{1u | (kTokSlow << 7), 0u, 0u, 0u};
        used += (uint32_t)kDistRoot;
        d = dst[(d >> 16) + ((uint32_t)(v >> used) & ((1u << ((d >> 8) & 31)) - 1))];
    }
    const uint32_t dl = d & 0xFF;
    if (dl == 0) return Token{1u | (kTokErrDist << 7), 0u, 0u, 0u};
    used += dl;
    const uint32_t deb = (d >> 8) & 31;
    const uint32_t distance = (d >> 16) + ((uint32_t)(v >> used) & ((1u << deb) - 1));
    used += deb;
    return Token{used | (len > 64 ? kTokLong << 7 : 0u), len, 0u, distance};
}
GZ_HD Token parse_token(uint64_t v, const uint32_t *lit, const uint32_t *dst) { return parse_token_t<true>(v, lit, dst); }
GZ_HD Token parse_token_fast(uint64_t v, const uint32_t *lit, const uint32_t *dst) { return parse_token_t<false>(v, lit, dst); }

// Blocks from job.start_bit on, until the first boundary >= job.stop_bit, a final block, or where room / input / validity end.
// Everything is committed boundary by boundary.  w / nbits: the whole compressed buffer; tabs: this lane's kTabWords words.
GZ_HD void decode_chunk(ChunkJob &job, const uint32_t *w, uint64_t nbits, uint32_t *tabs, uint16_t *sym) {
    Tables t = tables_at(tabs);
    HdrScratch scr;
    const bool no_history = (job.flags & kJobNoHistory) != 0;
    uint64_t at = job.start_bit;
    uint32_t n = 0, status = kStFound, err = kErrNone;
    bool any = false;
    for (;;) {
        if (at >= job.stop_bit && (any || !(job.flags & kJobKnown))) {
            // a boundary at or behind the stop: the chunk ends here -- unless what follows is a block no search can find (stored,
            // fixed, or final: the chunk behind this one starts at the first NON-FINAL DYNAMIC header), which this chunk takes too
            // (pigz and bgzip put an empty stored block between their pieces: without this, every such place costs a follow-up job)
            bool hidden = false;
            if (any && at + 3 <= nbits) hidden = (bits_at(w, at) & 7) != 4;
            if (!hidden) { status |= kStStop; break; }
        }
        if (at + 3 > nbits) { status |= kStStarved; break; }
        Bits in{w, nbits, 0, 0, 0};
        seek(in, at);
        refill(in);
        const uint32_t final = take(in, 1), type = take(in, 2);
        uint32_t n2 = n, bad = 0;
        if (type == 0) {
            const uint64_t byte = (pos(in) + 7) >> 3;
            if ((byte + 4) * 8 > nbits) { status |= kStStarved; break; }
            seek(in, byte * 8);
            refill(in);
            const uint32_t len = take(in, 16);
            refill(in);
            const uint32_t nlen = take(in, 16);
            if ((len ^ 0xFFFFu) != nlen) { status |= kStError; err = kErrStoredLen; break; }
            if ((byte + 4 + len) * 8 > nbits) { status |= kStStarved; break; }
            if (n + len + 4 > job.sym_cap) { status |= kStNoRoom; break; }
            const uint8_t *bytes = reinterpret_cast<const uint8_t *>(w) + byte + 4;
            for (uint32_t k = 0; k < len; ++k) sym[n + k] = bytes[k];
            n2 = n + len;
            seek(in, (byte + 4 + len) * 8);
        } else if (type == 3) {
            status |= kStError;
            err = kErrBlockType;
            break;
        } else {
            if (type == 1) fixed_tables(t, scr);
            else bad = read_dynamic(in, t, false, true, scr);
            if (bad) {
                if (overran(in)) status |= kStStarved;          // an "error" read out of the zero padding: the block is not all here
                else { status |= kStError; err = bad; }
                break;
            }
            const uint32_t rc = decode_huffman(in, t, sym, n2, job.sym_cap, no_history, err);
            if (rc) {
                if (rc == kStError && overran(in)) status |= kStStarved;
                else status |= rc;
                break;
            }
            if (overran(in)) { status |= kStStarved; break; }
        }
        n = n2;
        at = pos(in);
        any = true;
        if (final) { status |= kStFinal; break; }
    }
    if (!any) status |= kStNoBlock;
    job.end_bit = at;
    job.n_out = n;
    job.status = status;
    job.err_code = (status & kStError) ? err : kErrNone;
}

// Strict parse of a candidate position (what the search accepts as a block start): true iff a whole dynamic header stands there.
// pre: kPreTabCap words of scratch for the code-length code's table (LDS on the device: ~300 look-ups per parse)
GZ_HD bool header_parses(const uint32_t *w, uint64_t nbits, uint64_t bit, uint32_t *pre) {
    Tables t{nullptr, nullptr, pre};
    HdrScratch scr;
    Bits in{w, nbits, 0, 0, 0};
    seek(in, bit + 3);
    return read_dynamic(in, t, true, false, scr) == kErrNone && !overran(in);
}

// The same verdict for a lane of k_gz_search, which runs 64 of these side by side: nothing is indexed by a value that was just read
// except 128 BYTES of LDS (the code-length code's table: symbol << 3 | bits; header_parses keeps 512 bytes of table and 1.4 KB of
// arrays per lane -- 37 KB of LDS and scratch memory for a wave, i.e. ONE wave per SIMD, and the search is a chain of dependent
// instructions per lane).  The counts per code length that the completeness rules need are kept in packed words, the lengths
// themselves are not kept at all (a repeat needs the previous one only).  pre8: 128 bytes of this lane's own.
constexpr uint32_t kPre8Stride = 132;                   // bytes from one lane's table to the next (33 words: the lanes' tables start in different banks)
struct LenCounts {                                      // how many codes of each length 1 .. 15: 12 bits each, five to a word
    uint64_t w0 = 0, w1 = 0, w2 = 0;
};
GZ_HD void len_counts_add(LenCounts &c, uint32_t len, uint32_t n) {      // len in 1 .. 15
    const uint32_t q = ((len - 1) * 13) >> 6, r = len - 1 - 5 * q;        // (len - 1) / 5 and the remainder
    const uint64_t inc = (uint64_t)n << (12 * r);
    c.w0 += q == 0 ? inc : 0;
    c.w1 += q == 1 ? inc : 0;
    c.w2 += q == 2 ? inc : 0;
}
GZ_HD bool len_counts_complete(const LenCounts &c) {     // complete(lens, n, true) from the counts
    int left = 1;
    uint32_t used = 0, ones = (uint32_t)(c.w0 & 0xFFF);
    for (int l = 0; l < 15; ++l) {
        const uint64_t wv = l < 5 ? c.w0 : l < 10 ? c.w1 : c.w2;
        const int n = (int)((wv >> (12 * (l % 5))) & 0xFFF);
        left = (left << 1) - n;
        if (left < 0) return false;
        used += (uint32_t)n;
    }
    return used == 0 || left == 0 || ones == used;
}
GZ_HD bool header_parses8(const uint32_t *w, uint64_t nbits, uint64_t bit, uint8_t *pre8) {
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    Bits in{w, nbits, 0, 0, 0};
    seek(in, bit + 3);
    refill(in);
    const uint32_t hlit = take(in, 5) + 257, hdist = take(in, 5) + 1, hclen = take(in, 4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    // the code-length code: 19 lengths of 3 bits, packed; counts per length 1 .. 7 in bytes
    uint64_t plen = 0, cnt = 0;
    for (uint32_t i = 0; i < hclen; ++i) {
        refill(in);
        const uint32_t l = take(in, 3);
        plen |= (uint64_t)l << (3 * order[i]);
        if (l) cnt += 1ull << (8 * l);
    }
    {   // complete (zlib's rule for this code: no exceptions), and the first code of every length
        int left = 1;
        for (int l = 1; l <= 7; ++l) {
            left = (left << 1) - (int)((cnt >> (8 * l)) & 0xFF);
            if (left < 0) return false;
        }
        if (left != 0) return false;
    }
    uint64_t next = 0;                                  // next code of length l in byte l
    {
        uint32_t code = 0;
        for (int l = 1; l <= 7; ++l) {
            code = (code + (uint32_t)((cnt >> (8 * (l - 1))) & 0xFF)) << 1;      // (byte 0 of cnt is 0)
            next |= (uint64_t)(code & 0xFF) << (8 * l);
        }
    }
    for (uint32_t i = 0; i < 19; ++i) {
        const uint32_t l = (uint32_t)(plen >> (3 * i)) & 7;
        if (!l) continue;
        const uint32_t code = (uint32_t)(next >> (8 * l)) & 0xFF;
        next += 1ull << (8 * l);
        const uint8_t e = (uint8_t)((i << 3) | l);
        for (uint32_t k = rev_bits(code, (int)l); k < 128; k += 1u << l) pre8[k] = e;      // (a complete code fills all 128)
    }
    // the literal/length and distance code lengths: counted, not kept
    LenCounts cl, cd;
    const uint32_t total = hlit + hdist;
    uint32_t i = 0, prev = 0, eob = 0;
    while (i < total) {
        if (overran(in)) return false;
        refill(in);
        const uint32_t e = pre8[in.bb & 127];
        take(in, e & 7);
        const uint32_t sym = e >> 3;
        uint32_t v, rep;
        if (sym < 16) { v = sym; rep = 1; }
        else if (sym == 16) {
            if (i == 0) return false;
            v = prev;
            rep = 3 + take(in, 2);
        } else if (sym == 17) { v = 0; rep = 3 + take(in, 3); }
        else { v = 0; rep = 11 + take(in, 7); }
        if (i + rep > total) return false;
        if (v) {
            const uint32_t n_lit = i < hlit ? (rep < hlit - i ? rep : hlit - i) : 0u;
            if (n_lit) len_counts_add(cl, v, n_lit);
            if (rep - n_lit) len_counts_add(cd, v, rep - n_lit);
        }
        if (i <= 256 && 256 < i + rep) eob = v;
        prev = v;
        i += rep;
    }
    if (eob == 0) return false;
    if (!len_counts_complete(cl) || !len_counts_complete(cd)) return false;
    return !overran(in);
}

// ---- CRC-32 (IEEE, reflected) as polynomial arithmetic over GF(2), after zlib's crc32.c (x2nmodp / multmodp) -----------------
constexpr uint32_t kCrcPoly = 0xEDB88320u;
GZ_HD uint32_t crc_multmodp(uint32_t a, uint32_t b) {          // a(x) * b(x) mod p(x); bit 31 = x^0
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}
GZ_HD uint32_t crc_x2nmodp(uint64_t n, uint32_t k) {           // x^(n * 2^k) mod p(x)
    uint32_t p = 1u << 31, sq = 1u << 30;                       // x^0, x^1
    for (uint32_t i = 0; i < k; ++i) sq = crc_multmodp(sq, sq);
    while (n) {
        if (n & 1) p = crc_multmodp(sq, p);
        n >>= 1;
        sq = crc_multmodp(sq, sq);
    }
    return p;
}
// crc32(A || B) from crc32(A), crc32(B) and xlen = x^(8 |B|) mod p
GZ_HD uint32_t crc_combine_op(uint32_t crc_a, uint32_t crc_b, uint32_t xlen) { return crc_multmodp(xlen, crc_a) ^ crc_b; }
GZ_HD uint32_t crc_table_entry(uint32_t i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ kCrcPoly : c >> 1;
    return c;
}

}  // namespace gz
}  // namespace hast

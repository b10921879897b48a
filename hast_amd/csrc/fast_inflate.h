// fast_inflate.h -- streaming gzip (RFC 1952 / 1951) decoder for the CLIs' ingest.
//
// Why: every real input of HAST is .fastq.gz, one inflate stream per file cannot be split, and with the GPU side at
// > 100 Gbp/s the end-to-end time of `classify` / `unshared_kmers` on gz input IS the inflate time (the reference
// reads through zlib with a 303-byte buffer, gzstream.h:47).  zlib's inflate decodes one symbol per loop with 9/6-bit
// first-level tables and byte-wise copies; this decoder uses a 64-bit branch-free bit buffer, 11/8-bit first-level
// tables whose entries carry base value + extra-bit count, literal runs without re-checking the buffer, and 8-byte
// match copies.  Members are checked against their CRC-32 and ISIZE trailers, so a decoding bug cannot pass as data.
// Concatenated members are decoded one after the other (as gzread does); bytes after the last member that do not start
// a new member are ignored (as gzread does).
//
// Usage: GzInflater z; z.open(FILE*); long n = z.read(buf, cap)  (> 0 bytes, 0 end of input, < 0 error: z.error()).
#pragma once
#include <immintrin.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace hast {

// ---- Huffman decoding tables (shared by the serial decoder below and the parallel one in par_inflate.h) ---------------
// table entry: bits 0-7 codeword bits to drop, 8-12 extra-bit count (or sub-table index bits), 13 literal, 14 end of block,
// 15 sub-table pointer, 16-31 value (literal byte / base length / base distance / sub-table start)
constexpr uint32_t kInflateLit = 1u << 13, kInflateEob = 1u << 14, kInflateSub = 1u << 15;
constexpr int kInflateLitBits = 11, kInflateDistBits = 8;
inline uint32_t inflate_rev_bits(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}
// lens[0..n): code lengths (0 = unused).  kind 0: literal/length alphabet, 1: distance alphabet, 2: code-length alphabet.
// returns nullptr, or what is wrong with the code
inline const char *inflate_build_table(const uint8_t *lens, int n, int kind, std::vector<uint32_t> &tab, int root) {
    constexpr uint32_t kLit = kInflateLit, kEob = kInflateEob, kSub = kInflateSub;
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    int count[16] = {0};
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    count[0] = 0;
    int left = 1, used = 0;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return "deflate: over-subscribed Huffman code";
        used += count[l];
    }
    tab.assign((size_t)1 << root, 0);                    // 0 = invalid code
    if (used == 0) return nullptr;                          // e.g. a block without distance codes
    uint32_t next[16];
    uint32_t code = 0;
    for (int l = 1; l <= 15; ++l) {
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    // sub-table size per first-level prefix = longest code with that prefix
    std::vector<uint8_t> sub_bits;
    std::vector<uint32_t> codes((size_t)n);
    bool any_long = false;
    for (int i = 0; i < n; ++i)
        if (lens[i]) {
            codes[i] = inflate_rev_bits(next[lens[i]]++, lens[i]);
            any_long |= lens[i] > root;
        }
    if (any_long) {
        sub_bits.assign((size_t)1 << root, 0);
        for (int i = 0; i < n; ++i)
            if (lens[i] > root) {
                uint8_t &b = sub_bits[codes[i] & ((1u << root) - 1)];
                b = std::max<uint8_t>(b, (uint8_t)(lens[i] - root));
            }
        for (size_t p = 0; p < sub_bits.size(); ++p)
            if (sub_bits[p]) {
                const size_t start = tab.size();
                tab.resize(start + ((size_t)1 << sub_bits[p]), 0);
                tab[p] = kSub | ((uint32_t)sub_bits[p] << 8) | (uint32_t)root | ((uint32_t)start << 16);
                if (start >> 16) return "deflate: Huffman table too large";
            }
    }
    for (int i = 0; i < n; ++i) {
        const int l = lens[i];
        if (!l) continue;
        uint32_t e;
        if (kind == 0) {
            if (i < 256) e = kLit | ((uint32_t)i << 16);
            else if (i == 256) e = kEob;
            else if (i < 286) e = ((uint32_t)len_extra[i - 257] << 8) | ((uint32_t)len_base[i - 257] << 16);
            else continue;                               // 286, 287 never occur in valid data: leave invalid
        } else if (kind == 1) {
            if (i >= 30) continue;
            e = ((uint32_t)dist_extra[i] << 8) | ((uint32_t)dist_base[i] << 16);
        } else e = (uint32_t)i << 16;                    // code-length alphabet: the symbol itself
        if (l <= root) {
            e |= (uint32_t)l;
            for (uint32_t k = codes[i]; k < (1u << root); k += 1u << l) tab[k] = e;
        } else {
            const uint32_t p = codes[i] & ((1u << root) - 1);
            const uint32_t ptr = tab[p];
            const uint32_t sb = (ptr >> 8) & 31, start = ptr >> 16;
            e |= (uint32_t)(l - root);
            for (uint32_t k = codes[i] >> root; k < (1u << sb); k += 1u << (l - root)) tab[start + k] = e;
        }
    }
    return nullptr;
}

class GzInflater {
  public:
    // continuation = true: the stream is the rest of a gzip file whose first members someone else has decoded (the
    // hand-over from BgzfReader): bytes that do not start a member are then trailing garbage, not a plain file
    bool open(FILE *f, size_t in_buf_bytes = 1u << 20, bool continuation = false) {
        fp_ = f;
        in_.assign(std::max<size_t>(in_buf_bytes, 64) + kBack + kPad, 0);
        in_pos_ = in_end_ = 0;
        in_eof_ = false;
        out_.assign(kWindow + kOutChunk + kSlack, 0);
        out_pos_ = out_have_ = crc_done_ = kWindow;
        member_out_ = 0;
        state_ = kMemberHeader;
        bitbuf_ = 0;
        bitcnt_ = 0;
        err_.clear();
        first_member_ = !continuation;
        (void)crc_state();
        build_fixed();
        return f != nullptr;
    }
    // up to cap bytes of decompressed data; 0 = end of input; < 0 = error
    long read(uint8_t *dst, size_t cap) {
        size_t got = 0;
        while (got < cap) {
            if (out_pos_ < out_have_) {
                const size_t n = std::min(cap - got, out_have_ - out_pos_);
                memcpy(dst + got, out_.data() + out_pos_, n);
                out_pos_ += n;
                got += n;
                continue;
            }
            if (state_ == kDone) break;
            if (state_ == kError) return -1;
            // slide the window: keep the last 32 KB in front of the next chunk
            if (out_have_ > kWindow) {
                account();
                memmove(out_.data(), out_.data() + out_have_ - kWindow, kWindow);
                out_pos_ = out_have_ = crc_done_ = kWindow;
            }
            if (!produce()) {
                state_ = kError;
                return -1;
            }
        }
        return (long)got;
    }
    const std::string &error() const { return err_; }
    // CRC-32 of the gzip trailer (carry-less-multiply folding where the CPU has it), for the other readers of this directory
    static uint32_t crc32(uint32_t crc, const uint8_t *p, size_t n) { return crc32_update(crc, p, n); }

  private:
    static constexpr size_t kWindow = 32768, kOutChunk = 1u << 20, kSlack = 512, kPad = 64, kBack = 8;
    static constexpr int kLitBits = kInflateLitBits, kDistBits = kInflateDistBits;
    static constexpr uint32_t kLit = kInflateLit, kEob = kInflateEob, kSub = kInflateSub;
    enum State { kMemberHeader, kBlockHeader, kStored, kHuffman, kTrailer, kRaw, kDone, kError };

    bool fail(const char *what) {
        err_ = what;
        return false;
    }

    // ---- input ---------------------------------------------------------------------------------------------
    // make at least `want` (<= kPad) real or padded bytes readable at in_pos_; real bytes come first
    void fill_input() {
        if (in_eof_) return;
        // Keep kBack bytes of history in front of the read position: the bit buffer may still hold up to 7 whole bytes
        // that align_to_byte() hands back by stepping in_pos_ backwards.
        const size_t back = std::min(in_pos_, kBack);
        const size_t from = in_pos_ - back, keep = in_end_ - from;
        if (keep && from) memmove(in_.data(), in_.data() + from, keep);
        in_pos_ = back;
        in_end_ = keep;
        const size_t room = in_.size() - kPad - in_end_;
        const size_t n = room ? fread(in_.data() + in_end_, 1, room, fp_) : 0;
        in_end_ += n;
        if (n < room) {
            in_eof_ = true;
            memset(in_.data() + in_end_, 0, kPad);           // zero padding lets the bit reader run past the end safely
        }
    }
    size_t in_left() const { return in_end_ - in_pos_; }
    // byte-aligned reads (headers, trailers, stored blocks): first give back whole bytes the bit buffer holds
    void align_to_byte() {
        const unsigned drop = bitcnt_ & 7;
        bitbuf_ >>= drop;
        bitcnt_ -= drop;
        in_pos_ -= bitcnt_ >> 3;                             // un-read the whole bytes still in the bit buffer
        bitbuf_ = 0;
        bitcnt_ = 0;
    }
    int get_byte() {
        if (in_pos_ >= in_end_) {
            fill_input();
            if (in_pos_ >= in_end_) return -1;
        }
        return in_[in_pos_++];
    }
    // ---- bits ------------------------------------------------------------------------------------------------
    static uint64_t load64(const uint8_t *p) {
        uint64_t v;
        memcpy(&v, p, 8);
        return v;                                            // little-endian hosts only (x86-64)
    }
    void refill() {                                          // at least 56 bits afterwards (input is zero padded)
        bitbuf_ |= load64(in_.data() + in_pos_) << bitcnt_;
        in_pos_ += (63 - bitcnt_) >> 3;
        bitcnt_ |= 56;
    }
    // the decoder may have loaded bytes beyond the real end of input (zero padding): consumed bits beyond it = truncated
    bool overran() const { return (uint64_t)in_pos_ * 8 > (uint64_t)in_end_ * 8 + bitcnt_; }
    uint32_t take(unsigned n) {                              // n <= 32, caller refilled
        const uint32_t v = (uint32_t)(bitbuf_ & ((1ull << n) - 1));
        bitbuf_ >>= n;
        bitcnt_ -= n;
        return v;
    }
    void ensure_input() {                                    // before a refill: >= kPad/2 readable bytes (real or padding)
        if (in_left() < 32 && !in_eof_) fill_input();
    }

    bool build(const uint8_t *lens, int n, int kind, std::vector<uint32_t> &tab, int root) {
        const char *bad = inflate_build_table(lens, n, kind, tab, root);
        return bad ? fail(bad) : true;
    }
    void build_fixed() {
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        build(l, 288, 0, fixed_lit_, kLitBits);
        uint8_t d[30];
        for (int i = 0; i < 30; ++i) d[i] = 5;
        build(d, 30, 1, fixed_dist_, kDistBits);
    }
    bool read_dynamic() {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        ensure_input();
        refill();
        const int hlit = (int)take(5) + 257, hdist = (int)take(5) + 1, hclen = (int)take(4) + 4;
        if (hlit > 286 || hdist > 30) return fail("deflate: bad code counts");
        uint8_t pre[19] = {0};
        for (int i = 0; i < hclen; ++i) {
            if (bitcnt_ < 3) { ensure_input(); refill(); }
            pre[order[i]] = (uint8_t)take(3);
        }
        std::vector<uint32_t> &pt = pre_tab_;
        if (!build(pre, 19, 2, pt, 7)) return false;
        uint8_t lens[286 + 30 + 140];
        int i = 0;
        const int total = hlit + hdist;
        while (i < total) {
            ensure_input();
            refill();
            const uint32_t e = pt[bitbuf_ & 127];
            if ((e & 0xFF) == 0) return fail("deflate: bad code-length code");
            take(e & 0xFF);
            const int sym = (int)(e >> 16);
            if (sym < 16) lens[i++] = (uint8_t)sym;
            else {
                int rep;
                uint8_t v = 0;
                if (sym == 16) {
                    if (i == 0) return fail("deflate: repeat without a previous length");
                    v = lens[i - 1];
                    rep = 3 + (int)take(2);
                } else if (sym == 17) rep = 3 + (int)take(3);
                else rep = 11 + (int)take(7);
                if (i + rep > total) return fail("deflate: too many code lengths");
                memset(lens + i, v, (size_t)rep);
                i += rep;
            }
        }
        if (overran()) return fail("gz: input ends inside a block header");
        if (lens[256] == 0) return fail("deflate: no end-of-block code");
        if (!build(lens, hlit, 0, dyn_lit_, kLitBits)) return false;
        if (!build(lens + hlit, hdist, 1, dyn_dist_, kDistBits)) return false;
        lit_ = dyn_lit_.data();
        dist_ = dyn_dist_.data();
        return true;
    }

    // ---- the hot loop: symbols of one Huffman block until the chunk is full, the input runs low, or the block ends ----
    // returns 0 = block finished, 1 = come back (flush output / refill input), -1 = error
    int decode_block() {
        uint8_t *const base = out_.data();
        uint8_t *out = base + out_have_;
        uint8_t *const start = out;
        const size_t hist0 = (size_t)std::min<uint64_t>(member_out_, kWindow);
        uint8_t *const out_stop = base + kWindow + kOutChunk;           // a match may write up to 258 + 7 bytes beyond
        const uint32_t *const lit = lit_, *const dst = dist_;
        uint64_t bb = bitbuf_;
        unsigned bc = bitcnt_;
        size_t ip = in_pos_;
        const uint8_t *const in = in_.data();
        // an iteration loads 8 bytes at ip once (refill); ip advances <= 8 per iteration.  Before the end of the file the
        // loads must stay inside real data; at the end they may run into the zero padding (overran() then tells whether
        // padding was CONSUMED, i.e. the stream is truncated).
        const size_t lim = in_eof_ ? in_end_ + kPad - 16 : in_end_;
        const size_t in_stop = lim < 16 ? 0 : lim - 16;
        if (lim < 16) return 1;
        int rc = 1;
#define HAST_REFILL()                      \
    do {                                   \
        bb |= load64(in + ip) << bc;       \
        ip += (63 - bc) >> 3;              \
        bc |= 56;                          \
    } while (0)
        constexpr uint32_t LM = (1u << kLitBits) - 1, DM = (1u << kDistBits) - 1;
        if (out < out_stop && ip <= in_stop) {
            HAST_REFILL();
            uint32_t e = lit[bb & LM];                                   // the entry of the NEXT symbol is always looked up ahead
            for (;;) {
                if (e & kSub) {
                    bb >>= kLitBits;
                    bc -= kLitBits;
                    e = lit[(e >> 16) + (bb & ((1u << ((e >> 8) & 31)) - 1))];
                }
                bb >>= (e & 0xFF);
                bc -= (e & 0xFF);
                if (e & kLit) {
                    // a literal, and up to two more on the bits already loaded (<= 15 + 11 + 11 of >= 56)
                    *out++ = (uint8_t)(e >> 16);
                    e = lit[bb & LM];
                    if (e & kLit) {
                        bb >>= (e & 0xFF);
                        bc -= (e & 0xFF);
                        *out++ = (uint8_t)(e >> 16);
                        e = lit[bb & LM];
                        if (e & kLit) {
                            bb >>= (e & 0xFF);
                            bc -= (e & 0xFF);
                            *out++ = (uint8_t)(e >> 16);
                            e = lit[bb & LM];
                        }
                    }
                    // e was looked up with >= 19 real bits left, so it stays valid across the refill
                    if (!(out < out_stop && ip <= in_stop)) break;
                    HAST_REFILL();
                    continue;
                }
                if ((e & 0xFF) == 0) {
                    err_ = "deflate: invalid literal/length code";
                    rc = -1;
                    break;
                }
                if (e & kEob) {
                    rc = 0;
                    break;
                }
                // a match: <= 15 + 5 + 15 + 13 = 48 bits of the >= 56 loaded
                const unsigned leb = (e >> 8) & 31;
                const unsigned len = (e >> 16) + (unsigned)(bb & ((1u << leb) - 1));
                bb >>= leb;
                bc -= leb;
                uint32_t d = dst[bb & DM];
                if (d & kSub) {
                    bb >>= kDistBits;
                    bc -= kDistBits;
                    d = dst[(d >> 16) + (bb & ((1u << ((d >> 8) & 31)) - 1))];
                }
                if ((d & 0xFF) == 0) {
                    err_ = "deflate: invalid distance code";
                    rc = -1;
                    break;
                }
                bb >>= (d & 0xFF);
                bc -= (d & 0xFF);
                const unsigned deb = (d >> 8) & 31;
                const size_t distance = (d >> 16) + (size_t)(bb & ((1u << deb) - 1));
                bb >>= deb;
                bc -= deb;
                if (distance > hist0 + (size_t)(out - start)) {          // before the start of this member's data
                    err_ = "deflate: distance too far back";
                    rc = -1;
                    break;
                }
                const uint8_t *src = out - distance;
                uint8_t *const end = out + len;
                if (distance >= 8) {                                     // 8 bytes at a time; may write up to 7 bytes past `end`
                    memcpy(out, src, 8);
                    memcpy(out + 8, src + 8, 8);
                    if (len > 16) {
                        out += 16;
                        src += 16;
                        do {
                            memcpy(out, src, 8);
                            out += 8;
                            src += 8;
                        } while (out < end);
                    }
                } else if (distance == 1) {
                    memset(out, *src, len);
                } else {
                    do *out++ = *src++;
                    while (out < end);
                }
                out = end;
                if (!(out < out_stop && ip <= in_stop)) break;
                HAST_REFILL();
                e = lit[bb & LM];
            }
        }
#undef HAST_REFILL
        member_out_ += (uint64_t)(out - start);
        out_have_ = (size_t)(out - base);
        bitbuf_ = bb;
        bitcnt_ = bc;
        in_pos_ = ip;
        return rc;
    }

    // ---- one step of the state machine; returns false on error ---------------------------------------------------
    bool produce() {
        for (;;) {
            switch (state_) {
            case kMemberHeader: {
                align_to_byte();
                const int b0 = get_byte();
                if (b0 < 0) {                                             // clean end of input (an empty file is no data, as for gzread)
                    state_ = kDone;
                    return true;
                }
                const int b1 = get_byte();
                if (b0 != 0x1f || b1 != 0x8b) {
                    if (first_member_) {                                  // not gzip at all: hand the bytes through, as gzread does
                        out_[out_have_++] = (uint8_t)b0;
                        if (b1 >= 0) out_[out_have_++] = (uint8_t)b1;
                        state_ = kRaw;
                        break;
                    }
                    state_ = kDone;                                       // trailing garbage after the last member: ignored
                    return true;
                }
                if (get_byte() != 8) return fail("gz: unknown compression method");
                const int flg = get_byte();
                for (int i = 0; i < 6; ++i)
                    if (get_byte() < 0) return fail("gz: truncated header");
                if (flg & 4) {
                    const int lo = get_byte(), hi = get_byte();
                    if (hi < 0) return fail("gz: truncated header");
                    for (int i = 0, n = lo | (hi << 8); i < n; ++i)
                        if (get_byte() < 0) return fail("gz: truncated header");
                }
                for (int bit = 8; bit <= 16; bit <<= 1)
                    if (flg & bit) {
                        int c;
                        while ((c = get_byte()) > 0) {}
                        if (c < 0) return fail("gz: truncated header");
                    }
                if (flg & 2) {
                    get_byte();
                    if (get_byte() < 0) return fail("gz: truncated header");
                }
                first_member_ = false;
                crc_ = 0;
                isize_ = 0;
                member_out_ = 0;
                state_ = kBlockHeader;
                break;
            }
            case kBlockHeader: {
                ensure_input();
                refill();
                final_ = take(1) != 0;
                const uint32_t type = take(2);
                if (type == 0) {
                    align_to_byte();
                    int b[4];
                    for (int &x : b)
                        if ((x = get_byte()) < 0) return fail("gz: truncated stored block");
                    stored_left_ = (size_t)(b[0] | (b[1] << 8));
                    if ((stored_left_ ^ 0xFFFF) != (size_t)(b[2] | (b[3] << 8))) return fail("deflate: stored block length check failed");
                    state_ = kStored;
                } else if (type == 1) {
                    lit_ = fixed_lit_.data();
                    dist_ = fixed_dist_.data();
                    state_ = kHuffman;
                } else if (type == 2) {
                    if (!read_dynamic()) return false;
                    state_ = kHuffman;
                } else return fail("deflate: reserved block type");
                break;
            }
            case kStored: {
                while (stored_left_) {
                    if (out_have_ >= kWindow + kOutChunk) return account();
                    if (in_left() == 0) {
                        fill_input();
                        if (in_left() == 0) return fail("gz: truncated stored block");
                    }
                    const size_t n = std::min(std::min(stored_left_, in_left()), kWindow + kOutChunk - out_have_);
                    memcpy(out_.data() + out_have_, in_.data() + in_pos_, n);
                    in_pos_ += n;
                    out_have_ += n;
                    member_out_ += n;
                    stored_left_ -= n;
                }
                state_ = final_ ? kTrailer : kBlockHeader;
                if (out_have_ >= kWindow + kOutChunk) return account();
                break;
            }
            case kHuffman: {
                ensure_input();
                const int rc = decode_block();
                if (rc < 0) return false;
                if (overran()) return fail("gz: input ends inside a compressed block");
                if (rc == 0) state_ = final_ ? kTrailer : kBlockHeader;
                if (out_have_ >= kWindow + kOutChunk) return account();
                break;
            }
            case kTrailer: {
                if (!account()) return false;                              // CRC over what this member produced so far
                align_to_byte();
                uint32_t v[2] = {0, 0};
                for (int w = 0; w < 2; ++w)
                    for (int i = 0; i < 4; ++i) {
                        const int c = get_byte();
                        if (c < 0) return fail("gz: truncated trailer");
                        v[w] |= (uint32_t)c << (8 * i);
                    }
                if (v[0] != crc_) return fail("gz: CRC-32 mismatch");
                if (v[1] != (uint32_t)isize_) return fail("gz: length check (ISIZE) failed");
                state_ = kMemberHeader;
                if (out_have_ > out_pos_) return true;
                break;
            }
            case kRaw: {
                if (in_left() == 0) fill_input();
                const size_t n = std::min(in_left(), kWindow + kOutChunk - out_have_);
                memcpy(out_.data() + out_have_, in_.data() + in_pos_, n);
                in_pos_ += n;
                out_have_ += n;
                if (n == 0 && in_left() == 0) state_ = kDone;
                return true;
            }
            case kDone:
            case kError:
                return state_ == kDone;
            }
        }
    }
    // CRC-32 / size bookkeeping of the bytes produced since the last call
    bool account() {
        if (out_have_ > crc_done_) {
            crc_ = crc32_update(crc_, out_.data() + crc_done_, out_have_ - crc_done_);
            isize_ += out_have_ - crc_done_;
        }
        crc_done_ = out_have_;
        return true;
    }

    // ---- CRC-32 (gzip polynomial), slicing by 8 -------------------------------------------------------------------
    struct CrcTables {
        uint32_t t[8][256];
        bool clmul = false;
        CrcTables() {
            for (uint32_t i = 0; i < 256; ++i) {
                uint32_t c = i;
                for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
                t[0][i] = c;
            }
            for (uint32_t i = 0; i < 256; ++i)
                for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
        }
    };
    // built once (function-local static: thread-safe).  The folding version is used only where the CPU has the instruction
    // AND it reproduces the table version on a self-test.
    static const CrcTables &crc_state() {
        static const CrcTables tabs = [] {
            CrcTables c;
            if (__builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1")) {
                uint8_t probe[256 + 3];
                for (size_t i = 0; i < sizeof probe; ++i) probe[i] = (uint8_t)(i * 151 + (i >> 3));
                bool same = true;
                for (size_t n = 16; n <= 256; n += 16)
                    same = same && crc32_clmul(0x9d2c5680u + (uint32_t)n, probe + 3, n) == crc32_tables(c, 0x9d2c5680u + (uint32_t)n, probe + 3, n);
                c.clmul = same;
            }
            return c;
        }();
        return tabs;
    }
    // carry-less-multiply folding (16 bytes per step), constants = x^n mod P in the bit-reflected domain; n % 16 == 0, n >= 16
    __attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul(uint32_t crc, const uint8_t *p, size_t n) {
        const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0), k5 = _mm_set_epi64x(0, 0x0163cd6124);
        const __m128i poly = _mm_set_epi64x(0x01f7011641, 0x01db710641), mask32 = _mm_set_epi32(0, 0, 0, -1);
        __m128i x1 = _mm_xor_si128(_mm_loadu_si128(reinterpret_cast<const __m128i *>(p)), _mm_cvtsi32_si128((int)~crc));
        for (p += 16, n -= 16; n >= 16; p += 16, n -= 16) {
            const __m128i x2 = _mm_clmulepi64_si128(x1, k3k4, 0x00);
            x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11);
            x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), _mm_loadu_si128(reinterpret_cast<const __m128i *>(p)));
        }
        __m128i x2 = _mm_clmulepi64_si128(x1, k3k4, 0x10);                 // 128 -> 64 bits
        x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x2);
        x2 = _mm_srli_si128(x1, 4);                                        // 64 -> 32 bits
        x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5, 0x00), x2);
        x2 = _mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10);  // Barrett reduction
        x2 = _mm_clmulepi64_si128(_mm_and_si128(x2, mask32), poly, 0x00);
        return ~(uint32_t)_mm_extract_epi32(_mm_xor_si128(x1, x2), 1);
    }
    static uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {
        const CrcTables &c = crc_state();
        if (c.clmul && n >= 64) {
            const size_t m = n & ~(size_t)15;
            crc = crc32_clmul(crc, p, m);
            p += m;
            n -= m;
        }
        return crc32_tables(c, crc, p, n);
    }
    static uint32_t crc32_tables(const CrcTables &tabs, uint32_t crc, const uint8_t *p, size_t n) {      // slicing by 8
        const auto &t = tabs.t;
        uint32_t c = ~crc;
        while (n && ((uintptr_t)p & 7)) {
            c = t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
            --n;
        }
        while (n >= 8) {
            uint64_t v;
            memcpy(&v, p, 8);
            v ^= c;
            c = t[7][v & 0xFF] ^ t[6][(v >> 8) & 0xFF] ^ t[5][(v >> 16) & 0xFF] ^ t[4][(v >> 24) & 0xFF] ^ t[3][(v >> 32) & 0xFF] ^
                t[2][(v >> 40) & 0xFF] ^ t[1][(v >> 48) & 0xFF] ^ t[0][v >> 56];
            p += 8;
            n -= 8;
        }
        while (n--) c = t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
        return ~c;
    }

    FILE *fp_ = nullptr;
    std::vector<uint8_t> in_, out_;
    size_t in_pos_ = 0, in_end_ = 0;
    bool in_eof_ = false;
    size_t out_pos_ = 0, out_have_ = 0, crc_done_ = 0;
    uint64_t member_out_ = 0;                              // bytes this member has produced so far
    State state_ = kDone;
    uint64_t bitbuf_ = 0;
    unsigned bitcnt_ = 0;
    bool final_ = false, first_member_ = true;
    size_t stored_left_ = 0;
    uint32_t crc_ = 0;
    uint64_t isize_ = 0;
    std::vector<uint32_t> fixed_lit_, fixed_dist_, dyn_lit_, dyn_dist_, pre_tab_;
    const uint32_t *lit_ = nullptr, *dist_ = nullptr;
    std::string err_;
};

}  // namespace hast

// par_inflate.h -- an ORDINARY gzip file (one long deflate stream, as `gzip` writes the .fq.gz inputs of HAST) inflated by
// several threads.
//
// Why: the reference reads its .gz inputs through zlib on one thread (gzstream.h:47, classify.cpp:257-273); with the GPU side
// at > 200 Gbp/s and the serial decoder of fast_inflate.h at ~1 GB/s per file, an ordinary .gz input IS the run time of the
// drop-in CLI.  A deflate stream cannot simply be cut into pieces: block boundaries are bit positions nobody wrote down,
// and a piece's first 32 KB may copy from the 32 KB in front of it.  The published way around both (pugz: Kerbiriou & Chikhi
// 2019; rapidgzip: Knespel & Brunst 2023) is what this file implements, in its own terms:
//
//   1. the compressed bytes are cut into chunks; every chunk but the first SEARCHES its first block boundary: the bit
//      positions from its nominal start on are tried as the header of a dynamic-Huffman block (type bits, code counts, a
//      COMPLETE code-length code, complete literal/length and distance codes as zlib demands -- about one position in 10^7
//      survives by chance);
//   2. from there the chunk is decoded with its window UNKNOWN: the output is 16-bit symbols, a literal byte or a marker
//      "byte i of the 32 KB in front of this chunk" (the symbol buffer starts with the 32768 markers, so a copy from the
//      unknown window is an ordinary copy); it stops at the first block boundary at or behind the next chunk's nominal start;
//   3. in file order (serial, cheap): a chunk is ACCEPTED iff the chunk in front of it, decoded from a boundary already
//      proven, ended exactly at the bit position the chunk started from -- so by induction from the stream's first block
//      every accepted chunk starts at a true boundary and a false positive of step 1 can only cost time: the chunk in front
//      then simply decodes on through it.  The 32-KB window behind each accepted chunk is resolved chunk after chunk;
//   4. markers -> bytes through a 64-K look-up table per chunk (all chunks at once, straight into the caller's buffer), and
//      every member's CRC-32 and ISIZE are checked (per-piece CRCs combined with crc32_combine), so a bug cannot pass as data.
//
// The first chunk of every batch starts from a known position with a known window (the end of the previous batch) and sees
// no markers at all.  The next batch is decoded by a background thread while the caller drains the current one.
// Anything this reader cannot start on (not a regular file, no gzip magic) is left to GzInflater (fast_inflate.h).
//
// Usage: ParGzReader z; if (ParGzReader::usable(f)) { z.open(f, threads); long n = z.read(buf, cap); ... }   (n as GzInflater)
#pragma once
#include <emmintrin.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fast_inflate.h"
#include "worker_pool.h"

namespace hast {

class ParGzReader {
  public:
    ~ParGzReader() { close(); }
    // a regular file positioned at the two magic bytes of a gzip member; the position is left where it was
    static bool usable(FILE *f) {
        struct stat st;
        const int fd = fileno(f);
        if (fd < 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return false;
        const off_t at = ftello(f);
        unsigned char h[2];
        return at >= 0 && pread(fd, h, 2, at) == 2 && h[0] == 0x1f && h[1] == 0x8b;
    }
    // chunk_bytes: compressed bytes per chunk (tests use tiny ones); a batch is 2 chunks per thread
    // max_chunk_out: a chunk stops at the first block boundary behind this many bytes of output (highly compressible data:
    // 1 MB of deflate can be 1 GB of zeros, and every chunk of a batch holds its output as 16-bit symbols)
    void open(FILE *f, int threads, size_t chunk_bytes = 1u << 20, size_t max_chunk_out = 48u << 20) {
        close();
        max_chunk_out_ = std::max<size_t>(max_chunk_out, 1);
        fd_ = fileno(f);
        file_pos_ = (uint64_t)ftello(f);
        threads_ = std::max(1, threads);
        chunk_bytes_ = std::max<size_t>(chunk_bytes, 64);
        pool_.reset(new WorkerPool(threads_));
        rpool_.reset(new WorkerPool(std::max(1, threads_ / 2)));
        err_.clear();
        cur_.reset();
        cur_chunk_ = 0;
        crc_ = 0;
        isize_ = 0;
        done_ = false;
        stop_ = false;
        ready_.reset();
        // stream state the producer carries from batch to batch
        next_bit_ = 0;
        in_base_ = 0;
        in_.assign(kPad, 0);
        in_len_ = 0;
        in_eof_ = false;
        carry_window_.assign(kWindow, 0);
        carry_member_out_ = 0;
        stream_done_ = false;
        at_member_header_ = true;
        bg_ = std::thread([this] { produce(); });
    }
    void close() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (bg_.joinable()) bg_.join();
        pool_.reset();
        rpool_.reset();
        cur_.reset();
        ready_.reset();
    }
    // up to cap bytes of decompressed data; 0 = end of input; < 0 = error (error())
    long read(uint8_t *dst, size_t cap) {
        size_t got = 0;
        while (got < cap) {
            if (!cur_ || cur_chunk_ >= cur_->chunks.size()) {
                if (done_) break;
                if (cur_ && cur_->last) {
                    done_ = true;
                    break;
                }
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return ready_ || stop_; });
                if (!ready_) {
                    if (err_.empty()) err_ = "gz: decoder stopped";
                    return -1;
                }
                cur_ = std::move(ready_);
                cur_chunk_ = 0;
                g.unlock();
                cv_.notify_all();
                if (!cur_->error.empty() && cur_->chunks.empty()) {
                    err_ = cur_->error;
                    return -1;
                }
                continue;
            }
            // pieces of the accepted chunks, in order, that fit the caller's buffer
            struct Piece { Chunk *c; size_t from, n, at; uint32_t crc; };
            std::vector<Piece> pieces;
            size_t room = cap - got, at = got;
            size_t k = cur_chunk_;
            while (room && k < cur_->chunks.size()) {
                Chunk &c = *cur_->chunks[k];
                size_t left = c.n - c.consumed;
                if (c.n == 0) pieces.push_back({&c, 0, 0, at, 0});     // no output, but members may end here (empty members)
                size_t from = c.consumed;
                while (left && room) {
                    size_t n = std::min(std::min(left, room), kPiece);
                    // a piece never crosses a member's end: the CRC bookkeeping below works member by member
                    for (const Event &e : c.ev)
                        if (e.at > from && e.at < from + n) { n = e.at - from; break; }
                    pieces.push_back({&c, from, n, at, 0});
                    from += n; at += n; left -= n; room -= n;
                }
                c.consumed = from;
                if (left == 0) ++k;
            }
            std::atomic<size_t> next{0};
            rpool_->run([&](int) {
                for (size_t i; (i = next.fetch_add(1)) < pieces.size();) {
                    Piece &p = pieces[i];
                    if (!p.n) continue;
                    resolve(*p.c, p.from, p.n, dst + p.at);
                    p.crc = GzInflater::crc32(0, dst + p.at, p.n);
                }
            });
            // CRC-32 / ISIZE of every member that ends inside what was just delivered (member ends sit at piece borders)
            for (Piece &p : pieces) {
                if (!check_events(*p.c, p.from)) return -1;
                if (p.n) {
                    crc_ = isize_ ? (uint32_t)crc32_combine(crc_, p.crc, (z_off_t)p.n) : p.crc;
                    isize_ += p.n;
                }
                if (p.from + p.n == p.c->n && !check_events(*p.c, p.c->n)) return -1;
            }
            for (; cur_chunk_ < k; ++cur_chunk_) give_chunk(std::move(cur_->chunks[cur_chunk_]));
            got = at;
            if (cur_chunk_ >= cur_->chunks.size() && !cur_->error.empty()) {     // what could be decoded has been delivered
                err_ = cur_->error;
                return -1;
            }
        }
        return (long)got;
    }
    const std::string &error() const { return err_; }

  private:
    static constexpr size_t kWindow = 32768, kPad = 64, kPiece = 1u << 20, kOutSlack = 320;
    static constexpr uint16_t kMarker = 0x8000;
    struct Event { size_t at; uint32_t crc, isize; bool checked; };      // a member ends in front of output symbol `at`
    struct Chunk {
        uint64_t search_from = 0, stop_bit = 0;      // absolute bit positions: where the search starts / first boundary >= ends the chunk
        bool known = false, found = false, eos = false, starved = false;
        uint64_t start_bit = 0, end_bit = 0;
        std::string err;                             // decode error (of a proven chunk: the stream is damaged)
        std::vector<uint16_t> sym;                   // kWindow symbols of history (markers, or the real window), then the output
        size_t n = 0, consumed = 0;
        std::vector<Event> ev;
        size_t hist_lo = 0;                          // lowest index of sym a copy may read (member starts are history barriers)
        bool at_member_header = false;               // the chunk ends in front of a gzip header that was not in the buffer yet
        std::vector<uint8_t> lut;                    // marker -> byte (built when the chunk in front has been resolved)
    };
    struct Batch {
        std::vector<std::unique_ptr<Chunk>> chunks;  // accepted chunks, in stream order
        std::string error;                           // the stream is damaged behind the last chunk
        bool last = false;                           // the stream ends with this batch
    };

    // ---- bit input over the batch buffer -------------------------------------------------------------------------
    struct Bits {
        const uint8_t *p;        // buffer (kPad zero bytes behind `len`)
        size_t len;              // real bytes
        uint64_t bb = 0;
        unsigned bc = 0;
        size_t ip = 0;
        static uint64_t load64(const uint8_t *q) { uint64_t v; memcpy(&v, q, 8); return v; }
        void seek(uint64_t bit) {
            ip = (size_t)(bit >> 3);
            bb = 0;
            bc = 0;
            refill();
            bb >>= (bit & 7);
            bc -= (unsigned)(bit & 7);
        }
        void refill() {
            bb |= load64(p + ip) << bc;
            ip += (63 - bc) >> 3;
            bc |= 56;
        }
        uint32_t take(unsigned n) {
            const uint32_t v = (uint32_t)(bb & ((1ull << n) - 1));
            bb >>= n;
            bc -= n;
            return v;
        }
        uint64_t pos() const { return (uint64_t)ip * 8 - bc; }
        bool overran() const { return pos() > (uint64_t)len * 8; }
        bool low() const { return ip + 16 > len; }            // fewer than 16 real bytes ahead: decode carefully / stop
    };
    struct Tables {
        std::vector<uint32_t> lit, dist, pre;
    };

    // ---- dynamic block header: code lengths -> tables.  strict = zlib's rules for complete codes (the search uses them) ----
    static const char *read_dynamic(Bits &in, Tables &t, bool strict) {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        in.refill();
        const int hlit = (int)in.take(5) + 257, hdist = (int)in.take(5) + 1, hclen = (int)in.take(4) + 4;
        if (hlit > 286 || hdist > 30) return "deflate: bad code counts";
        uint8_t pre[19] = {0};
        for (int i = 0; i < hclen; ++i) {
            if (in.bc < 3) in.refill();
            pre[order[i]] = (uint8_t)in.take(3);
        }
        if (strict && !complete(pre, 19, false)) return "deflate: incomplete code-length code";
        if (const char *bad = inflate_build_table(pre, 19, 2, t.pre, 7)) return bad;
        uint8_t lens[286 + 30 + 140];
        int i = 0;
        const int total = hlit + hdist;
        while (i < total) {
            if (in.overran()) return "gz: input ends inside a block header";
            in.refill();
            const uint32_t e = t.pre[in.bb & 127];
            if ((e & 0xFF) == 0) return "deflate: bad code-length code";
            in.take(e & 0xFF);
            const int sym = (int)(e >> 16);
            if (sym < 16) lens[i++] = (uint8_t)sym;
            else {
                int rep;
                uint8_t v = 0;
                if (sym == 16) {
                    if (i == 0) return "deflate: repeat without a previous length";
                    v = lens[i - 1];
                    rep = 3 + (int)in.take(2);
                } else if (sym == 17) rep = 3 + (int)in.take(3);
                else rep = 11 + (int)in.take(7);
                if (i + rep > total) return "deflate: too many code lengths";
                memset(lens + i, v, (size_t)rep);
                i += rep;
            }
        }
        if (in.overran()) return "gz: input ends inside a block header";
        if (lens[256] == 0) return "deflate: no end-of-block code";
        if (strict && (!complete(lens, hlit, true) || !complete(lens + hlit, hdist, true))) return "deflate: incomplete code";
        if (const char *bad = inflate_build_table(lens, hlit, 0, t.lit, kInflateLitBits)) return bad;
        if (const char *bad = inflate_build_table(lens + hlit, hdist, 1, t.dist, kInflateDistBits)) return bad;
        return nullptr;
    }
    // zlib's inflate_table: a code must be complete, except (lone_ok) a single code of length 1, or no code at all
    static bool complete(const uint8_t *lens, int n, bool lone_ok) {
        int count[16] = {0}, used = 0, maxl = 0;
        for (int i = 0; i < n; ++i)
            if (lens[i]) {
                count[lens[i]]++;
                ++used;
                maxl = std::max<int>(maxl, lens[i]);
            }
        if (used == 0) return lone_ok;
        int left = 1;
        for (int l = 1; l <= 15; ++l) {
            left = (left << 1) - count[l];
            if (left < 0) return false;
        }
        return left == 0 || (lone_ok && maxl == 1);
    }
    static void fixed_tables(Tables &t) {
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        inflate_build_table(l, 288, 0, t.lit, kInflateLitBits);
        uint8_t d[30];
        for (int i = 0; i < 30; ++i) d[i] = 5;
        inflate_build_table(d, 30, 1, t.dist, kInflateDistBits);
    }

    // ---- step 1: the first bit position in [from, to) that parses as the header of a non-final dynamic block -------------
    static bool find_block(const uint8_t *p, size_t len, uint64_t from, uint64_t to, uint64_t &found, Tables &t) {
        static const uint8_t kraft[8] = {0, 64, 32, 16, 8, 4, 2, 1};
        const uint64_t last = len >= 16 ? (uint64_t)(len - 16) * 8 : 0;
        to = std::min(to, last);
        for (uint64_t bit = from; bit < to; ++bit) {
            const uint64_t v = Bits::load64(p + (bit >> 3)) >> (bit & 7);           // >= 57 bits
            if ((v & 7) != 4) continue;                                             // BFINAL = 0, BTYPE = 10
            if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) continue;             // HLIT, HDIST
            const unsigned hclen = (unsigned)((v >> 13) & 15) + 4;
            // the code-length code must be complete: sum of 2^(7-len) == 128
            uint64_t w = v >> 17;                                                   // 40 bits = 13 lengths
            unsigned sum = 0, i = 0;
            for (; i < hclen && i < 13; ++i, w >>= 3) sum += kraft[w & 7];
            if (i < hclen) {
                const uint64_t bit2 = bit + 17 + 39;
                w = Bits::load64(p + (bit2 >> 3)) >> (bit2 & 7);
                for (; i < hclen; ++i, w >>= 3) sum += kraft[w & 7];
            }
            if (sum != 128) continue;
            Bits in{p, len};
            in.seek(bit + 3);
            if (read_dynamic(in, t, true)) continue;
            found = bit;
            return true;
        }
        return false;
    }

    // ---- step 2: symbols of one Huffman block.  0 = block finished, 1 = more room needed, 2 = input ran out, -1 = error ----
    static int decode_huffman(Bits &in, const Tables &t, uint16_t *base, size_t &n_out, size_t cap, size_t hist_lo, const char *&err) {
        const uint32_t *const lit = t.lit.data(), *const dst = t.dist.data();
        uint16_t *out = base + n_out;
        uint16_t *const out_stop = base + cap - kOutSlack;
        uint16_t *const lo = base + hist_lo;
        constexpr uint32_t LM = (1u << kInflateLitBits) - 1, DM = (1u << kInflateDistBits) - 1;
        constexpr uint32_t kLit = kInflateLit, kEob = kInflateEob, kSub = kInflateSub;
        uint64_t bb = in.bb;
        unsigned bc = in.bc;
        size_t ip = in.ip;
        const uint8_t *const p = in.p;
        const size_t in_stop = in.len + kPad - 16;                     // loads stay inside the padded buffer
        int rc = 1;
#define HAST_PREFILL()                        \
    do {                                      \
        bb |= Bits::load64(p + ip) << bc;     \
        ip += (63 - bc) >> 3;                 \
        bc |= 56;                             \
    } while (0)
        for (;;) {
            if (out >= out_stop) { rc = 1; break; }
            if (ip > in_stop) { rc = 2; break; }
            HAST_PREFILL();
            uint32_t e = lit[bb & LM];
            if (e & kSub) {
                bb >>= kInflateLitBits;
                bc -= kInflateLitBits;
                e = lit[(e >> 16) + (bb & ((1u << ((e >> 8) & 31)) - 1))];
            }
            bb >>= (e & 0xFF);
            bc -= (e & 0xFF);
            if (e & kLit) {
                *out++ = (uint16_t)(e >> 16);
                // up to two more literals on the bits already loaded (<= 15 + 11 + 11 of >= 56)
                e = lit[bb & LM];
                if (e & kLit) {
                    bb >>= (e & 0xFF);
                    bc -= (e & 0xFF);
                    *out++ = (uint16_t)(e >> 16);
                    e = lit[bb & LM];
                    if (e & kLit) {
                        bb >>= (e & 0xFF);
                        bc -= (e & 0xFF);
                        *out++ = (uint16_t)(e >> 16);
                    }
                }
                continue;
            }
            if ((e & 0xFF) == 0) { err = "deflate: invalid literal/length code"; rc = -1; break; }
            if (e & kEob) { rc = 0; break; }
            const unsigned leb = (e >> 8) & 31;
            const unsigned len = (e >> 16) + (unsigned)(bb & ((1u << leb) - 1));
            bb >>= leb;
            bc -= leb;
            uint32_t d = dst[bb & DM];
            if (d & kSub) {
                bb >>= kInflateDistBits;
                bc -= kInflateDistBits;
                d = dst[(d >> 16) + (bb & ((1u << ((d >> 8) & 31)) - 1))];
            }
            if ((d & 0xFF) == 0) { err = "deflate: invalid distance code"; rc = -1; break; }
            bb >>= (d & 0xFF);
            bc -= (d & 0xFF);
            const unsigned deb = (d >> 8) & 31;
            const size_t distance = (d >> 16) + (size_t)(bb & ((1u << deb) - 1));
            bb >>= deb;
            bc -= deb;
            if ((size_t)(out - lo) < distance) { err = "deflate: distance too far back"; rc = -1; break; }
            const uint16_t *src = out - distance;
            uint16_t *const end = out + len;
            if (distance >= 4) {                                       // 4 symbols at a time; may write up to 3 symbols past `end`
                do {
                    memcpy(out, src, 8);
                    out += 4;
                    src += 4;
                } while (out < end);
            } else {
                do *out++ = *src++;
                while (out < end);
            }
            out = end;
        }
#undef HAST_PREFILL
        n_out = (size_t)(out - base);
        in.bb = bb;
        in.bc = bc;
        in.ip = ip;
        return rc;
    }

    // ---- blocks from the bit position `c.end_bit` (a block boundary, or a gzip header when c.at_member_header) until the
    // first boundary >= stop_bit, the end of the stream, or the end of the buffer.  Everything is committed boundary by
    // boundary: on starvation / error the chunk keeps its last complete boundary.  false = decode error (c.err). ----------
    bool decode_blocks(Chunk &c, uint64_t stop_bit, Tables &t) const {
        const uint8_t *const p = in_.data();
        const size_t len = in_len_;
        auto rel = [&](uint64_t abs_bit) { return abs_bit - in_base_ * 8; };
        auto ensure = [&](size_t extra) {
            if (c.sym.size() < kWindow + c.n + extra + kOutSlack) c.sym.resize(std::max(c.sym.size() * 2, kWindow + c.n + extra + kOutSlack));
        };
        c.starved = false;
        for (;;) {
            if (c.eos) return true;
            if (c.end_bit >= stop_bit) return true;
            if (c.n >= max_chunk_out_) {                                // enough output for one chunk: the batch ends here
                c.starved = true;
                return true;
            }
            Bits in{p, len};
            // ---- a gzip member header (RFC 1952) stands here --------------------------------------------------------
            if (c.at_member_header) {
                size_t at = (size_t)(rel(c.end_bit) >> 3);             // byte aligned by construction
                if (at >= len) {
                    if (in_eof_) { c.eos = true; return true; }        // clean end of the file
                    c.starved = true;
                    return true;
                }
                if (len - at < 2 || p[at] != 0x1f || p[at + 1] != 0x8b) {
                    if (len - at < 2 && !in_eof_) { c.starved = true; return true; }
                    c.eos = true;                                      // bytes that do not start a member: ignored, as gzread does
                    return true;
                }
                size_t q = at + 2;
                auto need = [&](size_t nbytes) { return q + nbytes <= len; };
                bool short_in = false;
                const char *bad = nullptr;
                do {
                    if (!need(8)) { short_in = true; break; }
                    if (p[q] != 8) { bad = "gz: unknown compression method"; break; }
                    const int flg = p[q + 1];
                    q += 8;
                    if (flg & 4) {
                        if (!need(2)) { short_in = true; break; }
                        const size_t xlen = p[q] | ((size_t)p[q + 1] << 8);
                        q += 2;
                        if (!need(xlen)) { short_in = true; break; }
                        q += xlen;
                    }
                    for (int bit = 8; bit <= 16 && !short_in; bit <<= 1)
                        if (flg & bit) {
                            while (q < len && p[q]) ++q;
                            if (q >= len) { short_in = true; break; }
                            ++q;
                        }
                    if (short_in) break;
                    if (flg & 2) {
                        if (!need(2)) { short_in = true; break; }
                        q += 2;
                    }
                } while (false);
                if (bad) { c.err = bad; return false; }
                if (short_in) {
                    if (in_eof_) { c.err = "gz: truncated header"; return false; }
                    c.starved = true;
                    return true;
                }
                c.end_bit = (in_base_ + q) * 8;
                c.at_member_header = false;
                c.hist_lo = kWindow + c.n;                             // nothing in front of a member's first byte may be copied
                continue;
            }
            // ---- one deflate block --------------------------------------------------------------------------------
            if (!in_eof_ && (size_t)(rel(c.end_bit) >> 3) + kMinAhead > len) { c.starved = true; return true; }
            in.seek(rel(c.end_bit));
            const size_t n0 = c.n;
            const bool final = in.take(1) != 0;
            const uint32_t type = in.take(2);
            const char *bad = nullptr;
            bool starved = false;
            if (type == 0) {
                size_t at = (size_t)((in.pos() + 7) >> 3);
                if (at + 4 > len) starved = true;
                else {
                    const size_t n = p[at] | ((size_t)p[at + 1] << 8), nn = p[at + 2] | ((size_t)p[at + 3] << 8);
                    if ((n ^ 0xFFFF) != nn) bad = "deflate: stored block length check failed";
                    else if (at + 4 + n > len) starved = true;
                    else {
                        ensure(n);
                        uint16_t *o = c.sym.data() + kWindow + c.n;
                        for (size_t i = 0; i < n; ++i) o[i] = p[at + 4 + i];
                        c.n += n;
                        in.seek((uint64_t)(at + 4 + n) * 8);
                    }
                }
            } else if (type == 3) bad = "deflate: reserved block type";
            else {
                if (type == 1) fixed_tables(t);
                else bad = read_dynamic(in, t, false);
                while (!bad && !starved) {
                    ensure(1u << 16);
                    size_t n_abs = kWindow + c.n;
                    const int rc = decode_huffman(in, t, c.sym.data(), n_abs, c.sym.size(), c.hist_lo, bad);
                    c.n = n_abs - kWindow;
                    if (rc == 0) break;
                    if (rc == 2) starved = true;
                }
                if (!bad && !starved && in.overran()) starved = true;
            }
            // an "error" found while reading the zero padding behind the buffer's last real byte is no error: the block
            // simply is not all here yet
            if (bad && in.overran()) { bad = nullptr; starved = true; }
            if (!bad && starved && in_eof_) bad = "gz: input ends inside a compressed block";
            if (bad || starved) {
                c.n = n0;                                              // back to the last boundary
                if (bad) { c.err = bad; return false; }
                c.starved = true;
                return true;
            }
            if (final) {
                const size_t at = (size_t)((in.pos() + 7) >> 3);
                if (at + 8 > len) {
                    c.n = n0;
                    if (in_eof_) { c.err = "gz: truncated trailer"; return false; }
                    c.starved = true;
                    return true;
                }
                Event e;
                e.at = c.n;
                e.crc = p[at] | (p[at + 1] << 8) | (p[at + 2] << 16) | ((uint32_t)p[at + 3] << 24);
                e.isize = p[at + 4] | (p[at + 5] << 8) | (p[at + 6] << 16) | ((uint32_t)p[at + 7] << 24);
                e.checked = false;
                c.ev.push_back(e);
                c.end_bit = (in_base_ + at + 8) * 8;
                c.at_member_header = true;
            } else c.end_bit = in_base_ * 8 + in.pos();
        }
    }
    static constexpr size_t kMinAhead = 32;      // a block is only started with this many bytes in the buffer (or at the file's end)

    // ---- the producer: batch after batch -----------------------------------------------------------------------------
    void produce() {
        for (;;) {
            std::unique_ptr<Batch> b(new Batch);
            make_batch(*b);
            const bool last = b->last || !b->error.empty();
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return !ready_ || stop_; });
                if (stop_) return;
                ready_ = std::move(b);
            }
            cv_.notify_all();
            if (last) return;
        }
    }
    // drop what lies in front of next_bit_, then read until want_bytes are in the buffer or the file ends
    void fill_input(size_t want_bytes) {
        const uint64_t keep_from = next_bit_ >> 3;
        if (keep_from > in_base_) {
            const size_t drop = (size_t)std::min<uint64_t>(keep_from - in_base_, in_len_);
            memmove(in_.data(), in_.data() + drop, in_len_ - drop);
            in_len_ -= drop;
            in_base_ += drop;
        }
        if (in_.size() < want_bytes + kPad) in_.resize(want_bytes + kPad);
        if (!in_eof_ && in_len_ < want_bytes) {
            // the decode pool is idle between two batches: its threads read the new bytes side by side (one thread copies
            // ~3 GB/s out of the page cache, and this read is serial time of every batch)
            const size_t need = want_bytes - in_len_, nthr = (size_t)pool_->size();
            const size_t share = ((need + nthr - 1) / nthr + 4095) & ~(size_t)4095;
            std::vector<size_t> got(nthr, 0);
            std::vector<int> io_errno(nthr, 0);                         // a failed read is an error, not the end of the file
            pool_->run([&](int t) {
                const size_t from = std::min(need, share * (size_t)t), to = std::min(need, from + share);
                size_t done = 0;
                while (from + done < to) {
                    const ssize_t r = pread(fd_, in_.data() + in_len_ + from + done, to - from - done, (off_t)(file_pos_ + in_base_ + in_len_ + from + done));
                    if (r < 0 && errno == EINTR) continue;
                    if (r < 0) io_errno[(size_t)t] = errno ? errno : EIO;
                    if (r <= 0) break;
                    done += (size_t)r;
                }
                got[(size_t)t] = done;
            });
            for (size_t t = 0; t < nthr; ++t)
                if (io_errno[t] && io_error_.empty()) io_error_ = std::string("gz: read failed: ") + strerror(io_errno[t]);
            size_t total = 0;                                           // contiguous bytes from the start: a short share is the file's end
            for (size_t t = 0; t < nthr; ++t) {
                const size_t from = std::min(need, share * t), to = std::min(need, from + share);
                total += got[t];
                if (got[t] < to - from) { in_eof_ = true; break; }
            }
            in_len_ += total;
        }
        memset(in_.data() + in_len_, 0, kPad);
    }
    // symbol buffers are recycled: a fresh 10-MB vector per chunk and batch would spend its time in page faults
    std::unique_ptr<Chunk> take_chunk() {
        std::unique_ptr<Chunk> c;
        {
            std::lock_guard<std::mutex> g(spare_mu_);
            if (!spare_.empty()) {
                c = std::move(spare_.back());
                spare_.pop_back();
            }
        }
        if (!c) c.reset(new Chunk);
        std::vector<uint16_t> sym = std::move(c->sym);
        std::vector<uint8_t> lut = std::move(c->lut);
        *c = Chunk();
        c->sym = std::move(sym);
        c->lut = std::move(lut);
        c->ev.clear();
        return c;
    }
    void give_chunk(std::unique_ptr<Chunk> c) {
        if (!c) return;
        std::lock_guard<std::mutex> g(spare_mu_);
        if (spare_.size() < (size_t)threads_ * 4) spare_.push_back(std::move(c));
    }
    void make_batch(Batch &b) {
        if (stream_done_) { b.last = true; return; }
        const size_t nchunks = (size_t)threads_ * 2;
        // in_base_ counts from the first byte of the stream (file_pos_)
        const uint64_t first_byte = next_bit_ >> 3;
        fill_input((nchunks + 1) * chunk_bytes_);
        if (!io_error_.empty()) {                    // an I/O error is not the end of the stream
            b.error = io_error_;
            b.last = true;
            stream_done_ = true;
            return;
        }
        std::vector<std::unique_ptr<Chunk>> ch(nchunks);
        for (size_t i = 0; i < nchunks; ++i) {
            ch[i] = take_chunk();
            Chunk &c = *ch[i];
            c.search_from = (first_byte + i * chunk_bytes_) * 8;
            c.stop_bit = (first_byte + (i + 1) * chunk_bytes_) * 8;
        }
        {   // chunk 0: a proven position, the real window
            Chunk &c = *ch[0];
            c.known = c.found = true;
            c.start_bit = c.end_bit = next_bit_;
            c.at_member_header = at_member_header_;
            if (c.sym.size() < kWindow + chunk_bytes_ * 5 + kOutSlack) c.sym.resize(kWindow + chunk_bytes_ * 5 + kOutSlack);
            for (size_t i = 0; i < kWindow; ++i) c.sym[i] = carry_window_[i];
            c.hist_lo = kWindow - (size_t)std::min<uint64_t>(carry_member_out_, kWindow);
        }
        const uint64_t buf_end_bit = (in_base_ + in_len_) * 8;
        const bool trace = getenv("HAST_GZ_TRACE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        std::atomic<size_t> next{0};
        pool_->run([&](int) {
            Tables t;
            for (size_t i; (i = next.fetch_add(1)) < nchunks;) {
                Chunk &c = *ch[i];
                if (i == 0) {
                    decode_blocks(c, c.stop_bit, t);
                    continue;
                }
                if (c.search_from >= buf_end_bit) continue;
                // search, decode; a candidate whose blocks do not decode was no boundary: search on behind it
                uint64_t from = c.search_from;
                for (int tries = 0; tries < 64; ++tries) {
                    uint64_t at;
                    if (!find_block(in_.data(), in_len_, from - in_base_ * 8, c.stop_bit - in_base_ * 8, at, t)) break;
                    at += in_base_ * 8;
                    c.found = true;
                    c.start_bit = c.end_bit = at;
                    c.n = 0;
                    c.ev.clear();
                    c.err.clear();
                    c.eos = c.at_member_header = false;
                    c.hist_lo = 0;
                    if (c.sym.size() < kWindow + chunk_bytes_ * 5) c.sym.resize(kWindow + chunk_bytes_ * 5 + kOutSlack);
                    for (size_t k = 0; k < kWindow; ++k) c.sym[k] = (uint16_t)(kMarker + k);
                    const bool ok = decode_blocks(c, c.stop_bit, t);
                    if (ok && (c.end_bit > c.start_bit || c.eos)) break;
                    c.found = false;
                    if (ok) break;                                      // starved in its first block: as good as nothing found
                    from = at + 1;
                }
            }
        });
        // ---- step 3: accept in order ------------------------------------------------------------------------------------
        const auto t1 = std::chrono::steady_clock::now();
        Tables t;
        size_t cur = 0;
        std::vector<size_t> accepted{0};
        bool failed = !ch[0]->err.empty();
        for (size_t nx = 1; nx < nchunks && !failed; ++nx) {
            Chunk &a = *ch[cur];
            Chunk &c = *ch[nx];
            if (a.eos || a.starved) break;
            if (c.found && a.end_bit < c.start_bit) {                   // e.g. the boundary behind a's stop was a stored or fixed block
                if (!decode_blocks(a, c.start_bit, t)) { failed = true; break; }
                if (a.eos || a.starved) break;
            }
            if (c.found && a.end_bit == c.start_bit && !a.at_member_header) {
                accepted.push_back(nx);
                cur = nx;
                continue;
            }
            // c did not start at a boundary (or found nothing): the chunk in front decodes on through it
            if (!decode_blocks(a, c.stop_bit, t)) { failed = true; break; }
        }
        Chunk &tail = *ch[cur];
        if (failed) b.error = ch[cur]->err.empty() ? "gz: damaged input" : ch[cur]->err;
        if (trace && failed) fprintf(stderr, "[pargz] failed in chunk %zu: known %d start %llu end %llu n %zu hist_lo %zu ev %zu stop %llu next_bit %llu carry_out %llu\n", cur, (int)ch[cur]->known,
                                     (unsigned long long)ch[cur]->start_bit, (unsigned long long)ch[cur]->end_bit, ch[cur]->n, ch[cur]->hist_lo, ch[cur]->ev.size(), (unsigned long long)ch[cur]->stop_bit,
                                     (unsigned long long)next_bit_, (unsigned long long)carry_member_out_);
        // a batch that made no progress with the whole buffer in hand: one block is larger than the read-ahead
        if (!failed && accepted.size() == 1 && tail.end_bit == next_bit_ && tail.n == 0 && tail.ev.empty() && !tail.eos) {
            if (in_eof_) b.error = "gz: input ends inside a compressed block";
            else {
                chunk_bytes_ *= 2;                                      // try again with twice the read-ahead
                make_batch(b);
                return;
            }
        }
        // ---- windows and look-up tables, chunk after chunk (32 KB each) ----------------------------------------------------
        std::vector<uint8_t> window = carry_window_;
        uint64_t member_out = carry_member_out_;
        for (size_t idx : accepted) {
            Chunk &c = *ch[idx];
            c.lut.resize(65536);
            for (int v = 0; v < 256; ++v) c.lut[(size_t)v] = (uint8_t)v;
            memcpy(c.lut.data() + kMarker, window.data(), kWindow);
            const uint16_t *endp = c.sym.data() + kWindow + c.n;
            std::vector<uint8_t> nw(kWindow);
            for (size_t i = 0; i < kWindow; ++i) nw[i] = c.lut[endp[(ptrdiff_t)i - (ptrdiff_t)kWindow]];
            window.swap(nw);
            member_out = c.ev.empty() ? member_out + c.n : (uint64_t)(c.n - c.ev.back().at);
        }
        carry_window_.swap(window);
        carry_member_out_ = member_out;
        next_bit_ = tail.end_bit;
        at_member_header_ = tail.at_member_header;
        if (tail.eos || failed || !b.error.empty()) {
            stream_done_ = true;
            b.last = true;
        }
        if (trace) {
            const auto t2 = std::chrono::steady_clock::now();
            size_t out = 0, found = 0;
            for (size_t idx : accepted) out += ch[idx]->n;
            for (auto &c : ch) found += c->found;
            fprintf(stderr, "[pargz] batch: %zu chunks, %zu found, %zu accepted, %.1f MB out, decode %.1f ms, accept+windows %.1f ms%s%s\n", nchunks, found,
                    accepted.size(), out / 1e6, std::chrono::duration<double>(t1 - t0).count() * 1e3, std::chrono::duration<double>(t2 - t1).count() * 1e3,
                    tail.starved ? " starved" : "", tail.eos ? " eos" : "");
        }
        for (size_t idx : accepted) b.chunks.push_back(std::move(ch[idx]));
        for (auto &c : ch) give_chunk(std::move(c));                    // the ones that were not accepted
    }

    // ---- step 4: markers -> bytes ---------------------------------------------------------------------------------------
    static void resolve(const Chunk &c, size_t from, size_t n, uint8_t *dst) {
        const uint16_t *s = c.sym.data() + kWindow + from;
        const uint8_t *lut = c.lut.data();
        size_t i = 0;
        for (; i + 16 <= n; i += 16) {
            const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i));
            const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i + 8));
            if ((_mm_movemask_epi8(_mm_or_si128(a, b)) & 0xAAAA) == 0)                       // no marker among the 16: pack
                _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + i), _mm_packus_epi16(a, b));
            else
                for (size_t k = 0; k < 16; ++k) dst[i + k] = lut[s[i + k]];
        }
        for (; i < n; ++i) dst[i] = lut[s[i]];
    }
    // members that end in front of output symbol `at` of chunk c: the running CRC-32 / length must match their trailers
    bool check_events(Chunk &c, size_t at) {
        for (Event &e : c.ev)
            if (e.at == at && !e.checked) {
                e.checked = true;
                if (e.crc != (isize_ ? crc_ : 0u)) { err_ = "gz: CRC-32 mismatch"; return false; }
                if (e.isize != (uint32_t)isize_) { err_ = "gz: length check (ISIZE) failed"; return false; }
                crc_ = 0;
                isize_ = 0;
            }
        return true;
    }

    int fd_ = -1, threads_ = 1;
    uint64_t file_pos_ = 0;
    size_t chunk_bytes_ = 1u << 20, max_chunk_out_ = 48u << 20;
    std::unique_ptr<WorkerPool> pool_, rpool_;
    std::string err_;
    std::string io_error_;       // set by fill_input when a pread fails (EIO ...)
    // consumer side
    std::unique_ptr<Batch> cur_;
    size_t cur_chunk_ = 0;
    uint32_t crc_ = 0;
    uint64_t isize_ = 0;
    bool done_ = false;
    // hand-over
    std::mutex mu_;
    std::condition_variable cv_;
    std::unique_ptr<Batch> ready_;
    bool stop_ = false;
    std::thread bg_;
    std::mutex spare_mu_;
    std::vector<std::unique_ptr<Chunk>> spare_;
    // producer side (bit positions and byte offsets count from the stream's first byte)
    std::vector<uint8_t> in_;
    uint64_t in_base_ = 0;
    size_t in_len_ = 0;
    bool in_eof_ = false;
    uint64_t next_bit_ = 0;
    std::vector<uint8_t> carry_window_;
    uint64_t carry_member_out_ = 0;
    bool stream_done_ = false, at_member_header_ = true;
};

}  // namespace hast

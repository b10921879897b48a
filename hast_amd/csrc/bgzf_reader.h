// bgzf_reader.h -- parallel inflate of BGZF files (blocked gzip: bgzip, htslib, many sequencer pipelines) for the CLIs' ingest.
//
// An ordinary .gz file is ONE deflate stream and can only be inflated by one thread (fast_inflate.h).  A BGZF file is a
// series of small gzip members (<= 64 KB each) whose headers carry the member's compressed size in an extra field ("BC"),
// and whose trailers carry the uncompressed size, so the members of a whole block of the file can be found without
// inflating anything and then be inflated side by side, each straight into its final place.  Every member's CRC-32 and
// size are checked.  A member that is not BGZF-shaped (a plain gzip member appended to the file) ends this reader: the
// caller continues with the serial decoder from that file offset.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace hast {

class BgzfReader {
  public:
    // true when the file at its current position starts with a BGZF member; the position is restored either way
    static bool probe(FILE *f) {
        unsigned char h[18];
        const long at = ftell(f);
        const size_t n = fread(h, 1, sizeof h, f);
        if (at >= 0) fseek(f, at, SEEK_SET);
        size_t hdr, total;
        return n == sizeof h && parse_header(h, n, hdr, total);
    }
    void open(FILE *f, int threads) {
        fp_ = f;
        threads_ = std::max(1, threads);
        in_.clear();
        scan_ = 0;
        pending_handover_ = false;
        file_off_ = ftell(f) < 0 ? 0 : (uint64_t)ftell(f);
        eof_ = false;
        spill_.clear();
        spill_pos_ = 0;
        err_.clear();
    }
    // up to cap bytes; 0 = end of input; -1 = error (error()); -2 = the next member is not BGZF: serial decoder from resume_offset()
    long read(uint8_t *dst, size_t cap) {
        size_t got = 0;
        if (scan_) {                                               // everything in front of scan_ has been inflated by earlier calls
            in_.erase(in_.begin(), in_.begin() + (long)scan_);
            file_off_ += scan_;
            scan_ = 0;
        }
        if (spill_pos_ < spill_.size()) {                         // rest of a member that did not fit the previous call
            const size_t n = std::min(cap, spill_.size() - spill_pos_);
            memcpy(dst, spill_.data() + spill_pos_, n);
            spill_pos_ += n;
            got = n;
            if (got == cap) return (long)got;
        }
        // gather whole members that fit
        std::vector<Member> batch;
        size_t out_bytes = 0;
        for (;;) {
            Member m;
            const int rc = next_member(m);
            if (rc == 0) break;                                   // end of input
            if (rc < 0) {
                if (rc == -2 && (got || !batch.empty())) {        // hand over after what we already have
                    pending_handover_ = true;
                    break;
                }
                return rc;
            }
            if (got + out_bytes + m.isize > cap) {
                if (got + out_bytes == 0) {                        // a single member larger than the caller's buffer: via the spill
                    spill_.resize(m.isize);
                    spill_pos_ = 0;
                    if (!inflate_member(m, spill_.data())) {
                        err_ = "bgzf: a block failed to inflate (damaged data, CRC-32 or size mismatch)";
                        return -1;
                    }
                    scan_ = m.at + m.total;
                    const size_t n = std::min(cap, spill_.size());
                    memcpy(dst, spill_.data(), n);
                    spill_pos_ = n;
                    return (long)n;
                }
                break;                                             // leave it for the next call
            }
            m.out_off = got + out_bytes;
            out_bytes += m.isize;
            batch.push_back(m);
            scan_ = m.at + m.total;
        }
        if (batch.empty()) {
            if (got) return (long)got;
            if (pending_handover_) {
                pending_handover_ = false;
                return -2;
            }
            return 0;
        }
        // inflate them side by side, each into its own place
        const int nt = (int)std::min<size_t>((size_t)threads_, batch.size());
        std::vector<int> ok((size_t)nt, 1);
        auto work = [&](int t) {
            for (size_t i = (size_t)t; i < batch.size(); i += (size_t)nt)
                if (!inflate_member(batch[i], dst + batch[i].out_off)) ok[(size_t)t] = 0;
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &t : th) t.join();
        for (int v : ok)
            if (!v) {
                if (err_.empty()) err_ = "bgzf: a block failed to inflate (damaged data, CRC-32 or size mismatch)";
                return -1;
            }
        return (long)(got + out_bytes);
    }
    const std::string &error() const { return err_; }
    uint64_t resume_offset() const { return file_off_ + scan_; }

  private:
    struct Member {
        size_t at = 0, hdr = 0, total = 0;        // position in in_, header bytes, whole member bytes
        uint32_t crc = 0, isize = 0;
        size_t out_off = 0;
    };
    // gzip header with the BGZF extra subfield: returns header length and total member length
    static bool parse_header(const unsigned char *h, size_t n, size_t &hdr, size_t &total) {
        if (n < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 4) == 0) return false;
        if (h[3] & ~4u) return false;                              // BGZF writers set FEXTRA only
        const size_t xlen = h[10] | ((size_t)h[11] << 8);
        // the BC subfield is the first one in every known writer; look through the part we have
        size_t p = 12;
        while (p + 4 <= std::min(n, 12 + xlen)) {
            const size_t slen = h[p + 2] | ((size_t)h[p + 3] << 8);
            if (h[p] == 'B' && h[p + 1] == 'C' && slen == 2 && p + 6 <= n) {
                total = (size_t)(h[p + 4] | ((size_t)h[p + 5] << 8)) + 1;
                hdr = 12 + xlen;
                return total >= hdr + 8 + 2;
            }
            p += 4 + slen;
        }
        return false;
    }
    void fill() {                                                  // append only: members gathered for the current call stay where they are
        if (eof_) return;
        const size_t old = in_.size();
        in_.resize(old + kChunk);
        const size_t n = fread(in_.data() + old, 1, kChunk, fp_);
        in_.resize(old + n);
        if (n < kChunk) eof_ = true;
    }
    // the member at scan_: 1 found, 0 end of input, -1 damaged, -2 not a BGZF member
    int next_member(Member &m) {
        for (int attempt = 0; attempt < 2; ++attempt) {
            const size_t have = in_.size() - scan_;
            if (have == 0 && eof_) return 0;
            size_t hdr = 0, total = 0;
            if (have >= 18 && parse_header(in_.data() + scan_, have, hdr, total) && have >= total) {
                m.at = scan_;
                m.hdr = hdr;
                m.total = total;
                const unsigned char *t = in_.data() + scan_ + total - 8;
                m.crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
                m.isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
                if (m.isize > (1u << 16)) {
                    err_ = "bgzf: a block claims more than 64 KB";
                    return -1;
                }
                return 1;
            }
            if (have >= 18 && !parse_header(in_.data() + scan_, have, hdr, total)) return -2;
            if (eof_) {
                // fewer than a header's worth of bytes after the last member: padding that does not start a gzip member is
                // ignored, as gzread and the serial decoder do (fast_inflate.h); the beginning of a member is a truncated file
                const unsigned char *q = in_.data() + scan_;
                if (have < 18 && !(have >= 2 && q[0] == 0x1f && q[1] == 0x8b) && !(have == 1 && q[0] == 0x1f)) return 0;
                err_ = "bgzf: the file ends inside a block";
                return -1;
            }
            fill();                                                // need more bytes
        }
        err_ = "bgzf: a block is larger than the read-ahead";
        return -1;
    }
    bool inflate_member(const Member &m, uint8_t *out) const {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return false;
        zs.next_in = const_cast<Bytef *>(in_.data() + m.at + m.hdr);
        zs.avail_in = (uInt)(m.total - m.hdr - 8);
        zs.next_out = out;
        zs.avail_out = m.isize;
        const int rc = inflate(&zs, Z_FINISH);
        const bool ok = (rc == Z_STREAM_END || (rc == Z_BUF_ERROR && m.isize == 0 && zs.avail_in == 0)) && zs.total_out == m.isize &&
                        (uint32_t)crc32(crc32(0L, Z_NULL, 0), out, m.isize) == m.crc;
        inflateEnd(&zs);
        return ok;
    }
    static constexpr size_t kChunk = 8u << 20;
    FILE *fp_ = nullptr;
    int threads_ = 1;
    std::vector<unsigned char> in_;
    size_t scan_ = 0;                          // start of the next member to look at, in in_
    uint64_t file_off_ = 0;
    bool eof_ = false, pending_handover_ = false;
    std::vector<uint8_t> spill_;
    size_t spill_pos_ = 0;
    std::string err_;
};

}  // namespace hast

// gz_chain.h -- the host side of the device inflate (gz_core.h, gz_kernels.hip): which decoded chunks form THE stream.
//
// Every chunk of the compressed bytes was decoded from the first dynamic-block header its search found (a "candidate": an
// unproven start).  A candidate is ACCEPTED iff the accepted chunk in front of it ended exactly at its start -- by induction
// from the first block of the first member every accepted chunk starts at a true block boundary, whatever the search has
// mistaken for a header (par_inflate.h does the same on host threads).  Where the chain does not meet the next candidate (a
// false or missing candidate, a chunk that ran out of room, a member's end followed by the next member's header) a FOLLOW-UP
// job with a known start decodes the hole.  Follow-up jobs are planned speculatively -- the walk goes on as if each landed on
// its target, so one round of jobs usually closes all holes -- and checked when their results are back.
// Pure host code, no HIP: tests/native/test_gz_core.cpp drives it with a CPU stand-in for the kernels.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "gz_core.h"

namespace hast {
namespace gz {

struct Accepted {
    ChunkJob job;                // as decoded (sym_off / n_out say where its symbols are)
    uint64_t out_off = 0;        // offset of its first byte in the inflated stream
    bool no_history = false;     // a member starts with it: its symbols hold no markers
    bool member_end = false;     // a member ends behind it: trailer values below
    uint32_t want_crc = 0, want_isize = 0;
    bool is_gap = false;         // decoded by a follow-up job (the owner may have given it a buffer of its own)
    uint64_t tag = 0;            // the owner's handle for the symbol buffer (copied from the planned job)
};

class Chain {
  public:
    // reads n bytes at file offset off (fewer at the end of the file); false = I/O error
    using ReadFn = std::function<bool(uint64_t off, size_t n, uint8_t *dst, size_t *got)>;

    void begin(uint64_t file_size, ReadFn read) {
        file_size_ = file_size;
        read_ = std::move(read);
        items_.clear();
        cands_.clear();
        err_.clear();
        st_ = State{};
        st_.at_header = true;
        handed_ = 0;
        confirmed_ = 0;
        out_confirmed_ = 0;
        waiting_gaps_ = false;
        all_in_ = false;
        searched_to_ = 0;
        (void)parse_member_header();                 // the first member's header: first_deflate_bit() is where chunk 0 starts
    }
    // after begin(): the first bit of the first member's deflate data (the owner decodes its first chunk from there, as a
    // known start without history), or ~0 when there is no member to decode (empty file, not gzip: finished() / failed() say which)
    uint64_t first_deflate_bit() const { return (st_.eos || !err_.empty() || st_.at_header) ? ~0ull : st_.end; }
    // results of the nominal pass, ascending by from_bit; all_in: no more candidates will follow
    void add_candidates(const ChunkJob *j, size_t n, bool all_in) {
        if (!waiting_gaps_ && confirmed_ == items_.size() && st_.ci > 65536) {      // nothing refers to the old ones any more
            cands_.erase(cands_.begin(), cands_.begin() + (long)st_.ci);
            st_.ci = 0;
        }
        cands_.insert(cands_.end(), j, j + n);
        for (size_t i = 0; i < n; ++i) searched_to_ = std::max(searched_to_, j[i].stop_bit);
        all_in_ = all_in;
    }
    // eager: where the search has found no block start between the chain's end and the end of what it has looked at (stored blocks,
    // fixed-Huffman blocks: a member of incompressible bytes), a follow-up job decodes on through that stretch at once instead of
    // waiting for a candidate behind it.  For an owner that cannot keep the input until one comes (gz_api.cpp: the ring)
    void set_eager(bool on) { eager_ = on; }
    // Walks on.  Follow-up jobs to run now are appended to `gaps` (from_bit, stop_bit, flags set; `want_syms` says how much
    // room to give; the owner fills sym_off / sym_cap and runs them, then calls gap_done with the results in the same order).
    // input_bits: compressed bits present on the device.  false = nothing to run (finished, waiting for candidates, or failed).
    struct Gap { ChunkJob job; uint64_t want_syms; };
    bool plan(std::vector<Gap> &gaps, uint64_t input_bits) {
        gaps.clear();
        if (!err_.empty() || waiting_gaps_ || st_.eos) return false;
        for (;;) {
            if (st_.eos) break;
            if (st_.at_header) {
                if (!parse_member_header()) break;             // (eos, error, or not enough bytes yet)
                if (st_.eos) break;
            }
            // the next candidate at or behind the chain's end
            // (a candidate whose first block did not fit its room holds no data, but where it starts is as good a target for a follow-up
            // job as any: waiting for the next one that DID fit is waiting for the end of the input when every block of a stream is
            // that large -- and a ring on the device never sees the end of the input before the chain has moved on)
            while (st_.ci < cands_.size() && (!(cands_[st_.ci].status & kStFound) || cands_[st_.ci].start_bit < st_.end ||
                                              ((cands_[st_.ci].status & kStNoBlock) && cands_[st_.ci].start_bit == st_.end)))
                st_.ci++;
            const bool have = st_.ci < cands_.size();
            if (have && cands_[st_.ci].start_bit == st_.end) {
                // (a member's first chunk decoded with markers allowed holds none if the stream is valid; if it is not, the
                // member's CRC-32 says so)
                Item it;
                it.acc.job = cands_[st_.ci];
                it.acc.no_history = (cands_[st_.ci].flags & kJobNoHistory) != 0;
                it.acc.tag = cands_[st_.ci].sym_off;
                it.before = st_;
                st_.ci++;
                if (!finish_item(it, input_bits)) break;
                items_.push_back(it);
                continue;
            }
            uint64_t open_stop = ~0ull;                            // (no candidate: the follow-up job runs to the end of the input ...)
            if (!have && !all_in_) {
                if (!eager_ || searched_to_ <= st_.end) break;     // more candidates to come
                open_stop = searched_to_;                          // (... or, eager, through what has been searched)
            }
            // a hole: from the chain's end to the next candidate (or to the end of the input)
            if (st_.retry_bits && input_bits <= st_.retry_bits) break;   // starved before: wait for more input
            Item it;
            it.is_gap = true;
            it.before = st_;
            ChunkJob &g = it.acc.job;
            memset(&g, 0, sizeof(g));
            g.from_bit = g.start_bit = st_.end;
            g.stop_bit = have ? cands_[st_.ci].start_bit : open_stop;
            g.flags = kJobKnown | (st_.no_history_next ? kJobNoHistory : 0u);
            it.acc.no_history = st_.no_history_next;
            it.acc.is_gap = true;
            it.target = have ? g.stop_bit : ~0ull;                 // (no candidate, no landing: where it ends is where the walk goes on)
            const uint64_t span_bytes = ((g.stop_bit != ~0ull ? g.stop_bit : std::min<uint64_t>(input_bits, st_.end + (8ull << 23))) - st_.end) / 8 + 64;
            uint64_t want = std::max<uint64_t>(span_bytes * 16, 1u << 16);
            if (st_.grow) want = std::max<uint64_t>(want, st_.grow);
            gaps.push_back(Gap{g, want});
            it.pending = true;
            items_.push_back(it);
            // speculation: the job lands on its target
            if (!have) break;                                      // (runs to the end of the stream: nothing to speculate about)
            st_.end = g.stop_bit;
            st_.no_history_next = false;
            st_.grow = 0;
            st_.retry_bits = 0;
        }
        waiting_gaps_ = !gaps.empty();
        confirm();
        return !gaps.empty();
    }
    void gap_done(const ChunkJob *res, size_t n, uint64_t input_bits) {
        waiting_gaps_ = false;
        size_t k = 0;
        for (size_t i = 0; i < items_.size() && k < n; ++i) {
            Item &it = items_[i];
            if (!it.pending) continue;
            const ChunkJob &r = res[k++];
            it.pending = false;
            it.acc.job = r;
            it.acc.tag = r.sym_off;
            it.acc.member_end = false;
            const bool landed = (r.status & kStStop) && !(r.status & (kStFinal | kStError | kStNoRoom | kStStarved)) && r.end_bit == it.target;
            if (landed) continue;
            // it ended somewhere else: what the walk assumed behind it is void; go on from where it really ended
            items_.resize(i + 1);
            st_ = it.before;
            Item copy = items_[i];
            items_.pop_back();
            st_.no_history_next = false;
            st_.grow = 0;
            st_.retry_bits = 0;
            if (finish_item(copy, input_bits)) items_.push_back(copy);
            break;
        }
        confirm();
    }
    // accepted chunks, in stream order, that are final (no unchecked follow-up job in front of them) and not yet handed out
    size_t take_confirmed(std::vector<Accepted> &out) {
        size_t n = 0;
        for (; handed_ < confirmed_; ++handed_, ++n) out.push_back(items_[handed_].acc);
        if (handed_ > 8192) {                                   // handed-out items are of no further use here
            items_.erase(items_.begin(), items_.begin() + (long)handed_);
            confirmed_ -= handed_;
            handed_ = 0;
        }
        return n;
    }
    bool finished() const { return err_.empty() && st_.eos && !waiting_gaps_ && confirmed_ == items_.size(); }
    bool failed() const { return !err_.empty(); }
    const std::string &error() const { return err_; }
    uint64_t proven_end_bit() const { return st_.end; }
    uint64_t out_bytes() const { return out_confirmed_; }

  private:
    struct State {
        uint64_t end = 0;            // first bit behind the chain (a block boundary, or -- at_header -- a member header's first byte * 8)
        size_t ci = 0;               // next candidate to look at
        bool at_header = false, eos = false, no_history_next = false;
        uint64_t grow = 0;           // the next follow-up job needs at least this much room
        uint64_t retry_bits = 0;     // the chain's end starved with this much input: wait for more
    };
    struct Item {
        Accepted acc;
        bool is_gap = false, pending = false;
        uint64_t target = 0;
        State before;
    };

    // books an item whose results are in: its output, what its end means for the walk.  false = stop walking (error / wait).
    bool finish_item(Item &it, uint64_t input_bits) {
        ChunkJob &j = it.acc.job;
        it.acc.member_end = false;
        st_.end = j.end_bit;
        st_.no_history_next = false;
        st_.grow = 0;
        st_.retry_bits = 0;
        // a member's first job that got nowhere: what follows it still is the member's first byte
        if (it.acc.no_history && j.end_bit == j.start_bit && j.n_out == 0 && !(j.status & kStFinal)) st_.no_history_next = true;
        if (j.status & kStFinal) {
            // the member's trailer: CRC-32 and ISIZE, byte aligned behind the final block
            const uint64_t at = (j.end_bit + 7) >> 3;
            uint8_t t[8];
            size_t got = 0;
            if (!read_(at, 8, t, &got)) { err_ = "gz: read failed"; return false; }
            if (got < 8) { err_ = "gz: truncated trailer"; return false; }
            it.acc.member_end = true;
            it.acc.want_crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
            it.acc.want_isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
            st_.end = (at + 8) * 8;
            st_.at_header = true;
            return true;
        }
        if (j.status & kStError) {
            err_ = std::string("gz: damaged input (") + err_name(j.err_code) + ")";
            return false;
        }
        if (j.status & kStStarved) {
            if (j.limit_bits && j.limit_bits < input_bits) {
                // not the input ended but the job's VIEW of a ring (ChunkJob.limit_bits: its lap and a piece behind it) -- a job that ran
                // on from its first boundary into blocks no search can find (a member's final block).  What it committed stands; a
                // follow-up job, with the view of where IT starts, goes on from there.  A job that committed nothing met a block
                // longer than the margin
                if (j.end_bit > j.start_bit) return true;
                err_ = "gz: a deflate block reaches further than the ring of compressed bytes on the device lets one job read";
                return false;
            }
            if (input_bits >= file_size_ * 8) { err_ = "gz: input ends inside a compressed block"; return false; }
            st_.retry_bits = input_bits;                           // the rest of this block is not on the device yet
            return true;
        }
        if (j.status & kStNoRoom) {
            // the next block did not fit: the follow-up job gets more room (a block that fits no buffer we are willing to
            // give is refused: 2^26 symbols = 64 MB of one block's output)
            const uint64_t had = j.sym_cap;
            st_.grow = (j.status & kStNoBlock) ? std::max<uint64_t>(had * 8, 1u << 22) : std::max<uint64_t>(had, 1u << 18);
            if (st_.grow > (1ull << 26)) { err_ = "gz: a deflate block of more than 64 MB (not supported by the device decoder)"; return false; }
            return true;
        }
        return true;
    }
    // RFC 1952 member header at byte st_.end / 8; sets st_.end to the first deflate bit.  false = stop (eos / error / wait)
    bool parse_member_header() {
        const uint64_t at = st_.end >> 3;
        if (at >= file_size_) { st_.eos = true; return false; }
        uint8_t h[10];
        size_t got = 0;
        if (!read_(at, 10, h, &got)) { err_ = "gz: read failed"; return false; }
        if (got < 2 || h[0] != 0x1f || h[1] != 0x8b) {
            if (at == 0) { err_ = "gz: not a gzip file"; return false; }
            st_.eos = true;                                        // bytes that do not start a member: ignored, as gzread does
            return false;
        }
        if (got < 10) { err_ = "gz: truncated header"; return false; }
        if (h[2] != 8) { err_ = "gz: unknown compression method"; return false; }
        const int flg = h[3];
        uint64_t q = at + 10;
        auto rd = [&](uint64_t off, size_t n, uint8_t *dst) {
            size_t g = 0;
            return read_(off, n, dst, &g) && g == n;
        };
        if (flg & 4) {
            uint8_t x[2];
            if (!rd(q, 2, x)) { err_ = "gz: truncated header"; return false; }
            q += 2 + (x[0] | ((uint64_t)x[1] << 8));
        }
        for (int bit = 8; bit <= 16; bit <<= 1)
            if (flg & bit) {
                for (;;) {
                    uint8_t buf[256];
                    size_t g = 0;
                    if (!read_(q, sizeof(buf), buf, &g) || g == 0) { err_ = "gz: truncated header"; return false; }
                    const void *z = memchr(buf, 0, g);
                    if (z) { q += (size_t)((const uint8_t *)z - buf) + 1; break; }
                    q += g;
                }
            }
        if (flg & 2) q += 2;
        if (q > file_size_) { err_ = "gz: truncated header"; return false; }
        st_.end = q * 8;
        st_.at_header = false;
        st_.no_history_next = true;
        return true;
    }
    // items become final in stream order, up to the first follow-up job whose results are not in; that is also where the
    // offsets in the inflated stream are known
    void confirm() {
        while (confirmed_ < items_.size() && !items_[confirmed_].pending) {
            items_[confirmed_].acc.out_off = out_confirmed_;
            out_confirmed_ += items_[confirmed_].acc.job.n_out;
            confirmed_++;
        }
    }
    static const char *err_name(uint32_t c) {
        static const char *n[] = {"", "reserved block type", "stored block length check failed", "bad code counts", "bad code-length code", "repeat without a previous length",
                                  "too many code lengths", "no end-of-block code", "over-subscribed Huffman code", "incomplete code", "Huffman table too large",
                                  "invalid literal/length code", "invalid distance code", "distance too far back"};
        return c < sizeof(n) / sizeof(n[0]) ? n[c] : "?";
    }

    uint64_t file_size_ = 0;
    ReadFn read_;
    std::vector<ChunkJob> cands_;
    bool all_in_ = false, eager_ = false;
    uint64_t searched_to_ = 0;       // the nominal passes have looked for block starts up to this bit
    std::vector<Item> items_;
    size_t handed_ = 0, confirmed_ = 0;
    uint64_t out_confirmed_ = 0;
    bool waiting_gaps_ = false;
    State st_;
    std::string err_;
};

}  // namespace gz
}  // namespace hast

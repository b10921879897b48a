// seqstream.h -- FASTA/FASTQ records -> one byte stream of bases for the k-mer counter (stage 00 ingest).
//
// The reference feeds its parental read files to `jellyfish count` (00.build_unshare_kmers_by_jellyfish/
// build_unshared_kmers.sh:187-190,219-222).  What that reader does with a file was probed on the vendored binary
// (jellyfish 2.3.0) and is reproduced here:
//   * the first byte of an input decides: '>' FASTA, '@' FASTQ, anything else is refused; an empty input is fine;
//   * FASTA: lines up to the next '>' line are ONE sequence (k-mers run across line breaks);
//   * FASTQ: header line, sequence lines up to the line that starts with '+', then as many quality bytes as bases
//     (over any number of lines), then the next '@' header; sequence and quality may span several lines;
//   * '\r' in front of '\n' belongs to the line break ("ACG\r\nTTT" is ACGTTT); blank lines are skipped;
//   * k-mers never span records.
// Deviations (both are silent data loss in the reference, an explicit error or the obvious result here): a quality
// string whose length differs from the sequence's, or a non-'@' line where a header is due, is an error; the last
// record counts even when the input does not end with '\n'.
//
// Output: the records' bases, each record followed by one '\n' (any non-base byte separates k-mers for the kernel).
#pragma once
#include <cstddef>
#include <cstring>
#include <string>

namespace hast {

// Sink: void append(const char *p, size_t n);  (n > 0: bases of the current record)  void separator();
template <class Sink>
class SeqParser {
  public:
    explicit SeqParser(Sink &sink) : sink_(sink) {}
    // feed the next piece of the input; pieces may cut lines anywhere.  Returns false on a format error (see error()).
    bool feed(const char *p, size_t n) {
        const char *end = p + n;
        if (n && fmt_ == kUnknown) {
            const char first = carry_.empty() ? *p : carry_[0];
            first_ = first;
            fmt_ = first == '>' ? kFasta : first == '@' ? kFastq : kBad;
            if (fmt_ == kBad) return fail("unsupported format: the input starts with neither '>' nor '@'");
        }
        while (p < end) {
            const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
            if (!nl) {
                carry_.append(p, (size_t)(end - p));
                break;
            }
            bool ok;
            if (carry_.empty()) ok = line(p, (size_t)(nl - p));
            else {
                carry_.append(p, (size_t)(nl - p));
                ok = line(carry_.data(), carry_.size());
                carry_.clear();
            }
            if (!ok) return false;
            p = nl + 1;
        }
        return true;
    }
    // true when the input seen so far ends exactly between two records: appending another input to it (`zcat a b`)
    // then parses like the two inputs one by one
    bool at_record_boundary() const {
        if (!carry_.empty()) return false;
        return fmt_ == kUnknown || fmt_ == kFasta || (state_ == kHeader && !in_record_);
    }
    char first_byte() const { return first_; }
    // end of this input: the unterminated last line, and the record in progress
    bool finish() {
        if (!carry_.empty()) {
            const bool ok = line(carry_.data(), carry_.size());
            carry_.clear();
            if (!ok) return false;
        }
        if (fmt_ == kFastq && state_ == kQual) return fail("FASTQ: the input ends inside a quality string");
        if (in_record_) sink_.separator();
        in_record_ = false;
        fmt_ = kUnknown;
        state_ = kHeader;
        return true;
    }
    const std::string &error() const { return err_; }
    size_t records() const { return records_; }

  private:
    enum Fmt { kUnknown, kFasta, kFastq, kBad };
    enum State { kHeader, kSeq, kQual };
    bool fail(const char *what) {
        err_ = what;
        return false;
    }
    bool line(const char *p, size_t n) {
        while (n && p[n - 1] == '\r') --n;
        if (fmt_ == kFasta) {
            if (n && p[0] == '>') {
                if (in_record_) sink_.separator();
                in_record_ = true;
                ++records_;
            } else if (n) sink_.append(p, n);
            return true;
        }
        switch (state_) {
        case kHeader:
            if (n == 0) return true;
            if (p[0] != '@') return fail("FASTQ: a record does not start with '@' (or the previous quality string is too long)");
            state_ = kSeq;
            seq_len_ = 0;
            in_record_ = true;
            ++records_;
            return true;
        case kSeq:
            if (n == 0) return true;
            if (p[0] == '+') {
                sink_.separator();
                in_record_ = false;
                qual_len_ = 0;
                state_ = seq_len_ ? kQual : kHeader;
                return true;
            }
            sink_.append(p, n);
            seq_len_ += n;
            return true;
        case kQual:
            qual_len_ += n;
            if (qual_len_ < seq_len_) return true;
            if (qual_len_ != seq_len_) return fail("FASTQ: a quality string is longer than its sequence");
            state_ = kHeader;
            return true;
        }
        return true;
    }
    Sink &sink_;
    Fmt fmt_ = kUnknown;
    State state_ = kHeader;
    std::string carry_, err_;
    size_t seq_len_ = 0, qual_len_ = 0, records_ = 0;
    bool in_record_ = false;
    char first_ = 0;
};

}  // namespace hast

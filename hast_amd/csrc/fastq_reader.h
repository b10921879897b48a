// fastq_reader.h -- host-side input framing of the classify CLI.
//
// Reproduces what the reference's reader accepts (01.classify_stlfr_reads/classify.cpp:238-278):
//   * gzip iff the file NAME ends in ".gz" (:245-254), otherwise raw bytes;
//   * a record is four '\n'-separated lines; only line 1 (header) and line 2 (bases) are used, no
//     '@'/'+' validation, no CR stripping (:257-268);
//   * the loop ends when the HEADER getline reaches end-of-file before a '\n' (:257), so a record
//     whose header line is not newline-terminated is dropped, while an unterminated line 2..4 of the
//     last record is still accepted.
// Implementation is ours: block reads (fread / gzread with a large buffer) + memchr.
#pragma once
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <string_view>
#include <vector>

namespace hast {

class LineSource {
  public:
    ~LineSource() { close(); }
    bool open(const std::string &path) {
        close();
        const size_t n = path.size();
        const bool gz = n > 3 && path.compare(n - 3, 3, ".gz") == 0;          // classify.cpp:245-249
        if (gz) {
            gz_ = gzopen(path.c_str(), "rb");
            if (!gz_) return false;
            gzbuffer(gz_, 4u << 20);
        } else {
            fp_ = fopen(path.c_str(), "rb");
            if (!fp_) return false;
        }
        buf_.resize(16u << 20);
        pos_ = len_ = 0;
        eof_ = false;
        return true;
    }
    void close() {
        if (gz_) gzclose(gz_);
        if (fp_) fclose(fp_);
        gz_ = nullptr;
        fp_ = nullptr;
    }
    // Next line without its '\n'.  hit_eof is what std::getline(...).eof() would report: true iff the
    // end of input was reached before a '\n'.  The view is valid until the next call.
    std::string_view getline(bool &hit_eof) {
        size_t scanned = 0;
        for (;;) {
            const char *line = buf_.data() + pos_;
            const void *nl = memchr(line + scanned, '\n', len_ - pos_ - scanned);
            if (nl) {
                size_t n = (const char *)nl - line;
                pos_ += n + 1;
                hit_eof = false;
                return std::string_view(line, n);
            }
            scanned = len_ - pos_;
            if (!fill()) {
                line = buf_.data() + pos_;
                size_t n = len_ - pos_;
                pos_ = len_;
                hit_eof = true;
                return std::string_view(line, n);
            }
        }
    }

  private:
    bool fill() {
        if (eof_) return false;
        if (pos_ > 0) {
            memmove(buf_.data(), buf_.data() + pos_, len_ - pos_);
            len_ -= pos_;
            pos_ = 0;
        }
        if (len_ == buf_.size()) buf_.resize(buf_.size() * 2);
        size_t want = buf_.size() - len_;
        if (want > (1u << 30)) want = 1u << 30;
        long got = gz_ ? (long)gzread(gz_, buf_.data() + len_, (unsigned)want) : (long)fread(buf_.data() + len_, 1, want, fp_);
        if (got <= 0) {
            eof_ = true;
            return false;
        }
        len_ += (size_t)got;
        return true;
    }
    FILE *fp_ = nullptr;
    gzFile gz_ = nullptr;
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;
    bool eof_ = false;
};

}  // namespace hast

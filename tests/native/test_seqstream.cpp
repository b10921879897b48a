// Host-only driver for hast_amd/csrc/seqstream.h (test infrastructure): parses the files given on the command line as ONE
// input each (or, with "-c", as one concatenated stream), feeding the parser in pieces of `piece` bytes, and writes the
// resulting base stream to stdout.  tests/test_seqstream_cpu.py counts that stream's k-mers and compares with the
// pinned file reader.  Exit 3 + message on a format error.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../hast_amd/csrc/seqstream.h"

struct Out {
    std::string s;
    void append(const char *p, size_t n) { s.append(p, n); }
    void separator() { s.push_back('\n'); }
};

int main(int argc, char **argv) {
    bool concat = false;
    size_t piece = 1 << 16;
    std::vector<std::string> files;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-c")) concat = true;
        else if (!strcmp(argv[i], "-p")) piece = (size_t)atol(argv[++i]);
        else files.push_back(argv[i]);
    }
    Out out;
    hast::SeqParser<Out> parser(out);
    for (size_t i = 0; i < files.size(); ++i) {
        FILE *f = fopen(files[i].c_str(), "rb");
        if (!f) return 2;
        std::vector<char> buf(piece);
        size_t n;
        while ((n = fread(buf.data(), 1, piece, f)) > 0)
            if (!parser.feed(buf.data(), n)) {
                fprintf(stderr, "%s\n", parser.error().c_str());
                return 3;
            }
        fclose(f);
        if ((!concat || i + 1 == files.size()) && !parser.finish()) {
            fprintf(stderr, "%s\n", parser.error().c_str());
            return 3;
        }
    }
    fwrite(out.s.data(), 1, out.s.size(), stdout);
    return 0;
}

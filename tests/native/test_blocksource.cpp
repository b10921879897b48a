// Host-only driver for hast::BlockSource (hast_amd/csrc/ingest.h), the byte source of every CLI (test infrastructure):
//   test_blocksource [-b block_bytes] [-u] FILE...
// Opens all FILEs at once (as the CLIs do: the ordinary .gz files open together share the inflate-thread budget), then writes
// their decoded bytes to stdout file after file.  -u reads through read_into() (the un-threaded route of the GPU framing path)
// instead of the background reader's blocks.  Exit 3 + message when a source reports an error.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../hast_amd/csrc/ingest.h"

int main(int argc, char **argv) {
    size_t block = 1u << 20;
    bool unthreaded = false;
    std::vector<std::string> paths;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-b")) block = (size_t)atol(argv[++i]);
        else if (!strcmp(argv[i], "-u")) unthreaded = true;
        else paths.push_back(argv[i]);
    }
    std::vector<std::unique_ptr<hast::BlockSource>> src;
    for (const std::string &p : paths) {
        src.emplace_back(new hast::BlockSource);
        if (!src.back()->open(p, block, !unthreaded)) {
            fprintf(stderr, "cannot open %s\n", p.c_str());
            return 2;
        }
    }
    for (size_t i = 0; i < src.size(); ++i) {
        if (unthreaded) {
            std::vector<char> buf(block);
            for (;;) {
                std::string trouble;
                const size_t n = src[i]->read_into(buf.data(), buf.size(), trouble);
                fwrite(buf.data(), 1, n, stdout);
                if (!trouble.empty()) { fprintf(stderr, "%s: %s\n", paths[i].c_str(), trouble.c_str()); return 3; }
                if (n < buf.size()) break;
            }
        } else {
            for (;;) {
                std::vector<char> b = src[i]->next();
                if (b.empty()) break;
                fwrite(b.data() + hast::BlockSource::kFrontPad, 1, b.size() - hast::BlockSource::kFrontPad, stdout);
                src[i]->recycle(std::move(b));
            }
            const std::string e = src[i]->error();
            if (!e.empty()) { fprintf(stderr, "%s: %s\n", paths[i].c_str(), e.c_str()); return 3; }
        }
        src[i]->close();
    }
    return 0;
}

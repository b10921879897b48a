// Host-only driver for hast_amd/csrc/fast_inflate.h (test infrastructure): decodes the file given on the command line in
// read() calls of `piece` bytes with an input buffer of `inbuf` bytes and writes the result to stdout;
// with "-z" it uses zlib's gzread instead (the behaviour to match).  Exit 3 + message on a decoding error.
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../hast_amd/csrc/bgzf_reader.h"
#include "../../hast_amd/csrc/fast_inflate.h"
#include "../../hast_amd/csrc/par_inflate.h"

int main(int argc, char **argv) {
    bool use_zlib = false, quiet = false, bgzf = false, par = false;
    size_t chunk = 1u << 20, max_out = 48u << 20;
    int threads = 4;
    size_t piece = 1 << 20, inbuf = 1 << 20;
    const char *path = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-z")) use_zlib = true;
        else if (!strcmp(argv[i], "-q")) quiet = true;
        else if (!strcmp(argv[i], "-b")) bgzf = true;                 // BgzfReader (+ hand-over to the serial decoder), as BlockSource does
        else if (!strcmp(argv[i], "-P")) par = true;                  // ParGzReader (several threads on one ordinary gzip stream)
        else if (!strcmp(argv[i], "-c")) chunk = (size_t)atol(argv[++i]);   // ... with this many compressed bytes per chunk
        else if (!strcmp(argv[i], "-m")) max_out = (size_t)atol(argv[++i]); // ... and at most about this much output per chunk
        else if (!strcmp(argv[i], "-t")) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-p")) piece = (size_t)atol(argv[++i]);
        else if (!strcmp(argv[i], "-i")) inbuf = (size_t)atol(argv[++i]);
        else path = argv[i];
    }
    std::vector<uint8_t> buf(piece);
    size_t total = 0;
    const auto t0 = std::chrono::steady_clock::now();
    if (use_zlib) {
        gzFile f = gzopen(path, "rb");
        if (!f) return 2;
        gzbuffer(f, 4u << 20);
        int n;
        while ((n = gzread(f, buf.data(), (unsigned)piece)) > 0) {
            if (!quiet) fwrite(buf.data(), 1, (size_t)n, stdout);
            total += (size_t)n;
        }
        if (n < 0) return 3;
        gzclose(f);
    } else {
        FILE *f = fopen(path, "rb");
        if (!f) return 2;
        hast::GzInflater z;
        long n;
        if (par) {
            if (!hast::ParGzReader::usable(f)) return 4;
            hast::ParGzReader pz;
            pz.open(f, threads, chunk, max_out);
            while ((n = pz.read(buf.data(), piece)) > 0) {
                if (!quiet) fwrite(buf.data(), 1, (size_t)n, stdout);
                total += (size_t)n;
            }
            if (n < 0) {
                fprintf(stderr, "%s\n", pz.error().c_str());
                return 3;
            }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            fprintf(stderr, "%zu bytes in %.3f s = %.1f MB/s\n", total, dt, total / dt / 1e6);
            return 0;
        }
        if (bgzf) {
            if (!hast::BgzfReader::probe(f)) return 4;
            hast::BgzfReader b;
            b.open(f, threads);
            while ((n = b.read(buf.data(), piece)) > 0) {
                if (!quiet) fwrite(buf.data(), 1, (size_t)n, stdout);
                total += (size_t)n;
            }
            if (n == -1) {
                fprintf(stderr, "%s\n", b.error().c_str());
                return 3;
            }
            if (n == 0) return 0;
            fseek(f, (long)b.resume_offset(), SEEK_SET);            // -2: an ordinary member follows
        }
        z.open(f, inbuf, bgzf);                                     // after a hand-over the stream is a continuation, as in BlockSource
        while ((n = z.read(buf.data(), piece)) > 0) {
            if (!quiet) fwrite(buf.data(), 1, (size_t)n, stdout);
            total += (size_t)n;
        }
        if (n < 0) {
            fprintf(stderr, "%s\n", z.error().c_str());
            return 3;
        }
        fclose(f);
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "%zu bytes in %.3f s = %.1f MB/s\n", total, dt, total / dt / 1e6);
    return 0;
}

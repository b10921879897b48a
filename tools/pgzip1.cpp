// pgzip1.cpp -- measurement tool: compress a file into ONE gzip member with many threads, the way pigz does, so that the C2-size inputs of
// tools/gpu/cli_c2.sh (2 x 34 GB of FASTQ) can be made in minutes on the GPU box (one `gzip -6` thread takes 16 MB/s there: 35 minutes per
// file) and still look like what HAST is given: a single deflate stream per .fq.gz (HAST.sh:162-166), here > 2 GB, so that the device
// inflate's ring engages by itself.
//   pgzip1 <in> <out.gz> [level=6] [threads=16] [chunk_mb=32]
// Each chunk of the input is deflated on its own (raw deflate, the 32 KB in front of it as the dictionary, so matches reach back over
// chunk borders as in a stream written by one thread) and ends with a sync flush (an empty stored block, byte aligned) except the last,
// which ends the stream; the pieces are written in order behind one gzip header, the CRC-32s are combined (crc32_combine), ISIZE is the
// length mod 2^32.  `gzip -t` / zlib read the result as any other .gz file.
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: pgzip1 in out.gz [level=6] [threads=16] [chunk_mb=32]\n");
        return 1;
    }
    const int level = argc > 3 ? atoi(argv[3]) : 6, threads = argc > 4 ? atoi(argv[4]) : 16;
    const size_t chunk = (size_t)(argc > 5 ? atoi(argv[5]) : 32) << 20;
    const int fd = open(argv[1], O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0) { perror(argv[1]); return 2; }
    const size_t size = (size_t)sb.st_size, n_chunks = size ? (size + chunk - 1) / chunk : 1;
    FILE *out = fopen(argv[2], "wb");
    if (!out) { perror(argv[2]); return 2; }
    static const unsigned char head[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
    fwrite(head, 1, 10, out);

    struct Piece { std::vector<unsigned char> z; unsigned long crc = 0; size_t len = 0; bool done = false; };
    std::vector<Piece> pieces(n_chunks);
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<size_t> next{0};
    size_t written = 0;                                    // pieces written so far (a worker may run at most 4 * threads pieces ahead)
    std::atomic<bool> failed{false};
    auto work = [&] {
        std::vector<unsigned char> in(chunk + 32768);
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n_chunks) return;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return i < written + 4 * (size_t)threads; });
            }
            const size_t off = i * chunk, len = std::min(chunk, size - off), dict = off ? 32768 : 0;
            size_t got = 0;
            while (got < len + dict) {
                const ssize_t r = pread(fd, in.data() + got, len + dict - got, (off_t)(off - dict + got));
                if (r <= 0) { failed = true; break; }
                got += (size_t)r;
            }
            Piece &p = pieces[i];
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { failed = true; return; }
            if (dict) deflateSetDictionary(&zs, in.data(), 32768);
            p.z.resize(deflateBound(&zs, (uLong)len) + 64);
            zs.next_in = in.data() + dict;
            zs.next_out = p.z.data();
            size_t in_left = len, out_at = 0;
            const bool last = i + 1 == n_chunks;
            for (;;) {                                      // (avail_in / avail_out are 32-bit: feed in pieces of <= 1 GB)
                const size_t feed = std::min<size_t>(in_left, 1u << 30);
                zs.avail_in = (uInt)feed;
                zs.avail_out = (uInt)std::min<size_t>(p.z.size() - out_at, 1u << 30);
                const uInt out0 = zs.avail_out;
                const int rc = deflate(&zs, in_left == feed ? (last ? Z_FINISH : Z_SYNC_FLUSH) : Z_NO_FLUSH);
                in_left -= feed - zs.avail_in;
                out_at += out0 - zs.avail_out;
                if (rc == Z_STREAM_END || (rc == Z_OK && in_left == 0 && zs.avail_out != 0)) break;
                if (rc != Z_OK && rc != Z_BUF_ERROR) { failed = true; break; }
            }
            deflateEnd(&zs);
            p.z.resize(out_at);
            p.crc = crc32(0L, Z_NULL, 0);
            for (size_t at = 0; at < len; at += 1u << 30) p.crc = crc32(p.crc, in.data() + dict + at, (uInt)std::min<size_t>(len - at, 1u << 30));
            p.len = len;
            std::lock_guard<std::mutex> g(mu);
            p.done = true;
            cv.notify_all();
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(work);
    unsigned long crc = crc32(0L, Z_NULL, 0);
    for (size_t i = 0; i < n_chunks; i++) {
        {
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&] { return pieces[i].done || failed.load(); });
        }
        if (failed) break;
        Piece &p = pieces[i];
        if (fwrite(p.z.data(), 1, p.z.size(), out) != p.z.size()) failed = true;
        crc = crc32_combine(crc, p.crc, (z_off_t)p.len);
        std::vector<unsigned char>().swap(p.z);
        std::lock_guard<std::mutex> g(mu);
        written = i + 1;
        cv.notify_all();
    }
    if (failed) {
        {
            std::lock_guard<std::mutex> g(mu);
            written = n_chunks;
        }
        cv.notify_all();
    }
    for (auto &t : th) t.join();
    unsigned char tail[8];
    for (int b = 0; b < 4; b++) {
        tail[b] = (unsigned char)(crc >> (8 * b));
        tail[4 + b] = (unsigned char)((size & 0xFFFFFFFFu) >> (8 * b));
    }
    fwrite(tail, 1, 8, out);
    if (fclose(out) != 0 || failed) { fprintf(stderr, "pgzip1: failed\n"); return 2; }
    return 0;
}

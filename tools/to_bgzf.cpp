// to_bgzf.cpp -- measurement tool: rewrites a file as blocked gzip (BGZF: <= 64-KB members with a BC extra field + the empty
// end-of-file member), compressing blocks on several threads.   to_bgzf <in> <out> [threads=16] [level=1]
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static std::vector<unsigned char> block(const unsigned char *d, size_t n, int level) {
    std::vector<unsigned char> out(18 + compressBound(n) + 8);
    const unsigned char head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(out.data(), head, 16);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = const_cast<unsigned char *>(d);
    zs.avail_in = (uInt)n;
    zs.next_out = out.data() + 18;
    zs.avail_out = (uInt)(out.size() - 26);
    deflate(&zs, Z_FINISH);
    const size_t raw = zs.total_out, total = 18 + raw + 8;
    deflateEnd(&zs);
    out[16] = (unsigned char)((total - 1) & 0xFF);
    out[17] = (unsigned char)((total - 1) >> 8);
    const unsigned long crc = crc32(crc32(0L, Z_NULL, 0), d, (uInt)n);
    for (int i = 0; i < 4; ++i) out[18 + raw + i] = (unsigned char)(crc >> (8 * i)), out[22 + raw + i] = (unsigned char)((unsigned long)n >> (8 * i));
    out.resize(total);
    return out;
}
int main(int argc, char **argv) {
    if (argc < 3) return 1;
    const int threads = argc > 3 ? atoi(argv[3]) : 16, level = argc > 4 ? atoi(argv[4]) : 1;
    FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[2], "wb");
    if (!in || !out) return 2;
    const size_t B = 65280, per = 256;                               // blocks per thread per round
    std::vector<unsigned char> buf(B * per * threads);
    for (;;) {
        const size_t n = fread(buf.data(), 1, buf.size(), in);
        if (!n) break;
        const size_t nb = (n + B - 1) / B;
        std::vector<std::vector<unsigned char>> res(nb);
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t)
            th.emplace_back([&, t] { for (size_t i = t; i < nb; i += threads) res[i] = block(buf.data() + i * B, std::min(B, n - i * B), level); });
        for (auto &t : th) t.join();
        for (auto &r : res) fwrite(r.data(), 1, r.size(), out);
    }
    auto eof = block(nullptr, 0, level);
    fwrite(eof.data(), 1, eof.size(), out);
    fclose(out);
    return 0;
}

// Host-only driver (test infrastructure) for the device inflate's per-lane code and host logic: hast_amd/csrc/gz_core.h (what
// a GPU lane runs per chunk) and gz_chain.h (which chunks form the stream), with plain loops standing in for the kernels of
// gz_kernels.hip -- search + decode per chunk, windows chunk after chunk, marker translation, CRC-32 by slices combined with
// the GF(2) operators.  Decodes the file on the command line and writes the inflated bytes to stdout; exit 3 + message on a
// decoding error.  -c compressed bytes per chunk, -s chunks per segment (candidates reach the chain segment by segment, the
// "input on the device" grows with them), -r symbols of room per compressed byte of a chunk.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../hast_amd/csrc/gz_chain.h"
#include "../../hast_amd/csrc/gz_core.h"

using namespace hast::gz;

static std::vector<uint32_t> g_words;       // the file, zero padded
static uint64_t g_size = 0;

static void run_job(ChunkJob &j, uint64_t nbits, std::vector<uint16_t> &sym, std::vector<uint32_t> &tabs) {
    sym.assign((size_t)j.sym_cap + 8, 0);
    j.status = 0;
    j.n_out = 0;
    j.err_code = 0;
    if (j.flags & kJobKnown) j.start_bit = j.from_bit;
    else {
        bool found = false;
        const uint64_t lim = nbits > 192 ? nbits - 192 : 0;
        for (uint64_t bit = j.from_bit; bit < j.from_bit + j.search_to_lo && bit < lim; ++bit) {
            if (!candidate(bits_at(g_words.data(), bit), bits_at(g_words.data(), bit + 56))) continue;
            if (!header_parses(g_words.data(), nbits, bit, tabs.data() + kLitTabCap + kDistTabCap)) continue;
            j.start_bit = bit;
            found = true;
            break;
        }
        if (!found) {
            j.start_bit = j.end_bit = j.from_bit;
            return;
        }
    }
    decode_chunk(j, g_words.data(), nbits, tabs.data(), sym.data());
}

int main(int argc, char **argv) {
    size_t chunk = 32768, seg = 64;
    double room = 12;
    const char *path = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-c")) chunk = (size_t)atol(argv[++i]);
        else if (!strcmp(argv[i], "-s")) seg = (size_t)atol(argv[++i]);
        else if (!strcmp(argv[i], "-r")) room = atof(argv[++i]);
        else path = argv[i];
    }
    FILE *f = fopen(path, "rb");
    if (!f) return 2;
    std::vector<uint8_t> bytes;
    {
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
        fclose(f);
    }
    g_size = bytes.size();
    g_words.assign((g_size + 3) / 4 + 32, 0);
    if (g_size) memcpy(g_words.data(), bytes.data(), g_size);
    Chain chain;
    chain.begin(g_size, [&](uint64_t off, size_t n, uint8_t *dst, size_t *got) {
        const size_t m = off >= g_size ? 0 : (size_t)std::min<uint64_t>(n, g_size - off);
        if (m) memcpy(dst, bytes.data() + off, m);
        *got = m;
        return true;
    });
    std::vector<uint32_t> tabs(kTabWords);
    struct Buf { std::vector<uint16_t> sym; };
    struct Bufs : std::vector<Buf *> { ~Bufs() { for (Buf *b : *this) delete b; } } bufs;    // tag = index
    std::vector<Accepted> acc;
    std::vector<uint8_t> window(kWindow, 0), out;
    uint32_t crc = 0;
    uint64_t isize = 0;
    bool crc_started = false;
    // what the consumer side does with confirmed chunks: window, translate, CRC
    auto consume = [&]() -> bool {
        acc.clear();
        chain.take_confirmed(acc);
        for (const Accepted &a : acc) {
            Buf *b = bufs[(size_t)a.tag];
            const uint32_t n = a.job.n_out;
            if (a.out_off != out.size()) { fprintf(stderr, "offset mismatch\n"); return false; }
            // translate through the window in front of the chunk
            const size_t at = out.size();
            out.resize(at + n);
            for (uint32_t i = 0; i < n; ++i) {
                const uint16_t s = b->sym[i];
                if (s >= kMarker && a.no_history) { fprintf(stderr, "marker in a member's first chunk\n"); return false; }
                out[at + i] = s < kMarker ? (uint8_t)s : window[s - kMarker];
            }
            // the window behind it
            if (n >= kWindow) memcpy(window.data(), out.data() + at + n - kWindow, kWindow);
            else {
                memmove(window.data(), window.data() + n, kWindow - n);
                if (n) memcpy(window.data() + kWindow - n, out.data() + at, n);
            }
            // CRC-32 as the kernel does it: 256 slices of equal length counted from the END (only the first may be short), a
            // table-driven CRC per slice, combined pairwise with x^(8 s 2^k)
            uint32_t c = 0;
            if (n) {
                const uint32_t S = 256, s = (n + S - 1) / S;
                uint32_t part[256];
                uint32_t table[256];
                for (uint32_t i = 0; i < 256; ++i) table[i] = crc_table_entry(i);
                for (uint32_t l = 0; l < S; ++l) {
                    const int64_t hi = (int64_t)n - (int64_t)(S - 1 - l) * s, lo = hi - s;
                    uint32_t v = 0xFFFFFFFFu;
                    for (int64_t i = lo < 0 ? 0 : lo; i < hi; ++i) v = table[(v ^ out[at + (size_t)i]) & 0xFF] ^ (v >> 8);
                    part[l] = hi <= 0 ? 0u : v ^ 0xFFFFFFFFu;
                }
                uint32_t xk = crc_x2nmodp(s, 3);
                for (uint32_t step = 1; step < S; step <<= 1) {
                    for (uint32_t l = 0; l + step < S; l += 2 * step) part[l] = crc_combine_op(part[l], part[l + step], xk);
                    xk = crc_multmodp(xk, xk);
                }
                c = part[0];
            }
            crc = crc_started ? crc_combine_op(crc, c, crc_x2nmodp(n, 3)) : c;
            crc_started = crc_started || n;
            isize += n;
            delete b;
            bufs[(size_t)a.tag] = nullptr;
            if (a.member_end) {
                if (crc != a.want_crc) { fprintf(stderr, "gz: CRC-32 mismatch\n"); return false; }
                if ((uint32_t)isize != a.want_isize) { fprintf(stderr, "gz: length check (ISIZE) failed\n"); return false; }
                crc = 0;
                isize = 0;
                crc_started = false;
            }
        }
        return true;
    };
    const uint64_t first = chain.first_deflate_bit();
    const size_t n_chunks = first == ~0ull ? 0 : (size_t)((g_size + chunk - 1) / chunk);
    std::vector<Chain::Gap> gaps;
    for (size_t c0 = 0; c0 < n_chunks || c0 == 0; c0 += seg) {
        const size_t c1 = std::min(n_chunks, c0 + seg);
        // input present: this segment and the next one (or everything)
        const uint64_t input_bits = std::min<uint64_t>(g_size, (uint64_t)(c1 + seg) * chunk) * 8;
        const bool all_in = c1 >= n_chunks;
        std::vector<ChunkJob> jobs;
        for (size_t c = c0; c < c1; ++c) {
            ChunkJob j;
            memset(&j, 0, sizeof(j));
            const uint64_t nominal = (uint64_t)c * chunk * 8;
            if (nominal + chunk * 8 <= first) continue;                       // all header
            if (nominal <= first) { j.from_bit = first; j.flags = kJobKnown | kJobNoHistory; }
            else j.from_bit = nominal;
            j.stop_bit = (uint64_t)(c + 1) * chunk * 8;
            j.search_to_lo = (uint32_t)(j.stop_bit - j.from_bit);
            j.sym_cap = (uint32_t)(chunk * room) + 600;
            bufs.push_back(new Buf);
            j.sym_off = bufs.size() - 1;
            run_job(j, all_in ? g_size * 8 : input_bits, bufs.back()->sym, tabs);
            jobs.push_back(j);
        }
        chain.add_candidates(jobs.data(), jobs.size(), all_in);
        int rounds = 0;
        while (chain.plan(gaps, all_in ? g_size * 8 : input_bits)) {
            std::vector<ChunkJob> res;
            for (Chain::Gap &g : gaps) {
                ChunkJob j = g.job;
                j.sym_cap = (uint32_t)std::min<uint64_t>(g.want_syms, 1u << 27);
                bufs.push_back(new Buf);
                j.sym_off = bufs.size() - 1;
                run_job(j, all_in ? g_size * 8 : input_bits, bufs.back()->sym, tabs);
                res.push_back(j);
            }
            chain.gap_done(res.data(), res.size(), all_in ? g_size * 8 : input_bits);
            if (++rounds > 100000) { fprintf(stderr, "no progress\n"); return 3; }
            if (!consume()) return 3;
        }
        if (!consume()) return 3;
        if (chain.failed()) break;
        if (all_in) break;
    }
    if (chain.failed()) {
        if (!out.empty()) fwrite(out.data(), 1, out.size(), stdout);
        fprintf(stderr, "%s\n", chain.error().c_str());
        return 3;
    }
    if (!chain.finished()) {
        fprintf(stderr, "chain not finished\n");
        return 3;
    }
    if (!out.empty()) fwrite(out.data(), 1, out.size(), stdout);
    return 0;
}

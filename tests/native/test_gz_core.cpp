// Host-only driver (test infrastructure) for the device inflate's per-lane code and host logic: hast_amd/csrc/gz_core.h (what
// a GPU lane runs per chunk) and gz_chain.h (which chunks form the stream), with plain loops standing in for the kernels of
// gz_kernels.hip -- search + decode per chunk, windows chunk after chunk, marker translation, CRC-32 by slices combined with
// the GF(2) operators.  Decodes the file on the command line and writes the inflated bytes to stdout; exit 3 + message on a
// decoding error.  -c compressed bytes per chunk, -s chunks per segment (candidates reach the chain segment by segment, the
// "input on the device" grows with them), -r symbols of room per compressed byte of a chunk, -w Huffman blocks decoded the way
// k_gz_decode's wave does it (a token parsed at each of 64 bit offsets, the chain walked, rounds of at most 64 symbols).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include "../../hast_amd/csrc/gz_chain.h"
#include "../../hast_amd/csrc/gz_core.h"

using namespace hast::gz;

static std::vector<uint32_t> g_words;       // the file, zero padded
static uint64_t g_size = 0;

static bool g_wave = false;
// -S (with -w): what the wave's steps were made of, to stderr at the end -- steps, rounds, rounds with a match, rounds with a copy out of the
// symbol buffer itself (further back than kRing), tokens, matches, long matches (profiles/round6_gz_decode_latency_ab.txt quotes these)
static bool g_stats = false;
static struct { unsigned long long steps, full, rounds, rounds_match, rounds_far, syms, toks, matches, longs, longs_far; } g_st;      // -w: Huffman blocks decoded the way k_gz_decode does it (64 token parses per step, chain, rounds)

// ---- the symbol loop of k_gz_decode (gz_kernels.hip), lane by lane: every lane parses a token at its own bit offset (parse_token),
// the chain of real tokens is walked from offset 0, the chain's tokens are written out in rounds of at most 64 symbols (a token
// whose source reaches into its own round waits for the next one), sources come out of a ring of the last 512 symbols, out of the
// symbol buffer, or are markers.  Same arithmetic as the kernel, plain loops instead of lanes; the ring is modelled so that its
// reach rule is checked (what the kernel would read there must be what the symbol buffer holds).
static uint32_t decode_block_wave(const uint32_t *w, uint64_t nbits, uint64_t &pos, const uint32_t *lit, const uint32_t *dst, uint16_t *sym,
                                  uint32_t &n_out, uint32_t cap, bool no_history, uint32_t &err, uint16_t *ring) {
    constexpr uint32_t kIsLit = 0x40000000u, kIsMatch = 0x80000000u;
    uint32_t n2 = n_out;
    bool full = false;                                               // this step parses with the second-level tables too
    auto ring_at = [&](int64_t src) {
        const uint16_t v = ring[(uint32_t)src & (kRing - 1)];
        if (v != sym[src]) { fprintf(stderr, "ring reach rule broken at %lld\n", (long long)src); abort(); }
        return v;
    };
    for (;;) {
        if (pos >= nbits) return kStStarved;
        const uint64_t avail = nbits - pos;
        const uint32_t limit = avail < 64 ? (uint32_t)avail : 64u;
        Token tk[64];
        for (uint32_t lane = 0; lane < 64; ++lane) tk[lane] = full ? parse_token(bits_at(w, pos + lane), lit, dst) : parse_token_fast(bits_at(w, pos + lane), lit, dst);
        const bool was_full = full;
        g_st.steps++;
        g_st.full += full;
        full = false;
        uint32_t p = 0;
        bool again = false;
        for (;;) {
            uint64_t tokmask = 0;
            uint32_t stop = 0;
            while (p < limit) {
                const uint32_t t = tk[p].info;
                if (t >= 128) { stop = t; break; }
                tokmask |= 1ull << p;
                p += t;
            }
            if (!stop && p > avail) return kStStarved;               // the last token reads past the input that is there
            // where in the output every token of the chain starts
            uint32_t start[64], incl[64], total = 0;
            for (uint32_t lane = 0; lane < 64; ++lane) {
                start[lane] = total;
                if ((tokmask >> lane) & 1) total += tk[lane].olen;
                incl[lane] = total;
            }
            for (uint32_t lane = 0; lane < 64; ++lane)
                if (((tokmask >> lane) & 1) && tk[lane].dist && tk[lane].dist > n2 + start[lane] && (no_history || tk[lane].dist > kWindow)) { err = kErrTooFar; return kStError; }
            uint32_t base = 0;
            uint64_t rem = tokmask;
            while (rem) {
                uint32_t first_viol = 64;
                for (uint32_t lane = 0; lane < 64 && first_viol == 64; ++lane) {
                    if (!((rem >> lane) & 1)) continue;
                    const uint32_t need = tk[lane].dist ? (tk[lane].dist > tk[lane].olen ? tk[lane].dist - tk[lane].olen : 0u) : 0xFFFFu;
                    if (incl[lane] - base > 64 || need < start[lane] - base) first_viol = lane;
                }
                const uint64_t cur = first_viol < 64 ? rem & ((1ull << first_viol) - 1) : rem;
                const uint32_t nsym = (first_viol < 64 ? start[first_viol] : total) - base;
                if (!cur || !nsym || nsym > 64) { fprintf(stderr, "round logic broken\n"); abort(); }
                if (n2 + nsym > cap) return kStNoRoom;
                const uint32_t bstart = n2;
                const int64_t ring_lo = (int64_t)bstart - (int64_t)kRing;
                bool r_match = false, r_far = false;
                g_st.rounds++;
                g_st.syms += nsym;
                g_st.toks += (unsigned)__builtin_popcountll(cur);
                // the round's tokens at the lanes of their symbols: literals as themselves, a match at its first symbol
                uint32_t s_tok[64];
                for (uint32_t j = 0; j < 64; ++j) s_tok[j] = 0;
                for (uint32_t lane = 0; lane < 64; ++lane) {
                    if (!((cur >> lane) & 1)) continue;
                    const uint32_t st = start[lane] - base;
                    if (tk[lane].dist) g_st.matches++;
                    if (tk[lane].dist) s_tok[st] = kIsMatch | st | (tk[lane].dist << 14);
                    else {
                        s_tok[st] = kIsLit | (tk[lane].val & 0xFF);
                        if (tk[lane].olen == 2) s_tok[st + 1] = kIsLit | (tk[lane].val >> 8);
                    }
                }
                uint16_t outv[64];
                for (uint32_t j = 0; j < nsym; ++j) {
                    const uint32_t mine = s_tok[j];
                    if (mine & kIsLit) { outv[j] = (uint16_t)(mine & 0xFF); continue; }
                    uint32_t L = 64;
                    for (uint32_t q = 0; q <= j; ++q)
                        if (s_tok[q] & kIsMatch) L = q;                  // the last match that starts at or in front of j
                    if (L == 64) { fprintf(stderr, "a symbol without a token\n"); abort(); }
                    r_match = true;
                    const uint32_t tv = s_tok[L], off = tv & 63, dist = (tv >> 14) & 0xFFFF, k = j - off;
                    uint32_t kk = k;
                    if (k >= dist) {
                        const uint32_t q = (uint32_t)(((float)k + 0.5f) * (1.0f / (float)dist));
                        kk = k - q * dist;
                        if (kk != k % dist) { fprintf(stderr, "float modulo broken\n"); abort(); }
                    }
                    const int64_t src = (int64_t)bstart + (int64_t)off - (int64_t)dist + (int64_t)kk;
                    if (src >= (int64_t)bstart) { fprintf(stderr, "source inside its own round\n"); abort(); }
                    if (src < 0) outv[j] = (uint16_t)(kMarker + (uint32_t)((int64_t)kWindow + src));
                    else if (src < ring_lo) { outv[j] = sym[src]; r_far = true; }
                    else outv[j] = ring_at(src);
                }
                for (uint32_t j = 0; j < nsym; ++j) {
                    ring[(bstart + j) & (kRing - 1)] = outv[j];
                    sym[bstart + j] = outv[j];
                }
                base += nsym;
                n2 += nsym;
                rem &= ~cur;
                g_st.rounds_match += r_match;
                g_st.rounds_far += r_far;
            }
            if (!stop) break;
            const uint32_t kind = stop >> 7, tl = stop & 127;
            if (kind == kTokSlow) {
                // a code longer than the first-level table: the step that starts at this token looks into the second level as well
                if (was_full) { fprintf(stderr, "a slow token in a full parse\n"); abort(); }
                full = true;
                again = true;
                break;
            }
            if (kind == kTokErrLit || kind == kTokErrDist) {
                if (p + 48 > avail) return kStStarved;               // (read out of what is not there yet)
                err = kind == kTokErrLit ? kErrLitCode : kErrDistCode;
                return kStError;
            }
            if (p + tl > avail) return kStStarved;
            if (kind == kTokEob) {
                pos += p + tl;
                n_out = n2;
                return 0;
            }
            // a long match, on its own, 64 symbols a step
            const uint32_t len = tk[p].olen, distance = tk[p].dist;
            if (distance > n2 && (no_history || distance > kWindow)) { err = kErrTooFar; return kStError; }
            if (n2 + len > cap) return kStNoRoom;
            const bool near = ring_holds_long_match(distance, len);
            g_st.longs++;
            g_st.longs_far += !near;
            for (uint32_t k0 = 0; k0 < len; k0 += 64) {
                uint16_t outv[64];
                for (uint32_t lane = 0; lane < 64 && k0 + lane < len; ++lane) {
                    const uint32_t k = k0 + lane, kk = k < distance ? k : k % distance;
                    const int64_t src = (int64_t)n2 - (int64_t)distance + (int64_t)kk;
                    outv[lane] = src < 0 ? (uint16_t)(kMarker + (uint32_t)((int64_t)kWindow + src)) : near ? ring_at(src) : sym[src];
                }
                for (uint32_t lane = 0; lane < 64 && k0 + lane < len; ++lane) {
                    ring[(n2 + k0 + lane) & (kRing - 1)] = outv[lane];
                    sym[n2 + k0 + lane] = outv[lane];
                }
            }
            n2 += len;
            p += tl;
        }
        (void)again;
        pos += p;
    }
}

// decode_chunk (gz_core.h) with its Huffman blocks decoded by decode_block_wave
static void decode_chunk_wave(ChunkJob &job, const uint32_t *w, uint64_t nbits, uint32_t *tabs, uint16_t *sym) {
    Tables t = tables_at(tabs);
    HdrScratch scr;
    static uint16_t ring[kRing];
    const bool no_history = (job.flags & kJobNoHistory) != 0;
    uint64_t at = job.start_bit;
    uint32_t n = 0, status = kStFound, err = kErrNone;
    bool any = false;
    for (;;) {
        if (at >= job.stop_bit && (any || !(job.flags & kJobKnown))) {
            bool hidden = false;
            if (any && at + 3 <= nbits) hidden = (bits_at(w, at) & 7) != 4;
            if (!hidden) { status |= kStStop; break; }
        }
        if (at + 3 > nbits) { status |= kStStarved; break; }
        Bits in{w, nbits, 0, 0, 0};
        seek(in, at);
        refill(in);
        const uint32_t final = take(in, 1), type = take(in, 2);
        uint32_t n2 = n, bad = 0;
        if (type == 0) {
            const uint64_t byte = (pos(in) + 7) >> 3;
            if ((byte + 4) * 8 > nbits) { status |= kStStarved; break; }
            const uint8_t *bytes = reinterpret_cast<const uint8_t *>(w) + byte;
            const uint32_t len = bytes[0] | ((uint32_t)bytes[1] << 8), nlen = bytes[2] | ((uint32_t)bytes[3] << 8);
            if ((len ^ 0xFFFFu) != nlen) { status |= kStError; err = kErrStoredLen; break; }
            if ((byte + 4 + len) * 8 > nbits) { status |= kStStarved; break; }
            if (n + len + 4 > job.sym_cap) { status |= kStNoRoom; break; }
            for (uint32_t k = 0; k < len; ++k) {
                sym[n + k] = bytes[4 + k];
                ring[(n + k) & (kRing - 1)] = bytes[4 + k];
            }
            n2 = n + len;
            at = (byte + 4 + len) * 8;
        } else if (type == 3) {
            status |= kStError;
            err = kErrBlockType;
            break;
        } else {
            if (type == 1) fixed_tables(t, scr);
            else bad = read_dynamic(in, t, false, true, scr);
            if (bad) {
                if (overran(in)) status |= kStStarved;
                else { status |= kStError; err = bad; }
                break;
            }
            if (overran(in)) { status |= kStStarved; break; }
            {   // two literals per first-level entry where both codes fit (every entry worked out before any is replaced)
                uint32_t paired[1u << kLitRoot];
                for (uint32_t i = 0; i < (1u << kLitRoot); ++i) paired[i] = pair_entry(t.lit, i);
                memcpy(t.lit, paired, sizeof(paired));
            }
            uint64_t p = pos(in);
            const uint32_t rc = decode_block_wave(w, nbits, p, t.lit, t.dist, sym, n2, job.sym_cap, no_history, err, ring);
            if (rc) { status |= rc; break; }
            at = p;
        }
        n = n2;
        any = true;
        if (final) { status |= kStFinal; break; }
    }
    if (!any) status |= kStNoBlock;
    job.end_bit = at;
    job.n_out = n;
    job.status = status;
    job.err_code = (status & kStError) ? err : kErrNone;
}

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
            // (the search kernel's register-only version of the strict parse must give the same verdict at every candidate)
            uint8_t pre8[128];
            const bool ok = header_parses(g_words.data(), nbits, bit, tabs.data() + kLitTabCap + kDistTabCap);
            if (ok != header_parses8(g_words.data(), nbits, bit, pre8)) { fprintf(stderr, "header_parses8 differs at bit %llu\n", (unsigned long long)bit); abort(); }
            if (!ok) continue;
            j.start_bit = bit;
            found = true;
            break;
        }
        if (!found) {
            j.start_bit = j.end_bit = j.from_bit;
            return;
        }
    }
    if (g_wave) decode_chunk_wave(j, g_words.data(), nbits, tabs.data(), sym.data());
    else decode_chunk(j, g_words.data(), nbits, tabs.data(), sym.data());
}

// -e BYTES: the walk as gz_api.cpp drives it for a file that goes round a RING on the device: the chain is eager (set_eager), and what a
// ring needs of it is checked after every segment -- the chain's end lies at most BYTES in front of the segment's end unless it waits
// for input that is not there (a ring can keep a few passes and a piece of the file, not more); "the end of the input" is never
// announced before the last segment, as the ring's uploader never gets there before the chain has moved on
static uint64_t g_ring_lag = 0;
// -v RING:PIECE: every job sees the file as a job on the device sees a ring of RING bytes with pieces of PIECE (gz_api.cpp job_view): from
// the lap its first bit lies in to a piece behind that lap's end (ChunkJob.limit_bits) -- what lies behind is "not there" for it, also
// when the whole input is (a job that runs on into a member's final block can get there: the chain must follow it up, not call it the
// end of the input)
static uint64_t g_view_ring = 0, g_view_piece = 0;
static void view(ChunkJob &j) {
    j.limit_bits = 0;
    if (g_view_ring) j.limit_bits = (((j.from_bit >> 3) / g_view_ring + 1) * g_view_ring + g_view_piece) * 8;
}
static uint64_t seen(const ChunkJob &j, uint64_t input_bits) { return j.limit_bits && j.limit_bits < input_bits ? j.limit_bits : input_bits; }

int main(int argc, char **argv) {
    size_t chunk = 32768, seg = 64;
    double room = 12;
    long fuzz = 0;
    const char *path = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-c")) chunk = (size_t)atol(argv[++i]);
        else if (!strcmp(argv[i], "-s")) seg = (size_t)atol(argv[++i]);
        else if (!strcmp(argv[i], "-r")) room = atof(argv[++i]);
        else if (!strcmp(argv[i], "-w")) g_wave = true;
        else if (!strcmp(argv[i], "-S")) g_stats = true;
        else if (!strcmp(argv[i], "-f")) fuzz = atol(argv[++i]);
        else if (!strcmp(argv[i], "-e")) g_ring_lag = (uint64_t)atol(argv[++i]);     // a ring on the device: see below
        else if (!strcmp(argv[i], "-v")) {
            const char *a = argv[++i], *c = strchr(a, ':');
            g_view_ring = (uint64_t)atol(a);
            g_view_piece = c ? (uint64_t)atol(c + 1) : 65536;
        }
        else path = argv[i];
    }
    if (fuzz) {
        // header_parses8 against header_parses on random bits (most fail at the counts or at the code-length code; a few per cent get into
        // the length loop) and on random bits behind a code-length code that IS complete
        uint64_t x = 0x9E3779B97F4A7C15ull, agree_true = 0;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        std::vector<uint32_t> buf(64);
        std::vector<uint32_t> tabs(kTabWords);
        for (long it = 0; it < fuzz; ++it) {
            for (uint32_t &v : buf) v = (uint32_t)rnd();
            if (it & 1) {
                // 3 type bits, HLIT, HDIST, HCLEN = 15 (19 lengths), then a complete code: lengths from a random full binary tree
                uint8_t lens[19] = {0};
                int n = 1; lens[0] = 0;
                uint8_t depth[19] = {0};
                while (n < 19 && (rnd() % 8)) {             // split a random leaf that is not at depth 7
                    const int k = (int)(rnd() % (uint64_t)n);
                    if (depth[k] >= 7) continue;
                    depth[k]++;
                    depth[n++] = depth[k];
                }
                if (n == 1) depth[0] = 1, depth[n++] = 1;
                int perm[19];
                for (int i = 0; i < 19; ++i) perm[i] = i;
                for (int i = 18; i > 0; --i) std::swap(perm[i], perm[rnd() % (uint64_t)(i + 1)]);
                for (int i = 0; i < n; ++i) lens[perm[i]] = depth[i];
                const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint64_t hdr = 4 | ((rnd() % 30) << 3) | ((rnd() % 30) << 8) | (15ull << 13);
                int at = 17;
                std::vector<uint8_t> bits(17 + 57);
                for (int i = 0; i < 17; ++i) bits[i] = (hdr >> i) & 1;
                for (int i = 0; i < 19; ++i)
                    for (int b = 0; b < 3; ++b) bits[at++] = (lens[order[i]] >> b) & 1;
                for (int i = 0; i < at; ++i) {
                    buf[i >> 5] &= ~(1u << (i & 31));
                    buf[i >> 5] |= (uint32_t)bits[i] << (i & 31);
                }
            }
            uint8_t pre8[128];
            const uint64_t nb = (it % 7 == 0) ? 64 * 8 + (rnd() % 600) : 64 * 32 - 192;
            const bool a = header_parses(buf.data(), nb, 0, tabs.data() + kLitTabCap + kDistTabCap), b = header_parses8(buf.data(), nb, 0, pre8);
            if (a != b) { fprintf(stderr, "header_parses8 differs (trial %ld: %d vs %d)\n", it, (int)a, (int)b); return 3; }
            agree_true += a;
        }
        printf("%llu accepted\n", (unsigned long long)agree_true);
        return 0;
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
    chain.set_eager(g_ring_lag != 0);
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
            view(j);
            run_job(j, seen(j, all_in ? g_size * 8 : input_bits), bufs.back()->sym, tabs);
            if ((j.status & kStStarved) && j.limit_bits && j.limit_bits < (all_in ? g_size * 8 : input_bits)) fprintf(stderr, "view ended: a pass's job from bit %llu\n", (unsigned long long)j.start_bit);
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
                view(j);
                run_job(j, seen(j, all_in ? g_size * 8 : input_bits), bufs.back()->sym, tabs);
                if ((j.status & kStStarved) && j.limit_bits && j.limit_bits < (all_in ? g_size * 8 : input_bits)) fprintf(stderr, "view ended: a follow-up job from bit %llu\n", (unsigned long long)j.start_bit);
                res.push_back(j);
            }
            chain.gap_done(res.data(), res.size(), all_in ? g_size * 8 : input_bits);
            if (++rounds > 100000) { fprintf(stderr, "no progress\n"); return 3; }
            if (!consume()) return 3;
        }
        if (!consume()) return 3;
        if (chain.failed()) break;
        if (all_in) break;
        if (g_ring_lag) {
            const uint64_t end_byte = chain.proven_end_bit() >> 3, seg_end = (uint64_t)c1 * chunk;
            if (end_byte + g_ring_lag < seg_end) {
                fprintf(stderr, "ring: the chain stands at byte %llu, %llu behind the end of the segment (allowed: %llu)\n", (unsigned long long)end_byte,
                        (unsigned long long)(seg_end - end_byte), (unsigned long long)g_ring_lag);
                return 4;
            }
        }
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
    if (g_stats && g_wave)
        fprintf(stderr, "wave steps %llu (with second-level tables %llu) rounds %llu with_match %llu with_far_copy %llu symbols_in_rounds %llu tokens %llu matches %llu long_matches %llu long_far %llu\n",
                g_st.steps, g_st.full, g_st.rounds, g_st.rounds_match, g_st.rounds_far, g_st.syms, g_st.toks, g_st.matches, g_st.longs, g_st.longs_far);
    return 0;
}

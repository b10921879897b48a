// gz_kernels.hip -- gfx950 device code of the gzip inflate that feeds the FASTQ framer (reference: gzstream.h:47,
// classify.cpp:245-254 read each .gz input through ONE zlib stream on the host).  gz_core.h holds what a lane does with a
// chunk of the compressed bytes; here are the launches around it:
//   k_gz_search     a wave per chunk: 64 bit positions per step through the cheap header test (gz_core.h candidate), survivors
//                   parsed in full by their own lane, lowest position first
//   k_gz_decode     a WAVE per chunk: blocks -> 16-bit symbols (literal, or marker "byte i of the 32 KB in front of this chunk").
//                   Serial by nature -- Huffman codes have no boundaries anyone wrote down -- so the parallelism is chunks (a
//                   614-MB .fq.gz is 19 000 chunks of 32 KB) plus, inside a chunk, the lanes that copy a match together
//   k_gz_maps / k_gz_carry / k_gz_window   the 32 KB behind every accepted chunk (each depends on the one in front of it: maps
//                   composed per group of 64 chunks in LDS, the groups chained by one workgroup, then everything at once)
//   k_gz_crc        per chunk: CRC-32 of its bytes by 256 slices, combined with GF(2) operators (gz_core.h crc_*)
//   k_gz_translate  symbols -> bytes (markers through the window in front of the chunk) straight into the caller's buffer,
//                   e.g. a block buffer of the FASTQ framer
// Bound: k_gz_decode by LATENCY per symbol (an LDS table look-up depends on the bits the look-up in front of it consumed), hence by
// waves in flight (16 per CU: 8 KB of tables + ring each, 128 VGPRs); the other kernels stream (2 B read + 1 B written per byte of output).
#include <hip/hip_runtime.h>

#include "gz_core.h"
#include "gz_device.h"

namespace hast {
namespace gz {

// ---- search: the first position in [from_bit, from_bit + search_to_lo) that parses as a non-final dynamic block header -----
// A wave per chunk, 64 bit positions per step, three sieves: (1) the 17 header bits every lane tests for itself (type bits 100b,
// HLIT, HDIST in range: one position in 9 passes); (2) the survivors are queued (LDS, in position order) and, 64 at a time, tested
// for a COMPLETE code-length code (one in ~100 passes); (3) what is left is queued again and parsed in full 64 at a time, a lane
// per candidate; the lowest position that parses is the chunk's start.
__global__ void __launch_bounds__(64) k_gz_search(ChunkJob *jobs, uint32_t n_jobs, const uint32_t *w, uint64_t nbits) {
    __shared__ uint32_t s_q[1024];                                            // positions (relative to from_bit) that passed sieve 1
    __shared__ uint32_t s_q3[128];                                            // ... that passed sieve 2 (position order)
    __shared__ uint32_t s_pre[64 * kPreTabCap];                               // the strict parse's code-length table, one per lane
    const uint32_t j = blockIdx.x, lane = threadIdx.x;
    if (j >= n_jobs) return;
    ChunkJob &job = jobs[j];
    if (job.flags & kJobKnown) {
        if (lane == 0) {
            job.start_bit = job.from_bit;
            job.status = kStFound;
        }
        return;
    }
    const uint64_t from = job.from_bit;
    const uint64_t lim = nbits > 192 ? nbits - 192 : 0;                       // (bits_at reads 12 bytes, the strict parse more: the buffer is padded)
    uint64_t to = from + job.search_to_lo;
    if (to > lim) to = lim;
    uint64_t found = ~0ull;
    uint32_t qh = 0, qn = 0, q3n = 0;                                         // sieve-1 queue: s_q[qh, qh + qn)
    // sieve 3 over the first `cnt` entries of s_q3, every lane its own candidate: a strict parse decodes ~290 code lengths one after
    // the other (~50 us on a lane)
    auto parse_batch = [&](uint32_t cnt) {
        bool ok = false;
        if (lane < cnt) ok = header_parses(w, nbits, from + s_q3[lane], s_pre + lane * kPreTabCap);
        const unsigned long long m = __ballot(ok);
        if (m) found = from + s_q3[__builtin_ctzll(m)];                       // the lowest position that parses
        const uint32_t rest = q3n - cnt;
        uint32_t v = 0;
        if (lane < rest) v = s_q3[cnt + lane];
        __syncthreads();
        if (lane < rest) s_q3[lane] = v;
        q3n = rest;
        __syncthreads();
    };
    // sieve 2 over the first `cnt` entries of the sieve-1 queue (position order); what passes is queued for sieve 3
    auto drain = [&](uint32_t cnt) {
        bool c = false;
        uint32_t rel = 0;
        if (lane < cnt) {
            rel = s_q[qh + lane];
            const uint64_t bit = from + rel;
            c = candidate(bits_at(w, bit), bits_at(w, bit + 56));
        }
        const unsigned long long mask = __ballot(c);
        if (mask) {
            const uint32_t at = q3n + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (c) s_q3[at] = rel;
            q3n += (uint32_t)__popcll(mask);
        }
        qh += cnt;
        qn -= cnt;
        __syncthreads();
        if (q3n >= 64) parse_batch(64);
    };
    // sieve 1, a WORD per lane and step (2048 bit positions a step: one pair of coalesced loads instead of 32 -- the wave used to take
    // 64 positions a step and waited for the same two words 32 times over): the 13 header bits at each of the word's 32 offsets
    const uint64_t w_first = from >> 5, w_end = (to + 31) >> 5;
    for (uint64_t wb = w_first; wb < w_end && found == ~0ull; wb += 64) {
        const uint64_t wi = wb + lane;
        uint32_t m = 0;
        if (wi < w_end) {
            const uint64_t x = ((uint64_t)w[wi + 1] << 32) | w[wi];
#pragma unroll
            for (int o = 0; o < 32; ++o) {
                const uint32_t v = (uint32_t)(x >> o);
                const bool pre = (v & 7) == 4 && ((v >> 3) & 31) <= 29 && ((v >> 8) & 31) <= 29;
                m |= (uint32_t)pre << o;
            }
            // positions in front of `from` and at or behind `to`
            const uint64_t b0 = wi * 32;
            if (b0 < from) m &= ~0u << (uint32_t)(from - b0);
            if (b0 + 32 > to) m &= to > b0 ? ~0u >> (uint32_t)(b0 + 32 - to) : 0u;
        }
        // the survivors of all lanes into the queue, in position order: lane by lane, offset by offset
        uint32_t cnt = (uint32_t)__popc(m), incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= (uint32_t)d) incl += t;
        }
        const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
        if (total) {
            // the remainder of the queue to its front (fewer than 64 entries), then this step's survivors behind it
            uint32_t keep = 0;
            if (lane < qn) keep = s_q[qh + lane];
            __syncthreads();
            if (lane < qn) s_q[lane] = keep;
            qh = 0;
            uint32_t at = qn + incl - cnt;
            const uint32_t rel0 = (uint32_t)(wi * 32 - from);                 // (wraps for positions in front of `from`: those bits are masked off)
            while (m) {
                const uint32_t o = (uint32_t)__builtin_ctz(m);
                m &= m - 1;
                s_q[at++] = rel0 + o;
            }
            qn += total;
            __syncthreads();
            while (qn >= 64 && found == ~0ull) drain(64);
        }
    }
    while (qn && found == ~0ull) drain(qn < 64 ? qn : 64);
    while (q3n && found == ~0ull) parse_batch(q3n < 64 ? q3n : 64);
    if (lane == 0) {
        job.start_bit = found == ~0ull ? from : found;
        job.end_bit = job.start_bit;
        job.n_out = 0;
        job.err_code = 0;
        job.status = found == ~0ull ? 0u : kStFound;
    }
}

// ---- decode: a WAVE per chunk -----------------------------------------------------------------------------------------------------
// Huffman decoding is serial (no code boundary is known before the code in front of it is decoded), so one symbol at a time --
// but by the whole wave in lockstep: every lane carries the same bit buffer and reads the same table entry (a broadcast read of
// the tables in LDS, built per block by lane 0 with the code of gz_core.h; the bit buffer and everything derived from it live in
// SGPRs), which keeps the control flow uniform, and the part that IS parallel runs on all lanes: the symbols of a block are gathered
// as tokens (literals, matches) and written out 64 at a time, every lane one symbol -- out of a ring of the chunk's last 512
// symbols in LDS when its source reaches back less than that, out of the symbol buffer otherwise (one global load for all far
// symbols of a batch, past the L1, behind one wait for the stores in flight).  Output symbols: literal byte, or kMarker + i = "byte i
// of the 32 KB in front of this chunk" for a copy that reaches in front of the chunk (gz_core.h).
struct WBits {                     // wave-uniform bit input.  The next 64 words of the stream sit in a VGPR, one per lane (`win`,
    const uint32_t *w;             // word wbase + lane), the 64 behind them in `nxt` (loaded when `win` is taken into use, a few hundred
    uint64_t nwords;               // symbols before anyone needs them): a refill is a v_readlane, no memory access is on the
    uint64_t bb;                   // path from one symbol to the next
    uint64_t wp;                   // index of the next word to take
    uint64_t wbase;                // index of win's lane 0
    uint32_t bc, win, nxt;
};
__device__ __forceinline__ uint32_t wload(const WBits &b, uint64_t i) { return i < b.nwords ? b.w[i] : 0u; }
__device__ __forceinline__ uint32_t wword(WBits &b) {            // the word at wp
    uint64_t idx = b.wp - b.wbase;
    if (idx >= 64) {                                              // (uniform)
        b.win = b.nxt;
        b.wbase += 64;
        b.nxt = wload(b, b.wbase + 64 + threadIdx.x);
        idx -= 64;
    }
    return (uint32_t)__builtin_amdgcn_readlane((int)b.win, (int)idx);
}
__device__ __forceinline__ void wseek(WBits &b, uint64_t bit) {
    const uint64_t wi = bit >> 5;
    const uint32_t sh = (uint32_t)(bit & 31);
    b.wbase = wi;
    b.win = wload(b, wi + threadIdx.x);
    b.nxt = wload(b, wi + 64 + threadIdx.x);
    b.wp = wi;
    b.bb = (uint64_t)wword(b) >> sh;
    b.bc = 32 - sh;
    b.wp = wi + 1;
}
__device__ __forceinline__ void wrefill(WBits &b) {
    if (b.bc <= 32) {
        b.bb |= (uint64_t)wword(b) << b.bc;
        b.bc += 32;
        b.wp++;
    }
}
__device__ __forceinline__ uint64_t wpos(const WBits &b) { return b.wp * 32 - b.bc; }
__device__ __forceinline__ void sym_store(uint16_t *p, uint16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ uint16_t sym_load_far(const uint16_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // (past the L1)

// 4 waves per SIMD (128 VGPRs; the compiler took 180 = 2 waves per SIMD = 8 chunks per CU, and a chunk's time is latency -- far match
// copies out of the symbol buffer, ~20 ms whatever else runs -- so chunks in flight are throughput): 14 per CU, the LDS's limit.
// syms: the base of the symbol memory the jobs' buffers lie in (job.sym_off is an absolute address / 2: the buffer is reached as
// syms + offset, so that the compiler knows it for GLOBAL memory -- through a generic pointer the stores are FLAT instructions,
// which count as LDS operations too, and every table look-up then waits for the symbol stores in flight: measured 2000 cycles per
// symbol instead of ~300)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_gz_decode(ChunkJob *jobs, uint32_t n_jobs, const uint32_t *__restrict__ w, uint64_t nbits, uint16_t *__restrict__ syms) {
    __shared__ uint32_t s_tab[kTabWords];
    __shared__ uint32_t s_hdr[4];                                             // lane 0's header parse: error, bit position behind the header (lo, hi)
    // the chunk's last kRing symbols: a match that reaches back less than that (in FASTQ most: the few records in front) is
    // copied out of LDS -- a global load per match would put ~1 us of latency on the path of every symbol behind it
    // (512 symbols: with the batched copies a far source costs little, and 8.3 KB of LDS a wave instead of 11.3 is 16 waves per CU
    // instead of 14 -- read phase -3.5 % in an A/B on one box, profiles/round4_ab_gz_ring512.txt)
    constexpr uint32_t kRing = 512, kRingReach = kRing - 320;
    __shared__ uint16_t s_ring[kRing];
    const uint32_t j = blockIdx.x, lane = threadIdx.x;
    if (j >= n_jobs) return;
    ChunkJob job = jobs[j];
    if (!(job.status & kStFound)) {
        if (lane == 0) {
            jobs[j].n_out = 0;
            jobs[j].end_bit = job.start_bit;
        }
        return;
    }
    uint16_t *const sym = syms + (ptrdiff_t)((long long)job.sym_off - (long long)(reinterpret_cast<uintptr_t>(syms) >> 1));
    const uint32_t cap = job.sym_cap;
    const bool no_history = (job.flags & kJobNoHistory) != 0;
    const uint32_t *const lit = s_tab, *const dst = s_tab + kLitTabCap;
    constexpr uint32_t LM = (1u << kLitRoot) - 1, DM = (1u << kDistRoot) - 1;
    WBits in{w, ((nbits + 31) >> 5) + 2, 0, 0, 0, 0, 0, 0};
    uint64_t at = job.start_bit;
    uint32_t n = 0, status = kStFound, err = kErrNone;
    bool any = false;
    for (;;) {
        if (at >= job.stop_bit && (any || !(job.flags & kJobKnown))) {
            // a boundary at or behind the stop: the chunk ends here -- unless what follows is a block no search can find (stored,
            // fixed, or final: the chunk behind this one starts at the first NON-FINAL DYNAMIC header), which this chunk takes too
            bool hidden = false;
            if (any && at + 3 <= nbits) {
                const uint32_t h = (uint32_t)(bits_at(w, at) & 7);
                hidden = h != 4;
            }
            if (!hidden) { status |= kStStop; break; }
        }
        if (at + 3 > nbits) { status |= kStStarved; break; }
        wseek(in, at);
        wrefill(in);
        const uint32_t final = (uint32_t)(in.bb & 1), type = (uint32_t)((in.bb >> 1) & 3);
        in.bb >>= 3;
        in.bc -= 3;
        uint32_t n2 = n, rc = 0;
        if (type == 0) {
            const uint64_t byte = (wpos(in) + 7) >> 3;
            if ((byte + 4) * 8 > nbits) { status |= kStStarved; break; }
            const uint8_t *bytes = reinterpret_cast<const uint8_t *>(w) + byte;
            const uint32_t len = bytes[0] | ((uint32_t)bytes[1] << 8), nlen = bytes[2] | ((uint32_t)bytes[3] << 8);
            if ((len ^ 0xFFFFu) != nlen) { status |= kStError; err = kErrStoredLen; break; }
            if ((byte + 4 + len) * 8 > nbits) { status |= kStStarved; break; }
            if (n + len + 4 > cap) { status |= kStNoRoom; break; }
            for (uint32_t k0 = 0; k0 < len; k0 += 64) {                     // (64 consecutive symbols per step: ring slots of a step are distinct)
                const uint32_t k = k0 + lane;
                if (k < len) {
                    const uint16_t v = bytes[4 + k];
                    s_ring[(n + k) & (kRing - 1)] = v;
                    sym_store(sym + n + k, v);
                }
            }
            __syncthreads();
            n2 = n + len;
            at = (byte + 4 + len) * 8;
        } else if (type == 3) {
            status |= kStError;
            err = kErrBlockType;
            break;
        } else {
            // tables for this block: lane 0 parses the header / builds them in LDS (gz_core.h), the wave waits
            __syncthreads();                                                  // (everyone is done with the previous block's tables)
            if (lane == 0) {
                Tables t = tables_at(s_tab);
                uint32_t bad = 0;
                uint64_t behind = wpos(in);
                if (type == 1) fixed_tables(t);
                else {
                    Bits hb{w, nbits, 0, 0, 0};
                    seek(hb, behind);
                    bad = read_dynamic(hb, t, false);
                    if (bad && overran(hb)) bad = 0x80000000u;                // an "error" read out of the padding: the block is not all here
                    if (!bad && overran(hb)) bad = 0x80000000u;
                    behind = pos(hb);
                }
                s_hdr[0] = bad;
                s_hdr[1] = (uint32_t)behind;
                s_hdr[2] = (uint32_t)(behind >> 32);
            }
            __syncthreads();
            const uint32_t bad = s_hdr[0];
            if (bad) {
                if (bad == 0x80000000u) status |= kStStarved;
                else { status |= kStError; err = bad; }
                break;
            }
            wseek(in, (uint64_t)s_hdr[1] | ((uint64_t)s_hdr[2] << 32));
            // ---- TWO literals per look-up where both codes fit the 9 root bits: bit 8 of a literal's entry says "a second literal in
            // bits 24..31", the length is that of both codes.  The bases of a FASTQ record are literals of ~2.2 bits each (half of a file's
            // bytes), quality values mostly fit in pairs too: a look-up -- the LDS round trip every symbol waits for -- then yields two
            // symbols.  All 64 lanes, 8 root entries each: every entry is read (with the entry of the bits behind its code) before any is
            // written (LDS operations of a wave execute in order).
            {
                uint32_t pe[(1u << kLitRoot) / 64];
#pragma unroll
                for (uint32_t q = 0; q < (1u << kLitRoot) / 64; ++q) {
                    const uint32_t idx = q * 64 + lane, e1 = lit[idx], l1 = e1 & 0xFF;
                    uint32_t ne = e1;
                    if ((e1 & (kLit | kSub)) == kLit && l1 > 0 && l1 < (uint32_t)kLitRoot) {
                        const uint32_t e2 = lit[idx >> l1], l2 = e2 & 0xFF;     // (the bits behind the first code, zeros above them: an
                        // entry whose code is no longer than the bits that are really there does not depend on those zeros)
                        if ((e2 & (kLit | kSub)) == kLit && l2 > 0 && l1 + l2 <= (uint32_t)kLitRoot)
                            ne = kLit | (1u << 8) | (l1 + l2) | (e1 & 0x00FF0000u) | ((e2 & 0x00FF0000u) << 8);
                    }
                    pe[q] = ne;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (uint32_t q = 0; q < (1u << kLitRoot) / 64; ++q) s_tab[q * 64 + lane] = pe[q];
                __syncthreads();
            }
            // ---- the block's symbols ----
            // Symbols are decoded into a BATCH of tokens (literal(s), or match: length, distance) and written out 64 at a time: token i
            // sits in lane i of two VGPRs, a 64-bit mask marks where in the batch each token starts, and a flush gives
            // every lane ONE symbol of the batch -- its token by a bit count below it in the mask, its source in the ring, in the symbol
            // buffer (a far match: one global load for all far symbols of the batch, behind ONE wait for the stores in flight) or in front
            // of the chunk (a marker).  A match whose source reaches into the batch itself (a run, a copy of the symbols just decoded)
            // closes the batch first.  Before: every match waited for its own copy -- random DNA is coded as short matches at random
            // distances in the 32-KB window, ~12 000 a chunk, each a round trip to HBM (the symbol buffers of the chunks in flight are
            // GBs: no cache holds them) -- and every literal was a global store of its own.
            uint32_t tokA = 0, tokB = 0;                 // lane i: token i = offset | length << 6 | distance << 16 (0: literals), literal bytes
            uint32_t nb_tok = 0, nb_sym = 0;             // (wave-uniform)
            uint64_t smask = 0;
            uint32_t bstart = n2;                        // position of the batch's first symbol
            auto flush = [&]() {
                if (!nb_sym) return;
                const bool act = lane < nb_sym;
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(smask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)smask, 0u));
                const uint32_t ti = act ? below + (uint32_t)((smask >> lane) & 1) - 1 : 0u;      // (bit 0 is set: the first token starts the batch)
                const uint32_t A = (uint32_t)__shfl((int)tokA, (int)ti, 64), B = (uint32_t)__shfl((int)tokB, (int)ti, 64);
                const uint32_t off = A & 63, dist = A >> 16, k = lane - off;
                const int64_t ring_lo = (int64_t)bstart + 64 - (int64_t)kRing;                  // positions the ring still holds while this batch is written
                int64_t src = 0;
                if (dist) src = (int64_t)bstart + (int64_t)off - (int64_t)dist + (int64_t)(k < dist ? k : k % dist);
                const bool far = act && dist && src >= 0 && src < ring_lo;
                // a copy out of the symbol buffer itself reads what this wave stored a while ago: the stores must have reached the L2
                // (s_waitcnt: every store of this wave acknowledged) and the loads go there (sc1, past the L1).  No cache maintenance:
                // an agent-scope fence here writes back and invalidates the L2 -- measured: every kernel on the GPU 10 x slower
                if (__ballot(far)) __builtin_amdgcn_s_waitcnt(0);
                uint16_t v = 0;
                if (act) {
                    if (!dist) v = (uint16_t)((B >> (8 * k)) & 0xFF);
                    else if (src < 0) v = (uint16_t)(kMarker + (uint32_t)((int64_t)kWindow + src));
                    else if (far) v = sym_load_far(sym + src);
                    else v = s_ring[(uint32_t)src & (kRing - 1)];
                }
                // (every lane has read before any lane writes: LDS operations of a wave execute in order)
                if (act) {
                    s_ring[(bstart + lane) & (kRing - 1)] = v;
                    sym_store(sym + bstart + lane, v);
                }
                bstart += nb_sym;
                nb_tok = nb_sym = 0;
                smask = 0;
            };
            for (;;) {
                if (n2 + 260 > cap) { rc = kStNoRoom; break; }
                if (in.bc <= 32) {
                    // (the end of the input is looked for when a word is taken, not per symbol: the buffer is padded with zeros, and the
                    // block's end checks once more)
                    if (wpos(in) > nbits) { rc = kStStarved; break; }
                    wrefill(in);
                }
                // (every lane reads the same entry: it is a wave-uniform value, and saying so keeps the tests on it scalar)
                uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)lit[in.bb & LM]);
                if (e & kSub) {
                    in.bb >>= kLitRoot;
                    in.bc -= kLitRoot;
                    e = (uint32_t)__builtin_amdgcn_readfirstlane((int)lit[(e >> 16) + (uint32_t)(in.bb & ((1u << ((e >> 8) & 31)) - 1))]);
                }
                in.bb >>= (e & 0xFF);
                in.bc -= (e & 0xFF);
                if (e & kLit) {
                    const uint32_t cnt = 1 + ((e >> 8) & 1);                          // a second literal rides along
                    if (nb_sym + cnt > 64) flush();                                   // (a token holds at least one symbol: never more than 64 tokens)
                    tokA = lane == nb_tok ? (nb_sym | (cnt << 6)) : tokA;
                    tokB = lane == nb_tok ? (e >> 16) : tokB;
                    smask |= 1ull << nb_sym;
                    ++nb_tok;
                    nb_sym += cnt;
                    n2 += cnt;
                    continue;
                }
                if ((e & 0xFF) == 0) { err = kErrLitCode; rc = kStError; break; }
                if (e & kEob) break;
                const uint32_t leb = (e >> 8) & 31;
                const uint32_t len = (e >> 16) + (uint32_t)(in.bb & ((1u << leb) - 1));
                in.bb >>= leb;
                in.bc -= leb;
                wrefill(in);
                uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)dst[in.bb & DM]);
                if (d & kSub) {
                    in.bb >>= kDistRoot;
                    in.bc -= kDistRoot;
                    d = (uint32_t)__builtin_amdgcn_readfirstlane((int)dst[(d >> 16) + (uint32_t)(in.bb & ((1u << ((d >> 8) & 31)) - 1))]);
                }
                if ((d & 0xFF) == 0) { err = kErrDistCode; rc = kStError; break; }
                in.bb >>= (d & 0xFF);
                in.bc -= (d & 0xFF);
                const uint32_t deb = (d >> 8) & 31;
                const uint32_t distance = (d >> 16) + (uint32_t)(in.bb & ((1u << deb) - 1));
                in.bb >>= deb;
                in.bc -= deb;
                if (distance > n2 && (no_history || distance > kWindow)) { err = kErrTooFar; rc = kStError; break; }
                if (len <= 64) {
                    // its source must lie in front of the batch (what the batch holds is not written yet)
                    const int64_t src_end = (int64_t)n2 - (int64_t)distance + (int64_t)(len < distance ? len : distance);
                    if (src_end > (int64_t)bstart || nb_sym + len > 64) flush();
                    tokA = lane == nb_tok ? (nb_sym | (len << 6) | (distance << 16)) : tokA;
                    smask |= 1ull << nb_sym;
                    ++nb_tok;
                    nb_sym += len;
                    n2 += len;
                    continue;
                }
                // a long match, on its own: symbol n2 + k comes from n2 + k - distance, or -- periodic -- from the first `distance` of
                // them; what lies in front of the chunk is a marker.  64 symbols per step: the lanes of a step read before any of them
                // writes, and a later step reads what the earlier ones wrote.
                flush();
                const bool near = distance <= kRingReach;
                if (!near) __builtin_amdgcn_s_waitcnt(0);
                for (uint32_t k0 = 0; k0 < len; k0 += 64) {
                    const uint32_t k = k0 + lane;
                    if (k < len) {
                        const uint32_t kk = k < distance ? k : k % distance;
                        const int64_t src = (int64_t)n2 - (int64_t)distance + (int64_t)kk;
                        const uint16_t v = src < 0 ? (uint16_t)(kMarker + (uint32_t)((int64_t)kWindow + src))
                                                   : near ? s_ring[(uint32_t)src & (kRing - 1)] : sym_load_far(sym + src);
                        s_ring[(n2 + k) & (kRing - 1)] = v;
                        sym_store(sym + n2 + k, v);
                    }
                }
                n2 += len;
                bstart = n2;
            }
            if (!rc) flush();
            if (rc) {
                status |= rc;
                break;
            }
            if (wpos(in) > nbits) { status |= kStStarved; break; }
            at = wpos(in);
        }
        n = n2;
        any = true;
        if (final) { status |= kStFinal; break; }
    }
    if (!any) status |= kStNoBlock;
    if (lane == 0) {
        jobs[j].end_bit = at;
        jobs[j].n_out = n;
        jobs[j].status = status;
        jobs[j].err_code = (status & kStError) ? err : kErrNone;
    }
}

// ---- windows: W[c] = the kWindow bytes of the stream behind accepted chunk c (W[-1] = `carry`, what the batch in front left) ----
// W[c] depends on W[c-1] wherever chunk c's last 32 KB hold markers (or are fewer than 32 KB): a chain through the whole batch.
// A "map" of 32768 16-bit entries says for every byte of a window what it is: a literal, or kMarker + i = byte i of an EARLIER
// window; maps compose, so the chain is cut into groups of kGroup chunks:
//   k_gz_maps    a workgroup per group walks its chunks in order, composing in LDS: maps[c] = W[c] in terms of the window in front
//                of the GROUP
//   k_gz_carry   one workgroup walks the groups: the window in front of each group (gwin[g]) from the last map of the group before
//   k_gz_window  every chunk at once: W[c] = maps[c] resolved through gwin[group of c]
constexpr uint32_t kGroup = 64;
__global__ void __launch_bounds__(1024) k_gz_maps(const AccDev *acc, uint32_t n_acc, uint16_t *maps) {
    extern __shared__ uint16_t s_map[];                                       // 2 x kWindow
    const uint32_t tid = threadIdx.x;
    const uint32_t c0 = blockIdx.x * kGroup, c1 = c0 + kGroup < n_acc ? c0 + kGroup : n_acc;
    uint16_t *cur = s_map, *nxt = s_map + kWindow;
    for (uint32_t k = tid; k < kWindow; k += 1024) cur[k] = (uint16_t)(kMarker + k);      // the identity: byte k of the window in front of the group
    __syncthreads();
    for (uint32_t c = c0; c < c1; ++c) {
        const AccDev a = acc[c];
        const uint16_t *s = a.sym;
        const uint32_t own = a.n_out < kWindow ? a.n_out : kWindow;           // symbols of this chunk in its window
        uint16_t *out = maps + (size_t)c * kWindow;
        for (uint32_t k = tid; k < kWindow; k += 1024) {
            uint16_t v;
            if (k < kWindow - own) v = a.no_history ? (uint16_t)0 : cur[k + own];      // the window in front, shifted (nothing in front of a member)
            else {
                v = s[a.n_out - own + (k - (kWindow - own))];
                if (v >= kMarker) v = cur[v - kMarker];
            }
            nxt[k] = v;
            out[k] = v;
        }
        __syncthreads();
        uint16_t *t = cur;
        cur = nxt;
        nxt = t;
    }
}
__global__ void __launch_bounds__(1024) k_gz_carry(const uint16_t *maps, uint32_t n_acc, const uint8_t *carry, uint8_t *gwin) {
    extern __shared__ uint8_t s_win[];                                        // 2 x kWindow
    const uint32_t tid = threadIdx.x;
    uint8_t *cur = s_win, *nxt = s_win + kWindow;
    for (uint32_t k = tid; k < kWindow; k += 1024) cur[k] = carry[k];
    __syncthreads();
    const uint32_t n_groups = (n_acc + kGroup - 1) / kGroup;
    for (uint32_t g = 0; g < n_groups; ++g) {
        uint8_t *out = gwin + (size_t)g * kWindow;
        for (uint32_t k = tid; k < kWindow; k += 1024) out[k] = cur[k];
        const uint32_t last = (g + 1) * kGroup < n_acc ? (g + 1) * kGroup - 1 : n_acc - 1;
        const uint16_t *m = maps + (size_t)last * kWindow;
        for (uint32_t k = tid; k < kWindow; k += 1024) {
            const uint16_t v = m[k];
            nxt[k] = v < kMarker ? (uint8_t)v : cur[v - kMarker];
        }
        __syncthreads();
        uint8_t *t = cur;
        cur = nxt;
        nxt = t;
    }
}
__global__ void __launch_bounds__(256) k_gz_window(const uint16_t *maps, uint32_t n_acc, const uint8_t *gwin, uint8_t *windows) {
    const uint32_t c = blockIdx.x;
    if (c >= n_acc) return;
    const uint16_t *m = maps + (size_t)c * kWindow;
    const uint8_t *gw = gwin + (size_t)(c / kGroup) * kWindow;
    uint8_t *out = windows + (size_t)c * kWindow;
    for (uint32_t k = threadIdx.x; k < kWindow; k += 256) {
        const uint16_t v = m[k];
        out[k] = v < kMarker ? (uint8_t)v : gw[v - kMarker];
    }
}

// ---- CRC-32 of every chunk's bytes: 256 slices of equal length counted from the END (only the first may be short, and a short
// FIRST operand needs no length), a table-driven CRC per slice, then combined pairwise with x^(8 s 2^k) ---------------------------
__global__ void __launch_bounds__(256) k_gz_crc(const AccDev *acc, uint32_t n_acc, const uint8_t *windows, const uint8_t *carry, uint32_t *crc_out) {
    __shared__ uint32_t s_tab[256], s_part[256];
    const uint32_t c = blockIdx.x, tid = threadIdx.x;
    if (c >= n_acc) return;
    const AccDev a = acc[c];
    s_tab[tid] = crc_table_entry(tid);
    __syncthreads();
    const uint32_t n = a.n_out;
    if (n == 0) {
        if (tid == 0) crc_out[c] = 0;
        return;
    }
    const uint8_t *prev = c ? windows + (size_t)(c - 1) * kWindow : carry;
    const uint16_t *s = a.sym;
    const uint32_t slice = (n + 255) / 256;
    const long long hi = (long long)n - (long long)(255 - tid) * slice, lo = hi - slice;
    uint32_t v = 0xFFFFFFFFu;
    auto step = [&](uint32_t x) {
        const uint32_t b = x < kMarker ? x : prev[x - kMarker];
        v = s_tab[(v ^ b) & 0xFF] ^ (v >> 8);
    };
    // a lane's slice is contiguous: 8 symbols per 16-byte load (the buffer is 16-byte aligned), so that a fetched line is used up
    // by the next few loads of the same lane -- symbol by symbol, the 64 lanes of a load touch 64 lines for 2 bytes each and evict
    // one another's lines before they come back for the rest
    long long i = lo < 0 ? 0 : lo;
    for (; i < hi && (i & 7); ++i) step(s[i]);
    for (; i + 8 <= hi; i += 8) {
        const uint4 q = *reinterpret_cast<const uint4 *>(s + i);
        step(q.x & 0xFFFFu); step(q.x >> 16); step(q.y & 0xFFFFu); step(q.y >> 16);
        step(q.z & 0xFFFFu); step(q.z >> 16); step(q.w & 0xFFFFu); step(q.w >> 16);
    }
    for (; i < hi; ++i) step(s[i]);
    s_part[tid] = hi <= 0 ? 0u : v ^ 0xFFFFFFFFu;
    uint32_t xk = crc_x2nmodp(slice, 3);
    __syncthreads();
    for (uint32_t step = 1; step < 256; step <<= 1) {
        if ((tid & (2 * step - 1)) == 0) s_part[tid] = crc_combine_op(s_part[tid], s_part[tid + step], xk);
        xk = crc_multmodp(xk, xk);
        __syncthreads();
    }
    if (tid == 0) crc_out[c] = s_part[0];
}

// ---- translate: the bytes [o_lo, o_hi) of the inflated stream, taken from the chunks acc[c_first ..], to dst[0 ..) ---------------
constexpr uint32_t kTile = 8192;                                                // symbols per workgroup
__global__ void __launch_bounds__(256) k_gz_translate(const AccDev *acc, uint32_t c_first, const uint8_t *windows, const uint8_t *carry,
                                                      uint64_t o_lo, uint64_t o_hi, uint8_t *dst) {
    const uint32_t c = c_first + blockIdx.y;
    const AccDev a = acc[c];
    const uint32_t t0 = blockIdx.x * kTile;
    if (t0 >= a.n_out) return;
    const uint8_t *prev = c ? windows + (size_t)(c - 1) * kWindow : carry;
    const uint16_t *s = a.sym;
    const uint32_t t1 = t0 + kTile < a.n_out ? t0 + kTile : a.n_out;
    auto byte_of = [&](uint32_t x) -> uint32_t { return x < kMarker ? x : prev[x - kMarker]; };
    // 8 symbols per lane and step: one 16-byte load (the symbol buffer is 16-byte aligned, tiles start at multiples of 8)
    for (uint32_t i = t0 + 8 * threadIdx.x; i < t1; i += 8 * 256) {
        const uint64_t o = a.out_off + i;
        if (o >= o_hi || o + 8 <= o_lo) continue;
        if (i + 8 <= t1 && o >= o_lo && o + 8 <= o_hi) {
            const uint4 q = *reinterpret_cast<const uint4 *>(s + i);
            const uint32_t lo4 = byte_of(q.x & 0xFFFFu) | byte_of(q.x >> 16) << 8 | byte_of(q.y & 0xFFFFu) << 16 | byte_of(q.y >> 16) << 24;
            const uint32_t hi4 = byte_of(q.z & 0xFFFFu) | byte_of(q.z >> 16) << 8 | byte_of(q.w & 0xFFFFu) << 16 | byte_of(q.w >> 16) << 24;
            uint8_t *d = dst + (o - o_lo);
            if ((reinterpret_cast<uintptr_t>(d) & 3) == 0) {
                reinterpret_cast<uint32_t *>(d)[0] = lo4;
                reinterpret_cast<uint32_t *>(d)[1] = hi4;
            } else {
                for (int k = 0; k < 4; ++k) d[k] = (uint8_t)(lo4 >> (8 * k));
                for (int k = 0; k < 4; ++k) d[4 + k] = (uint8_t)(hi4 >> (8 * k));
            }
        } else {
            for (uint32_t k = i; k < i + 8 && k < t1; ++k) {
                const uint64_t ok = a.out_off + k;
                if (ok >= o_lo && ok < o_hi) dst[ok - o_lo] = (uint8_t)byte_of(s[k]);
            }
        }
    }
}

// ---- launchers -------------------------------------------------------------------------------------------------------------------
hipError_t launch_search(ChunkJob *d_jobs, uint32_t n, const uint32_t *d_w, uint64_t nbits, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_gz_search, dim3(n), dim3(64), 0, s, d_jobs, n, d_w, nbits);
    return hipGetLastError();
}
hipError_t launch_decode(ChunkJob *d_jobs, uint32_t n, const uint32_t *d_w, uint64_t nbits, uint16_t *d_syms, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_gz_decode, dim3(n), dim3(64), 0, s, d_jobs, n, d_w, nbits, d_syms);
    return hipGetLastError();
}
size_t windows_scratch_bytes(uint32_t n) { return (size_t)n * kWindow * sizeof(uint16_t) + (size_t)((n + kGroup - 1) / kGroup) * kWindow; }
hipError_t launch_windows(const AccDev *d_acc, uint32_t n, uint8_t *d_windows, const uint8_t *d_carry, void *d_scratch, hipStream_t s) {
    if (!n) return hipSuccess;
    uint16_t *maps = reinterpret_cast<uint16_t *>(d_scratch);
    uint8_t *gwin = reinterpret_cast<uint8_t *>(maps + (size_t)n * kWindow);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_gz_maps), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kWindow * (int)sizeof(uint16_t));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_gz_carry), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (int)kWindow);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const uint32_t n_groups = (n + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(k_gz_maps, dim3(n_groups), dim3(1024), 2 * kWindow * sizeof(uint16_t), s, d_acc, n, maps);
    hipLaunchKernelGGL(k_gz_carry, dim3(1), dim3(1024), 2 * kWindow, s, maps, n, d_carry, gwin);
    hipLaunchKernelGGL(k_gz_window, dim3(n), dim3(256), 0, s, maps, n, gwin, d_windows);
    return hipGetLastError();
}
hipError_t launch_crc(const AccDev *d_acc, uint32_t n, const uint8_t *d_windows, const uint8_t *d_carry, uint32_t *d_crc, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_gz_crc, dim3(n), dim3(256), 0, s, d_acc, n, d_windows, d_carry, d_crc);
    return hipGetLastError();
}
hipError_t launch_translate(const AccDev *d_acc, uint32_t c_first, uint32_t n_chunks, uint32_t max_syms, const uint8_t *d_windows,
                            const uint8_t *d_carry, uint64_t o_lo, uint64_t o_hi, uint8_t *d_dst, hipStream_t s) {
    if (!n_chunks || !max_syms || o_hi <= o_lo) return hipSuccess;
    // (grid.y <= 65535: the caller hands over at most that many chunks per launch)
    hipLaunchKernelGGL(k_gz_translate, dim3((max_syms + kTile - 1) / kTile, n_chunks), dim3(256), 0, s, d_acc, c_first, d_windows, d_carry, o_lo, o_hi, d_dst);
    return hipGetLastError();
}

}  // namespace gz
}  // namespace hast

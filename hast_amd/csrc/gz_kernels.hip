// gz_kernels.hip -- gfx950 device code of the gzip inflate that feeds the FASTQ framer (reference: gzstream.h:47,
// classify.cpp:245-254 read each .gz input through ONE zlib stream on the host).  gz_core.h holds what a lane does with a
// chunk of the compressed bytes; here are the launches around it:
//   k_gz_search     a wave per chunk: 64 bit positions per step through the cheap header test (gz_core.h candidate), survivors
//                   parsed in full by their own lane, lowest position first
//   k_gz_decode     a WAVE per chunk: blocks -> 16-bit symbols (literal, or marker "byte i of the 32 KB in front of this chunk").
//                   Huffman codes have no boundaries anyone wrote down, so a token is parsed at each of 64 consecutive bit offsets
//                   at once (a lane each) and the chain of real tokens is walked with one scalar read per token; the chain's
//                   symbols are written by all lanes together
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
__global__ void __launch_bounds__(64) k_gz_search(ChunkJob *jobs, uint32_t n_jobs, const uint32_t *w_file, uint64_t nbits_up) {
    __shared__ uint32_t s_q[1024];                                            // positions (relative to from_bit) that passed sieve 1
    __shared__ uint32_t s_q3[128];                                            // ... that passed sieve 2 (position order)
    // the strict parse's code-length table, one per lane: 128 bytes (gz_core.h header_parses8 -- with header_parses' 512 bytes of
    // table and 1.4 KB of scratch arrays per lane a CU held ONE wave per SIMD, and the search is a chain of dependent instructions
    // per lane: profiles/round5 -- 13 KB of LDS a wave now, three waves per SIMD)
    __shared__ uint8_t s_pre8[64 * kPre8Stride];
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
    // (a ring of the file on the device: this job's view of it, ChunkJob)
    const uint32_t *const w = w_file - (ptrdiff_t)job.in_adj_words;
    const uint64_t nbits = job.limit_bits && job.limit_bits < nbits_up ? job.limit_bits : nbits_up;
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
        if (lane < cnt) ok = header_parses8(w, nbits, from + s_q3[lane], s_pre8 + lane * kPre8Stride);
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
// Huffman codes have no boundaries anyone wrote down: where token i + 1 starts is known once token i is decoded.  Round 4's kernel
// therefore decoded ONE symbol at a time, the whole wave in lockstep on a uniform bit buffer -- 80 instructions and two dependent LDS
// round trips per token, 63 of 64 lanes idle except in copies (profiles/round4_gz_kernels_pmc.txt: 13 instructions per byte of
// output).  This one decodes SPECULATIVELY AT EVERY BIT OFFSET: a step takes the 64 bit offsets pos .. pos + 63, lane i parses the
// token that would start at pos + i (gz_core.h parse_token: literal / length look-up, extra bits, distance look-up, extra bits -- its
// own 64 bits of the stream, the block's tables in LDS), and then the chain of REAL tokens is walked from offset 0 with one scalar
// read per token (v_readlane of the token's bit count).  What the chain's tokens stand for is written out by all lanes together:
// a prefix sum over the chain's tokens gives every token its place in the output, the tokens are handed to the lanes of their first
// symbols through LDS, every lane finds the token it belongs to (the last start at or in front of it) and fetches its symbol -- a
// literal, or a copy out of a ring of the chunk's last 512 symbols in LDS, out of the symbol buffer past the L1 (one global load for
// all far symbols of a round, behind one wait for the stores in flight), or a marker kMarker + i = "byte i of the 32 KB in front of
// this chunk" (gz_core.h).  A round holds at most 64 symbols; a match whose source reaches into its own round opens the next one;
// matches longer than 64 symbols, the end-of-block code and invalid codes stop the chain and are dealt with on their own.  On FASTQ
// a step yields 10 - 30 tokens (bases are 2-bit literals).  tests/native/test_gz_core.cpp -w runs the same steps with plain loops on
// the CPU against zlib.
struct WIn {                       // the compressed words around the read position: a ring of 128 words in LDS (s_in[word & 127]),
    const uint32_t *w;             // always the 128 words from rb on (rb a multiple of 64, the position inside the first 64); the 64
    uint64_t nwords;               // words behind them wait in a VGPR (nxt), loaded ~20 steps before they are put into the ring
    uint64_t rb;
    uint32_t nxt;
};
__device__ __forceinline__ uint32_t wload(const WIn &b, uint64_t i) { return i < b.nwords ? b.w[i] : 0u; }
__device__ __forceinline__ void win_at(WIn &b, uint32_t *s_in, uint64_t pos) {      // (uniform) the ring covers `pos` and the 4 words behind it
    const uint64_t wi = pos >> 5;
    const uint32_t lane = threadIdx.x;
    if (wi < b.rb || wi >= b.rb + 128) {
        b.rb = wi & ~63ull;
        __builtin_amdgcn_wave_barrier();
        s_in[(uint32_t)(b.rb + lane) & 127] = wload(b, b.rb + lane);
        s_in[(uint32_t)(b.rb + 64 + lane) & 127] = wload(b, b.rb + 64 + lane);
        b.nxt = wload(b, b.rb + 128 + lane);
        __builtin_amdgcn_wave_barrier();
    } else if (wi >= b.rb + 64) {
        __builtin_amdgcn_wave_barrier();
        s_in[(uint32_t)(b.rb + lane) & 127] = b.nxt;                          // (the words rb + 128 .. rb + 191 take the place of rb .. rb + 63)
        b.rb += 64;
        b.nxt = wload(b, b.rb + 128 + lane);
        __builtin_amdgcn_wave_barrier();
    }
}
__device__ __forceinline__ uint64_t win_bits(const WIn &b, const uint32_t *s_in, uint64_t pos) {      // the 64 bits at pos + lane
    const uint32_t rel = (uint32_t)(pos - b.rb * 32) + threadIdx.x, wi = (uint32_t)b.rb + (rel >> 5), sh = rel & 31;
    const uint32_t w0 = s_in[wi & 127], w1 = s_in[(wi + 1) & 127], w2 = s_in[(wi + 2) & 127];
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ void sym_store(uint16_t *p, uint16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ uint16_t sym_load_far(const uint16_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // (past the L1)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {              // inclusive prefix sum over the 64 lanes (DPP: rows, then row 15 -> next row, then 31 -> upper half)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
    return x;
}

// 5 waves per SIMD (96 VGPRs): a chunk's time is latency, so chunks in flight are throughput; 20 per CU (7.9 KB of LDS each: tables
// at zlib's own bounds, a ring of 512 symbols).
// syms: the base of the symbol memory the jobs' buffers lie in (job.sym_off is an absolute address / 2: the buffer is reached as
// syms + offset, so that the compiler knows it for GLOBAL memory -- through a generic pointer the stores are FLAT instructions,
// which count as LDS operations too, and every table look-up then waits for the symbol stores in flight: measured 2000 cycles per
// symbol instead of ~300)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) k_gz_decode(ChunkJob *jobs, uint32_t n_jobs, const uint32_t *__restrict__ w_file, uint64_t nbits_up, uint16_t *__restrict__ syms, uint32_t *pool_cursor, uint32_t pool_slots) {
    __shared__ uint32_t s_tab[kLitTabCap + kDistTabCap];
    // two lives: while lane 0 parses a block's header, the code-length code's table (kPreTabCap words) and what the parse indexes by
    // values it has just read (HdrScratch: no scratch memory); in the symbol loop, the compressed words around the read position
    // (WIn: 128 words) and a round's tokens at the lanes of their symbols (64 words)
    constexpr uint32_t kScrWords = (sizeof(HdrScratch) + 3) / 4;
    __shared__ uint32_t s_misc[kPreTabCap + (kScrWords > 64 ? kScrWords : 64)];
    __shared__ uint32_t s_hdr[4];                                             // lane 0's header parse: error, bit position behind the header (lo, hi)
    uint32_t *const s_in = s_misc, *const s_tok = s_misc + kPreTabCap;
    // the chunk's last kRing symbols: a match that reaches back less than that (in FASTQ most: the record or two in front) is
    // copied out of LDS -- a global load per match would put ~1 us of latency on the path of every symbol behind it
    __shared__ uint16_t s_ring[kRing];                                        // (gz_core.h: 512 symbols, and what a copy may take out of them)
    constexpr uint32_t kIsLit = 0x40000000u, kIsMatch = 0x80000000u;
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
    if (job.flags & kJobPoolSlot) {                                           // (wave-uniform) the next free slot of the pass's pool
        uint32_t slot = 0;
        if (lane == 0) slot = atomicAdd(pool_cursor, 1u);
        slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        if (slot >= pool_slots) {                                             // none left: a follow-up job's business (gz_chain.h)
            if (lane == 0) {
                jobs[j].n_out = 0;
                jobs[j].end_bit = job.start_bit;
                jobs[j].status = kStFound | kStNoRoom | kStNoBlock;
                jobs[j].err_code = kErrNone;
            }
            return;
        }
        job.sym_off += (uint64_t)slot * job.sym_cap;
        if (lane == 0) jobs[j].sym_off = job.sym_off;
    }
    const uint32_t *__restrict__ const w = w_file - (ptrdiff_t)job.in_adj_words;           // (a ring of the file on the device: this job's view, ChunkJob)
    const uint64_t nbits = job.limit_bits && job.limit_bits < nbits_up ? job.limit_bits : nbits_up;
    uint16_t *const sym = syms + (ptrdiff_t)((long long)job.sym_off - (long long)(reinterpret_cast<uintptr_t>(syms) >> 1));
    const uint32_t cap = job.sym_cap;
    const bool no_history = (job.flags & kJobNoHistory) != 0;
    const uint32_t *const lit = s_tab, *const dst = s_tab + kLitTabCap;
    constexpr uint64_t kFarAway = ~0ull >> 1;
    WIn in{w, ((nbits + 31) >> 5) + 2, kFarAway, 0};                         // (rb far away: the first win_at loads the ring)
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
        const uint32_t hdr3 = (uint32_t)(bits_at(w, at) & 7);
        const uint32_t final = hdr3 & 1, type = hdr3 >> 1;
        uint32_t n2 = n, rc = 0;
        if (type == 0) {
            const uint64_t byte = (at + 3 + 7) >> 3;
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
            __syncthreads();                                                  // (everyone is done with the previous block's tables, words and tokens)
            if (lane == 0) {
                Tables t{s_tab, s_tab + kLitTabCap, s_misc};
                HdrScratch &scr = *reinterpret_cast<HdrScratch *>(s_misc + kPreTabCap);
                uint32_t bad = 0;
                uint64_t behind = at + 3;
                if (type == 1) fixed_tables(t, scr);
                else {
                    Bits hb{w, nbits, 0, 0, 0};
                    seek(hb, behind);
                    bad = read_dynamic(hb, t, false, true, scr);
                    if (overran(hb)) bad = 0x80000000u;                       // (also an "error" read out of the padding: the block is not all here)
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
            uint64_t pos = (uint64_t)s_hdr[1] | ((uint64_t)s_hdr[2] << 32);
            in.rb = kFarAway;                                                 // (the header parse has used the words' place)
            {   // two literals per first-level entry where both codes fit (gz_core.h pair_entry): all 64 lanes, 8 entries each, every
                // entry worked out before any is replaced (LDS operations of a wave execute in order)
                uint32_t pe[(1u << kLitRoot) / 64];
                uint32_t l2 = lane;
                asm volatile("" : "+v"(l2));                                  // (or the eight q * 64 + lane live in registers for the whole kernel: hoisted, one of them spilled)
#pragma unroll
                for (uint32_t q = 0; q < (1u << kLitRoot) / 64; ++q) pe[q] = pair_entry(lit, q * 64 + l2);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (uint32_t q = 0; q < (1u << kLitRoot) / 64; ++q) s_tab[q * 64 + l2] = pe[q];
                __syncthreads();
            }
            // ---- the block's symbols, a step = the 64 bit offsets from pos on ----
            bool block_done = false, full = false;
            while (!block_done && !rc) {
                if (pos >= nbits) { rc = kStStarved; break; }
                const uint64_t avail = nbits - pos;
                const uint32_t limit = avail < 64 ? (uint32_t)avail : 64u;
                win_at(in, s_in, pos);
                const uint64_t bits = win_bits(in, s_in, pos);
                // (first-level tables only; a step that starts at a code longer than those looks into the second level too)
                const Token tk = full ? parse_token(bits, lit, dst) : parse_token_fast(bits, lit, dst);
                full = false;
                const uint32_t need = tk.dist ? (tk.dist > tk.olen ? tk.dist - tk.olen : 0u) : 0xFFFFu;      // symbols of the round that may stand in front of it
                uint32_t p = 0;
                for (;;) {                                                    // the chain, from one stop to the next
                    // while (p < limit) { t = info of lane p; if (t >= 128) { stop = t; break; }  tokmask |= 1 << p;  p += t; } -- by hand: six
                    // scalar instructions per token (the compiler turns the early exit into selects: sixteen)
                    uint64_t tokmask = 0;
                    uint32_t stop;
                    asm volatile("s_mov_b32 %[t], 0\n\t"
                                 "s_cmp_lt_u32 %[p], %[limit]\n\t"
                                 "s_cbranch_scc0 2f\n"
                                 "1:\n\t"
                                 "v_readlane_b32 %[t], %[info], %[p]\n\t"
                                 "s_cmpk_ge_u32 %[t], 0x80\n\t"
                                 "s_cbranch_scc1 2f\n\t"
                                 "s_bitset1_b64 %[mask], %[p]\n\t"
                                 "s_add_u32 %[p], %[p], %[t]\n\t"
                                 "s_cmp_lt_u32 %[p], %[limit]\n\t"
                                 "s_cbranch_scc1 1b\n\t"
                                 "s_mov_b32 %[t], 0\n"
                                 "2:\n"
                                 : [t] "=&s"(stop), [mask] "+s"(tokmask), [p] "+s"(p)
                                 : [info] "v"(tk.info), [limit] "s"(limit)
                                 : "scc");
                    if (!stop && p > avail) { rc = kStStarved; break; }       // the last token reads past the input that is there
                    if (tokmask) {
                        const bool is_tok = (tokmask >> lane) & 1;
                        const uint32_t incl = wave_incl_scan(is_tok ? tk.olen : 0u), start = incl - (is_tok ? tk.olen : 0u);
                        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                        if (__ballot(is_tok && tk.dist > n2 + start && (no_history || tk.dist > kWindow))) { err = kErrTooFar; rc = kStError; break; }
                        uint32_t base = 0;
                        uint64_t rem = tokmask;
                        while (rem) {
                            const bool in_rem = (rem >> lane) & 1;
                            const unsigned long long vm = __ballot(in_rem && (incl - base > 64 || need < start - base));
                            uint64_t cur = rem;
                            uint32_t nsym = total - base;
                            if (vm) {
                                const uint32_t fv = (uint32_t)__builtin_ctzll(vm);
                                cur = rem & ((1ull << fv) - 1);
                                nsym = (uint32_t)__builtin_amdgcn_readlane((int)start, (int)fv) - base;
                            }
                            if (n2 + nsym > cap) { rc = kStNoRoom; break; }
                            // the round's tokens at the lanes of their symbols: literals as themselves, a match at its first symbol (the
                            // lanes behind it find it: the last match that starts at or in front of them)
                            __builtin_amdgcn_wave_barrier();
                            s_tok[lane] = 0;
                            if ((cur >> lane) & 1) {
                                const uint32_t st = start - base;
                                if (tk.dist) s_tok[st] = kIsMatch | st | (tk.dist << 14);
                                else {
                                    s_tok[st] = kIsLit | (tk.val & 0xFF);
                                    if (tk.olen == 2) s_tok[st + 1] = kIsLit | (tk.val >> 8);
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                            const uint32_t mine = s_tok[lane];
                            const unsigned long long mm = __ballot((mine & kIsMatch) != 0);
                            const bool act = lane < nsym;
                            const uint32_t bstart = n2;
                            uint16_t v = (uint16_t)(mine & 0xFF);
                            if (mm) {
                                const unsigned long long upto = mm & (~0ull >> (63 - lane));
                                const uint32_t leader = 63u - (uint32_t)__builtin_clzll(upto | 1ull);
                                const uint32_t tv = (uint32_t)__shfl((int)mine, (int)leader, 64);
                                const bool is_m = act && !(mine & kIsLit);
                                const uint32_t off = tv & 63, dist = (tv >> 14) & 0xFFFF, k = lane - off;
                                const int64_t ring_lo = (int64_t)bstart - (int64_t)kRing;                   // what the ring holds (the round reads before it writes)
                                uint32_t kk = k;
                                if (k >= dist) {                               // a run: symbol k repeats symbol k mod dist (k < 64: exact in float)
                                    const uint32_t q = (uint32_t)(((float)k + 0.5f) * __frcp_rn((float)dist));
                                    kk = k - q * dist;
                                }
                                const int64_t src = (int64_t)bstart + (int64_t)off - (int64_t)dist + (int64_t)kk;
                                const bool far = is_m && src >= 0 && src < ring_lo;
                                // a copy out of the symbol buffer itself reads what this wave stored a while ago: the stores must have reached the L2
                                // (s_waitcnt: every store of this wave acknowledged) and the loads go there (sc1, past the L1).  No cache maintenance:
                                // an agent-scope fence here writes back and invalidates the L2 -- measured: every kernel on the GPU 10 x slower
                                if (__ballot(far)) __builtin_amdgcn_s_waitcnt(0);
                                if (is_m) {
                                    if (src < 0) v = (uint16_t)(kMarker + (uint32_t)((int64_t)kWindow + src));
                                    else if (far) v = sym_load_far(sym + src);
                                    else v = s_ring[(uint32_t)src & (kRing - 1)];
                                }
                            }
                            // (every lane has read before any lane writes: LDS operations of a wave execute in order)
                            __builtin_amdgcn_wave_barrier();
                            if (act) {
                                s_ring[(bstart + lane) & (kRing - 1)] = v;
                                sym_store(sym + bstart + lane, v);
                            }
                            base += nsym;
                            n2 += nsym;
                            rem &= ~cur;
                        }
                        if (rc) break;
                    }
                    if (!stop) break;
                    const uint32_t kind = stop >> 7, tl = stop & 127;
                    if (kind == kTokSlow) {                                   // the step that starts here parses in full
                        full = true;
                        break;
                    }
                    if (kind == kTokErrLit || kind == kTokErrDist) {
                        if (p + 48 > avail) rc = kStStarved;                  // (read out of what is not there yet)
                        else { err = kind == kTokErrLit ? kErrLitCode : kErrDistCode; rc = kStError; }
                        break;
                    }
                    if (p + tl > avail) { rc = kStStarved; break; }
                    if (kind == kTokEob) {
                        pos += p + tl;
                        block_done = true;
                        break;
                    }
                    // a long match, on its own: symbol n2 + k comes from n2 + k - distance, or -- periodic -- from the first `distance` of
                    // them; what lies in front of the chunk is a marker.  64 symbols per step: the lanes of a step read before any of them
                    // writes, and a later step reads what the earlier ones wrote.
                    const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)tk.olen, (int)p), distance = (uint32_t)__builtin_amdgcn_readlane((int)tk.dist, (int)p);
                    if (distance > n2 && (no_history || distance > kWindow)) { err = kErrTooFar; rc = kStError; break; }
                    if (n2 + len > cap) { rc = kStNoRoom; break; }
                    const bool near = ring_holds_long_match(distance, len);
                    if (!near) __builtin_amdgcn_s_waitcnt(0);
                    for (uint32_t k0 = 0; k0 < len; k0 += 64) {
                        const uint32_t k = k0 + lane;
                        uint16_t v = 0;
                        if (k < len) {
                            const uint32_t kk = k < distance ? k : k % distance;
                            const int64_t src = (int64_t)n2 - (int64_t)distance + (int64_t)kk;
                            v = src < 0 ? (uint16_t)(kMarker + (uint32_t)((int64_t)kWindow + src))
                                        : near ? s_ring[(uint32_t)src & (kRing - 1)] : sym_load_far(sym + src);
                        }
                        __builtin_amdgcn_wave_barrier();
                        if (k < len) {
                            s_ring[(n2 + k) & (kRing - 1)] = v;
                            sym_store(sym + n2 + k, v);
                        }
                    }
                    n2 += len;
                    p += tl;
                }
                if (!block_done && !rc) pos += p;
            }
            if (rc) {
                status |= rc;
                break;
            }
            at = pos;
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
// (round 6: ONE map / window in LDS, composed in place -- a lane keeps its 32 new entries in registers over the barrier.  With two
// buffers k_gz_maps took 128 KB of the CU's 160: it could only start on a CU nothing else held LDS on, and while the classifier's
// workgroups -- 32 KB each, on every CU, also the ones the decode passes leave free -- came and went it waited: windows_crc_s of a
// 34-GB stream 0.6 s in most runs and 2.5-3.3 s in others, profiles/round6_cli_c2_ab_free_cus.txt)
__global__ void __launch_bounds__(1024) k_gz_maps(const AccDev *acc, uint32_t n_acc, uint16_t *maps) {
    extern __shared__ uint16_t s_map[];                                       // kWindow entries
    const uint32_t tid = threadIdx.x;
    const uint32_t c0 = blockIdx.x * kGroup, c1 = c0 + kGroup < n_acc ? c0 + kGroup : n_acc;
    uint16_t *cur = s_map;
    constexpr uint32_t kPer = kWindow / 1024;
    for (uint32_t k = tid; k < kWindow; k += 1024) cur[k] = (uint16_t)(kMarker + k);      // the identity: byte k of the window in front of the group
    __syncthreads();
    for (uint32_t c = c0; c < c1; ++c) {
        const AccDev a = acc[c];
        const uint16_t *s = a.sym;
        const uint32_t own = a.n_out < kWindow ? a.n_out : kWindow;           // symbols of this chunk in its window
        uint16_t *out = maps + (size_t)c * kWindow;
        uint16_t nv[kPer];
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) {
            const uint32_t k = tid + q * 1024;
            uint16_t v;
            if (k < kWindow - own) v = a.no_history ? (uint16_t)0 : cur[k + own];      // the window in front, shifted (nothing in front of a member)
            else {
                v = s[a.n_out - own + (k - (kWindow - own))];
                if (v >= kMarker) v = cur[v - kMarker];
            }
            nv[q] = v;
            out[k] = v;
        }
        __syncthreads();                                                       // (every lane has read the old map)
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) cur[tid + q * 1024] = nv[q];
        __syncthreads();
    }
}
__global__ void __launch_bounds__(1024) k_gz_carry(const uint16_t *maps, uint32_t n_acc, const uint8_t *carry, uint8_t *gwin) {
    extern __shared__ uint8_t s_win[];                                        // kWindow bytes
    const uint32_t tid = threadIdx.x;
    uint8_t *cur = s_win;
    constexpr uint32_t kPer = kWindow / 1024;
    for (uint32_t k = tid; k < kWindow; k += 1024) cur[k] = carry[k];
    __syncthreads();
    const uint32_t n_groups = (n_acc + kGroup - 1) / kGroup;
    for (uint32_t g = 0; g < n_groups; ++g) {
        uint8_t *out = gwin + (size_t)g * kWindow;
        for (uint32_t k = tid; k < kWindow; k += 1024) out[k] = cur[k];
        const uint32_t last = (g + 1) * kGroup < n_acc ? (g + 1) * kGroup - 1 : n_acc - 1;
        const uint16_t *m = maps + (size_t)last * kWindow;
        uint8_t nv[kPer];
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) {
            const uint16_t v = m[tid + q * 1024];
            nv[q] = v < kMarker ? (uint8_t)v : cur[v - kMarker];
        }
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) cur[tid + q * 1024] = nv[q];
        __syncthreads();
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
hipError_t launch_decode(ChunkJob *d_jobs, uint32_t n, const uint32_t *d_w, uint64_t nbits, uint16_t *d_syms, uint32_t *d_pool_cursor, uint32_t pool_slots, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_gz_decode, dim3(n), dim3(64), 0, s, d_jobs, n, d_w, nbits, d_syms, d_pool_cursor, pool_slots);
    return hipGetLastError();
}
size_t windows_scratch_bytes(uint32_t n) { return (size_t)n * kWindow * sizeof(uint16_t) + (size_t)((n + kGroup - 1) / kGroup) * kWindow; }
hipError_t launch_windows(const AccDev *d_acc, uint32_t n, uint8_t *d_windows, const uint8_t *d_carry, void *d_scratch, hipStream_t s) {
    if (!n) return hipSuccess;
    uint16_t *maps = reinterpret_cast<uint16_t *>(d_scratch);
    uint8_t *gwin = reinterpret_cast<uint8_t *>(maps + (size_t)n * kWindow);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_gz_maps), hipFuncAttributeMaxDynamicSharedMemorySize, kWindow * (int)sizeof(uint16_t));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_gz_carry), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWindow);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const uint32_t n_groups = (n + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(k_gz_maps, dim3(n_groups), dim3(1024), kWindow * sizeof(uint16_t), s, d_acc, n, maps);
    hipLaunchKernelGGL(k_gz_carry, dim3(1), dim3(1024), kWindow, s, maps, n, d_carry, gwin);
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

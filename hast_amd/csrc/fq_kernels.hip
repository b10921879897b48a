// fq_kernels.hip -- gfx950 device code: FASTQ framing on the GPU (the producer side of the hot path).
//
// The reference's producer (processFastq, 01.classify_stlfr_reads/classify.cpp:238-278) reads the file one getline at a
// time: a record is four '\n'-separated lines counted from the start of the file, line 1 is the header (its barcode is the
// text between the last '#' and the last '/', parseName :112-119), line 2 the bases; nothing else is looked at.  Here the raw
// bytes of the file go to HBM as they are and three small kernels do the same framing for a whole block at once:
//   k_fq_count   newlines per 4-KB tile (16 bytes per lane, has-zero-byte on x ^ '\n\n\n\n')
//   k_fq_scan    exclusive scan of the tile counts (one workgroup; a block has a few thousand tiles)
//   k_fq_index   positions of all newlines, in order (each tile re-derives its masks and writes at its scanned base)
//   k_fq_records one lane per record: header / bases extents, barcode extent, length checks; the bytes after the last
//                complete record (the "tail") are kept for the next block of the same file.
// k_classify_f then reads the bases straight out of the raw block (offsets + lengths), the host only maps barcode text to
// dense ids (the dictionary has to be one per job, not per GPU) and hands them back for k_commit_votes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fq_device.h"

namespace hast {

constexpr int kFqTile = 4096;                    // bytes per tile = 256 lanes x 16 B

__device__ __forceinline__ uint32_t nl_mask16(const uint8_t *p, const uint8_t *lo, const uint8_t *hi) {
    // bit i set <=> p[i] == '\n', for the 16 bytes at p (16-B aligned); bytes outside [lo, hi) never count
    uint32_t m = 0;
    if (p + 16 <= lo || p >= hi) return 0;
    const uint4 v = *reinterpret_cast<const uint4 *>(p);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t y = w[i] ^ 0x0A0A0A0Au;
        const uint32_t z = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;     // 0x80 in every zero byte, exact
        m |= (((z >> 7) * 0x00204081u) >> 21 & 0xFu) << (4 * i);                          // bits 0,8,16,24 -> 4 adjacent bits
    }
    if (p < lo) m &= 0xFFFFu << (uint32_t)(lo - p);
    if (p + 16 > hi) m &= 0xFFFFu >> (uint32_t)(p + 16 - hi);
    return m & 0xFFFFu;
}

// tiles are 4-KB aligned pieces of the buffer; `lo`/`hi` bound the bytes that belong to this parse
__global__ void __launch_bounds__(256) k_fq_count(const uint8_t *buf, const FqState *st, uint32_t *tile_cnt) {
    const uint64_t lo = st->parse_lo, hi = st->parse_hi;
    const uint64_t tile = blockIdx.x;                                                      // tiles cover the buffer from byte 0
    if (tile * kFqTile >= hi || (tile + 1) * kFqTile <= lo) { if (threadIdx.x == 0) tile_cnt[blockIdx.x] = 0; return; }
    const uint8_t *p = buf + tile * kFqTile + threadIdx.x * 16;
    uint32_t c = __popc(nl_mask16(p, buf + lo, buf + hi));
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    __shared__ uint32_t s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// exclusive scan of n tile counts in place (n <= a few 10^4), total -> st->n_nl; one workgroup of 1024
__global__ void __launch_bounds__(1024) k_fq_scan(uint32_t *tile_cnt, uint32_t n, FqState *st) {
    __shared__ uint32_t s_part[1024];
    const uint32_t per = (n + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += tile_cnt[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {                 // Hillis-Steele inclusive scan
        const uint32_t v = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? s_part[threadIdx.x - 1] : 0;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t c = tile_cnt[i];
        tile_cnt[i] = run;
        run += c;
    }
    if (threadIdx.x == 1023) st->n_nl = s_part[1023];
}

__global__ void __launch_bounds__(256) k_fq_index(const uint8_t *buf, const FqState *st, const uint32_t *tile_base, uint32_t *nl) {
    const uint64_t lo = st->parse_lo, hi = st->parse_hi;
    const uint64_t tile = blockIdx.x;
    if (tile * kFqTile >= hi || (tile + 1) * kFqTile <= lo) return;
    const uint64_t at = tile * kFqTile + threadIdx.x * 16;
    uint32_t m = nl_mask16(buf + at, buf + lo, buf + hi);
    const uint32_t c = __popc(m);
    // exclusive prefix of c over the workgroup
    uint32_t incl = c;
    const uint32_t lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += v;
    }
    __shared__ uint32_t s[4];
    if (lane == 63) s[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = tile_base[blockIdx.x] + incl - c;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) base += s[w];
    while (m) {
        const uint32_t b = __ffs(m) - 1;
        nl[base++] = (uint32_t)(at + b);
        m &= m - 1;
    }
}

// record i of a block: header line [h0, h1), bases [h1 + 1, s1)
__device__ __forceinline__ void fq_emit_record(const uint8_t *buf, FqState *st, uint32_t i, uint64_t h0, uint64_t h1, uint64_t s1, uint64_t *r_off,
                                               uint32_t *r_len, uint32_t *bc_pos, uint32_t *bc_len, uint32_t *h_bc, uint32_t *d_text, uint32_t h_cap,
                                               uint32_t k) {
    const uint32_t len = (uint32_t)(s1 - h1 - 1 > 0xFFFFFFFFull ? 0xFFFFFFFFull : s1 - h1 - 1);
    r_off[i] = h1 + 1;
    r_len[i] = len;
    // parseName (classify.cpp:112-119): s = last '#', e = last '/'; barcode = [s+1, e) when e > s, else [s+1, end of header)
    int64_t s = -1, e = -1;
    for (int64_t p = (int64_t)h1 - 1; p >= (int64_t)h0; --p) {
        const uint8_t c = buf[p];
        if (c == '/' && e < 0) e = p;
        if (c == '#') { s = p; break; }
    }
    const int64_t start = s < 0 ? (int64_t)h0 : s + 1;
    const int64_t stop = (e > s && e >= 0) ? e : (int64_t)h1;
    bc_pos[i] = (uint32_t)start;
    bc_len[i] = (uint32_t)(stop - start);
    if (i < h_cap) {
        h_bc[i] = (uint32_t)start;
        h_bc[h_cap + i] = (uint32_t)(stop - start);
        const uint32_t bl = (uint32_t)(stop - start);
        uint32_t w[4] = {0, 0, 0, 0};
        if (bl <= 15) {
            w[0] = bl;
            for (uint32_t j = 0; j < bl; ++j) w[(j + 1) >> 2] |= (uint32_t)buf[start + j] << (8 * ((j + 1) & 3));
        } else w[0] = 0xFF;
        reinterpret_cast<uint4 *>(h_bc + 2 * (size_t)h_cap)[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    {   // the same text record for the naming kernel, for EVERY record of the block (d_text holds the whole record table): the dictionary
        // must see a barcode whichever block it comes in -- one named by the host in a block with more records than the pinned arrays
        // hold and by the device in the next block would have two ids
        const uint32_t bl = (uint32_t)(stop - start);
        uint32_t w[4] = {0, 0, 0, 0};
        if (bl <= 15) {
            w[0] = bl;
            for (uint32_t j = 0; j < bl; ++j) w[(j + 1) >> 2] |= (uint32_t)buf[start + j] << (8 * ((j + 1) & 3));
        } else w[0] = 0xFF;
        reinterpret_cast<uint4 *>(d_text)[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    atomicMax(&st->max_len, len);
    atomicAdd(reinterpret_cast<unsigned long long *>(&st->bases), (unsigned long long)len);
    if (len < k) {                                            // the reference aborts on such a read unless it holds 'N' (kmer.h:171)
        bool has_n = false;
        for (uint64_t p = h1 + 1; p < s1 && !has_n; ++p) has_n = buf[p] == 'N';
        if (!has_n) atomicOr(&st->flags, 1u);
    }
}

// One lane per complete record i (lines 4i .. 4i+3 of the parse range, which always starts at a record boundary).
// `last`: the input ends with this block, so a final record whose header line is terminated counts even when its bases /
// '+' / quality lines lack their newlines (classify.cpp:257-268: a getline that hits EOF still yields the line read).
// h_bc: pinned HOST memory the kernel writes the first h_cap barcode extents to ([pos x h_cap | len x h_cap]), so that the host
// needs no device-to-host copy for them (such a copy would queue up behind the next block's 16-MB upload), followed by the
// barcode TEXT itself, 16 bytes per record: length byte + up to 15 bytes (0xFF: longer, take it from the block).  The host
// then names a record from one compact, sequentially read array instead of one cache miss per header line.
__global__ void __launch_bounds__(256) k_fq_records(const uint8_t *buf, FqState *st, const uint32_t *nl, uint64_t *r_off, uint32_t *r_len,
                                                    uint32_t *bc_pos, uint32_t *bc_len, uint32_t *h_bc, uint32_t *d_text, uint32_t h_cap, uint32_t k, int last) {
    const uint64_t lo = st->parse_lo, hi = st->parse_hi;
    const uint32_t n_nl = st->n_nl;
    uint32_t n_rec = n_nl / 4;
    // the EOF record without all four newlines: it counts iff its header line is terminated (even with an empty bases line)
    const bool partial = last && n_nl - 4 * n_rec >= 1;
    const uint32_t total = n_rec + (partial ? 1 : 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->n_rec = total;
        const uint64_t consumed = last ? hi : (n_rec ? (uint64_t)nl[4 * n_rec - 1] + 1 : lo);
        st->tail_lo = consumed;                               // bytes [consumed, hi) belong to the next block's first record
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint64_t h0 = i ? (uint64_t)nl[4 * i - 1] + 1 : lo;
        const uint64_t h1 = nl[4 * i];                            // end of the header line (exists for every counted record)
        const uint64_t s1 = (4 * i + 1 < n_nl) ? nl[4 * i + 1] : hi;   // end of the bases line, or EOF
        fq_emit_record(buf, st, i, h0, h1, s1, r_off, r_len, bc_pos, bc_len, h_bc, d_text, h_cap, k);
    }
}

// Striped streams: the blocks of ONE file are framed on several GPUs (classify.cpp:211-219 spreads the reads of one file over
// all its workers).  A block cannot wait for the block in front of it, so it is framed from what the host can know early: the
// number of newlines in the bytes in front of it (k_fq_count of the earlier blocks, summed on the host), i.e. the line `phase`
// its first byte lies in.  Records are four lines counted from the start of the file (no '@'/'+' validation), so a record
// starts behind every newline whose index in the file is 3 mod 4 -- and at the block's first byte when phase == 0 and a line
// starts there.  A block owns the records that START in its own bytes; their header and base lines may reach into the view's
// overlap (the first bytes of the next block); '+' and quality lines are only ever counted.
__global__ void __launch_bounds__(256) k_fq_records_striped(const uint8_t *buf, FqState *st, const uint32_t *nl, uint64_t *r_off, uint32_t *r_len,
                                                            uint32_t *bc_pos, uint32_t *bc_len, uint32_t *h_bc, uint32_t *d_text, uint32_t h_cap,
                                                            uint32_t k) {
    const uint64_t lo = st->parse_lo, hi = st->parse_hi, own_hi = st->own_hi;
    const uint32_t n_nl = st->n_nl, s0 = (3u - st->phase) & 3u;       // nl[s0], nl[s0 + 4], ... end a record
    const uint32_t has0 = (st->phase == 0 && st->bol) ? 1u : 0u;      // a record starts at lo
    const bool eof = st->eof != 0;
    const uint32_t cand = has0 + (n_nl > s0 ? (n_nl - s0 + 3) / 4 : 0);
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < cand; c += gridDim.x * blockDim.x) {
        const uint32_t j = s0 + 4 * (c - has0);                       // (c >= has0) the newline in front of the record
        const uint64_t h0 = c < has0 ? lo : (uint64_t)nl[j] + 1;
        if (h0 >= own_hi) continue;                                   // starts in the next block (or at EOF): not this block's
        const uint32_t hidx = c < has0 ? 0u : j + 1;                  // the newline that ends its header line
        if (hidx >= n_nl) {
            // the header line is not terminated inside the view: at EOF the reference drops such a record (classify.cpp:257-259);
            // anywhere else the record is larger than the overlap
            if (!eof) atomicOr(&st->flags, 2u);
            continue;
        }
        const uint64_t h1 = nl[hidx];
        uint64_t s1 = hi;                                             // at EOF the bases are whatever follows (classify.cpp:260-268)
        if (hidx + 1 < n_nl) s1 = nl[hidx + 1];
        else if (!eof) { atomicOr(&st->flags, 2u); continue; }
        atomicMax(&st->n_rec, c + 1);                                 // the records of a block are a prefix of its candidates
        fq_emit_record(buf, st, c, h0, h1, s1, r_off, r_len, bc_pos, bc_len, h_bc, d_text, h_cap, k);
    }
}

__global__ void k_fq_view(FqState *st, uint64_t lo, uint64_t hi, uint64_t own_hi, uint32_t phase, uint32_t bol, uint32_t eof) {
    st->parse_lo = lo;
    st->parse_hi = hi;
    st->own_hi = own_hi;
    st->phase = phase;
    st->bol = bol;
    st->eof = eof;
}

// copies the previous block's tail in front of this block's bytes and sets the parse range; one workgroup is plenty
// (a tail is part of one record).  prev == nullptr: first block of a file.
__global__ void __launch_bounds__(256) k_fq_begin(uint8_t *buf, FqState *st, const uint8_t *prev_buf, const FqState *prev, uint64_t pad,
                                                  uint64_t n_bytes) {
    uint64_t tail = 0;
    if (prev) {
        tail = prev->parse_hi - prev->tail_lo;
        if (tail > pad) { tail = pad; if (threadIdx.x == 0) atomicOr(&st->flags, 2u); }      // a record larger than the pad
        const uint8_t *src = prev_buf + prev->parse_hi - tail;
        uint8_t *dst = buf + pad - tail;
        for (uint64_t i = threadIdx.x; i < tail; i += blockDim.x) dst[i] = src[i];
    }
    if (threadIdx.x == 0) {
        st->parse_lo = pad - tail;
        st->parse_hi = pad + n_bytes;
        st->tail_in = tail;
    }
}

// ---- device-side name cache ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t name_hash(const uint32_t k[4]) {
    uint32_t h = k[0] * 0x9E3779B1u;
    h = (h ^ (h >> 15) ^ k[1]) * 0x85EBCA6Bu;
    h = (h ^ (h >> 13) ^ k[2]) * 0xC2B2AE35u;
    h = (h ^ (h >> 16) ^ k[3]) * 0x27D4EB2Fu;
    return h ^ (h >> 15);
}
// Runs on the context's stream, after every k_names_insert of earlier blocks: no insert is in flight while it probes.
__global__ void __launch_bounds__(256) k_fq_name(const uint32_t *text, uint32_t n, const NameEntry *tab, uint32_t mask, uint32_t *h_ids,
                                                 uint32_t *h_unknown) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint4 t = reinterpret_cast<const uint4 *>(text)[i];
        const uint32_t k[4] = {t.x, t.y, t.z, t.w};
        uint32_t id = kNameUnknown;
        if ((t.x & 0xFFu) != 0xFFu && tab) {
            uint32_t at = name_hash(k) & mask;
            for (uint32_t probe = 0; probe <= mask; ++probe, at = (at + 1) & mask) {
                const NameEntry &e = tab[at];
                if (e.state == 0) break;
                if (e.state == 2 && e.key[0] == k[0] && e.key[1] == k[1] && e.key[2] == k[2] && e.key[3] == k[3]) { id = e.id; break; }
            }
        }
        h_ids[i] = id;
        if (id == kNameUnknown) h_unknown[1 + atomicAdd(&h_unknown[0], 1u)] = i;             // (pinned host memory)
    }
}
// Two lanes of ONE wave may claim the same slot (the same new barcode twice in a block): the loser must not spin on the
// winner's "ready" flag inside a divergent branch -- which side of a branch a wave runs first is the compiler's choice.  So
// every lane is a small state machine stepped by a wave-uniform loop: a lane that finds a slot "being written" simply looks
// again in the next round, by when the winner (same wave or not) has had its turn to publish.
__global__ void __launch_bounds__(256) k_names_insert(const NamePub *pubs, uint32_t n, NameEntry *tab, uint32_t mask) {
    const uint32_t n_round = (n + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
    for (uint32_t r = 0; r < n_round; ++r) {
        const uint32_t i = r * gridDim.x * blockDim.x + blockIdx.x * blockDim.x + threadIdx.x;
        bool busy = i < n;
        NamePub p = {};
        if (busy) p = pubs[i];
        uint32_t at = busy ? name_hash(p.key) & mask : 0, probes = 0;
        while (__any(busy)) {
            if (busy) {
                NameEntry *e = &tab[at];
                uint32_t st = __hip_atomic_load(&e->state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (st == 0) {
                    st = atomicCAS(&e->state, 0u, 1u);
                    if (st == 0) {                                    // claimed: write and publish
                        e->key[0] = p.key[0]; e->key[1] = p.key[1]; e->key[2] = p.key[2]; e->key[3] = p.key[3];
                        e->id = p.id;
                        __hip_atomic_store(&e->state, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        busy = false;
                    }
                }
                if (busy && st == 2) {                                // a ready entry: the same barcode, or someone else's -> next slot
                    if (e->key[0] == p.key[0] && e->key[1] == p.key[1] && e->key[2] == p.key[2] && e->key[3] == p.key[3]) busy = false;
                    else {
                        at = (at + 1) & mask;
                        if (++probes > mask) busy = false;            // (a full table: the cache simply does not learn this one)
                    }
                }
                // st == 1: being written by another lane -- look again in the next round
            }
        }
    }
}

// ---- routing: steps 10-11 of the wrapper on the GPU (classify_stlfr_reads.sh:155-190, quartering_fastq.awk) -------------------------
// After `classify` the wrapper runs a single-threaded awk program over every input once more: a record (four lines) goes to
// <name>.{nobarcode,paternal,maternal,homozygous}.fastq by the class of its barcode.  awk splits a header at every '#' or '/'
// (-F '#|/'); NF > 1 and $2 != "0_0_0" -> $2 is looked up in the three lists (:22-35), otherwise the read has no barcode (:36-39).
// Here a lane does that for a record of a framed block: field 2 as a 16-byte text record (length byte + up to 15 bytes, the format of
// the name cache), looked up in a table text -> class the host filled from its barcode lists.  What the table cannot answer -- a field
// longer than 15 bytes, a barcode in no list (awk prints an ERROR line with its text and drops the record) -- flags the BLOCK for the
// host, which then routes that block itself from the same bytes; every other block leaves the GPU as four runs of whole records in
// input order.
template <bool STRIPED>
__device__ __forceinline__ bool route_extent(const FqState *st, const uint32_t *nl, uint32_t c, uint32_t n_cand, int last, uint64_t &h0, uint64_t &h1, uint64_t &end,
                                             RouteState *rs) {
    const uint64_t lo = st->parse_lo, hi = st->parse_hi;
    const uint32_t n_nl = st->n_nl;
    if (!STRIPED) {                                         // records are lines 4c .. 4c+3 of the parse range (it starts at a record boundary)
        h0 = c ? (uint64_t)nl[4 * c - 1] + 1 : lo;
        h1 = nl[4 * c];
        end = (uint64_t)nl[4 * c + 3] + 1;
        if (c + 1 == n_cand && last && end < hi) { rs->tail_lo = end; rs->tail_hi = hi; }
        return true;
    }
    const uint32_t s0 = (3u - st->phase) & 3u, has0 = (st->phase == 0 && st->bol) ? 1u : 0u;
    const uint32_t j = s0 + 4 * (c - has0);
    h0 = c < has0 ? lo : (uint64_t)nl[j] + 1;
    if (h0 >= st->own_hi) return false;                     // starts in the next block: not this block's
    const uint32_t hidx = c < has0 ? 0u : j + 1;
    if (hidx + 3 >= n_nl) {                                 // fewer than four newlines left in the view
        if (st->eof) { rs->tail_lo = h0; rs->tail_hi = hi; }    // the end of the file: the host's (at most one candidate gets here)
        else atomicOr(&rs->flags, 2u);
        return false;
    }
    h1 = nl[hidx];
    end = (uint64_t)nl[hidx + 3] + 1;
    return true;
}

__device__ __forceinline__ uint32_t route_candidates(const FqState *st, bool striped) {
    const uint32_t n_nl = st->n_nl;
    if (!striped) return n_nl / 4;
    const uint32_t s0 = (3u - st->phase) & 3u, has0 = (st->phase == 0 && st->bol) ? 1u : 0u;
    return has0 + (n_nl > s0 ? (n_nl - s0 + 3) / 4 : 0);
}

template <bool STRIPED>
__global__ void __launch_bounds__(kRouteTile) k_route_class(const uint8_t *buf, const FqState *st, const uint32_t *nl, int last, const NameEntry *tab, uint32_t mask,
                                                            uint32_t *r_start, uint32_t *r_len, uint8_t *r_cls, uint32_t *tile_sum, uint32_t max_rec, RouteState *rs) {
    uint32_t n_cand = route_candidates(st, STRIPED);
    if (n_cand > max_rec) n_cand = max_rec;                 // (cannot be: a record holds four newlines; the arrays hold view / 4)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        rs->n_cand = n_cand;
        if (!STRIPED && last && n_cand == 0 && st->parse_lo < st->parse_hi) { rs->tail_lo = st->parse_lo; rs->tail_hi = st->parse_hi; }
    }
    __shared__ uint32_t s_sum[8];
    const uint32_t n_tiles = (n_cand + kRouteTile - 1) / kRouteTile;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (threadIdx.x < 8) s_sum[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t c = tile * kRouteTile + threadIdx.x;
        if (c < n_cand) {
            uint64_t h0 = 0, h1 = 0, end = 0;
            uint8_t cls = kRouteSkip;
            uint32_t len = 0;
            if (route_extent<STRIPED>(st, nl, c, n_cand, last, h0, h1, end, rs)) {
                len = (uint32_t)(end - h0);
                // awk's $2 under -F '#|/': the text between the first separator and the next one (or the end of the line)
                uint64_t a = h1;
                for (uint64_t p = h0; p < h1; ++p) {
                    const uint8_t ch = buf[p];
                    if (ch == '#' || ch == '/') { a = p; break; }
                }
                if (a == h1) cls = 0;                       // NF <= 1 (:22,36-39)
                else {
                    uint64_t b = h1;
                    for (uint64_t p = a + 1; p < h1; ++p) {
                        const uint8_t ch = buf[p];
                        if (ch == '#' || ch == '/') { b = p; break; }
                    }
                    const uint32_t fl = (uint32_t)(b - a - 1);
                    if (fl > 15) cls = kRouteHost;
                    else {
                        uint32_t w[4] = {fl, 0, 0, 0};
                        for (uint32_t q = 0; q < fl; ++q) w[(q + 1) >> 2] |= (uint32_t)buf[a + 1 + q] << (8 * ((q + 1) & 3));
                        if (w[0] == 0x305F3005u && w[1] == 0x0000305Fu) cls = 0;       // "0_0_0" (:22): length 5, then the text
                        else {
                            cls = kRouteUnclassified;
                            uint32_t at = name_hash(w) & mask;
                            for (uint32_t probe = 0; tab && probe <= mask; ++probe, at = (at + 1) & mask) {
                                const NameEntry &e = tab[at];
                                if (e.state == 0) break;
                                if (e.state == 2 && e.key[0] == w[0] && e.key[1] == w[1] && e.key[2] == w[2] && e.key[3] == w[3]) { cls = (uint8_t)e.id; break; }
                            }
                        }
                    }
                }
                if (cls >= kRouteUnclassified) atomicOr(&rs->flags, 1u);
                else {
                    atomicAdd(&s_sum[cls], len);
                    atomicAdd(&s_sum[4 + cls], 1u);
                }
            }
            r_start[c] = (uint32_t)h0;
            r_len[c] = len;
            r_cls[c] = cls;
        }
        __syncthreads();
        if (threadIdx.x < 8) tile_sum[(size_t)tile * 8 + threadIdx.x] = s_sum[threadIdx.x];
        __syncthreads();
    }
}

// exclusive scan of the tiles' byte sums per class, in place; totals and class counts -> rs.  One workgroup.
__global__ void __launch_bounds__(1024) k_route_scan(uint32_t *tile_sum, RouteState *rs) {
    __shared__ uint32_t s_part[4][1024];
    __shared__ uint32_t s_cnt[4];
    const uint32_t n = (rs->n_cand + kRouteTile - 1) / kRouteTile;
    const uint32_t per = (n + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    uint32_t sum[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
    for (uint32_t i = lo; i < hi; ++i)
        for (int c = 0; c < 4; ++c) { sum[c] += tile_sum[(size_t)i * 8 + c]; cnt[c] += tile_sum[(size_t)i * 8 + 4 + c]; }
    for (int c = 0; c < 4; ++c) s_part[c][threadIdx.x] = sum[c];
    __syncthreads();
    for (int c = 0; c < 4; ++c) if (cnt[c]) atomicAdd(&s_cnt[c], cnt[c]);
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v[4];
        for (int c = 0; c < 4; ++c) v[c] = threadIdx.x >= d ? s_part[c][threadIdx.x - d] : 0;
        __syncthreads();
        for (int c = 0; c < 4; ++c) s_part[c][threadIdx.x] += v[c];
        __syncthreads();
    }
    // class c's run starts behind the runs of the classes in front of it
    uint32_t base[4];
    base[0] = 0;
    for (int c = 1; c < 4; ++c) base[c] = base[c - 1] + s_part[c - 1][1023];
    uint32_t run[4];
    for (int c = 0; c < 4; ++c) run[c] = base[c] + (threadIdx.x ? s_part[c][threadIdx.x - 1] : 0);
    for (uint32_t i = lo; i < hi; ++i)
        for (int c = 0; c < 4; ++c) {
            const uint32_t v = tile_sum[(size_t)i * 8 + c];
            tile_sum[(size_t)i * 8 + c] = run[c];
            run[c] += v;
        }
    if (threadIdx.x < 4) {
        rs->bytes[threadIdx.x] = s_part[threadIdx.x][1023];
        rs->count[threadIdx.x] = s_cnt[threadIdx.x];
    }
}

// the records of a tile to their places: a lane per record finds its offset (prefix sum over the records of its class in front of
// it in the tile), then every wave copies its 64 records one after the other, 64 bytes per step
__global__ void __launch_bounds__(kRouteTile) k_route_copy(const uint8_t *buf, const uint32_t *r_start, const uint32_t *r_len, const uint8_t *r_cls,
                                                           const uint32_t *tile_base, const RouteState *rs, uint8_t *out) {
    __shared__ uint32_t s_src[kRouteTile], s_dst[kRouteTile], s_n[kRouteTile];
    __shared__ uint32_t s_wave[4][4];
    const uint32_t n_cand = rs->n_cand, n_tiles = (n_cand + kRouteTile - 1) / kRouteTile;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t c = tile * kRouteTile + threadIdx.x;
        const uint8_t cls = c < n_cand ? r_cls[c] : kRouteSkip;
        const uint32_t len = (c < n_cand && cls < 4) ? r_len[c] : 0;
        uint32_t incl[4];
        for (int k = 0; k < 4; ++k) {
            uint32_t v = cls == k ? len : 0;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t u = __shfl_up(v, off, 64);
                if (lane >= (uint32_t)off) v += u;
            }
            incl[k] = v;
            if (lane == 63) s_wave[wave][k] = v;
        }
        __syncthreads();
        uint32_t dst = 0;
        if (cls < 4) {
            dst = tile_base[(size_t)tile * 8 + cls] + incl[cls] - len;
            for (uint32_t w = 0; w < wave; ++w) dst += s_wave[w][cls];
        }
        s_src[threadIdx.x] = c < n_cand ? r_start[c] : 0;
        s_dst[threadIdx.x] = dst;
        s_n[threadIdx.x] = len;
        __syncthreads();
        for (uint32_t r = 0; r < 64; ++r) {
            const uint32_t i = wave * 64 + r;
            const uint32_t n = s_n[i];
            const uint8_t *src = buf + s_src[i];
            uint8_t *d = out + s_dst[i];
            for (uint32_t b = lane; b < n; b += 64) d[b] = src[b];
        }
        __syncthreads();
    }
}

hipError_t launch_fq_route(const uint8_t *d_buf, const FqState *d_st, const uint32_t *d_nl, int striped, int last, const NameEntry *tab, uint32_t mask,
                           uint32_t *d_rstart, uint32_t *d_rlen, uint8_t *d_rcls, uint32_t *d_rtile, uint32_t max_rec, RouteState *d_rs, uint8_t *d_out,
                           hipStream_t s) {
    hipError_t e = hipMemsetAsync(d_rs, 0, sizeof(RouteState), s);
    if (e != hipSuccess) return e;
    if (striped) hipLaunchKernelGGL(k_route_class<true>, dim3(1024), dim3(kRouteTile), 0, s, d_buf, d_st, d_nl, last, tab, mask, d_rstart, d_rlen, d_rcls, d_rtile, max_rec, d_rs);
    else hipLaunchKernelGGL(k_route_class<false>, dim3(1024), dim3(kRouteTile), 0, s, d_buf, d_st, d_nl, last, tab, mask, d_rstart, d_rlen, d_rcls, d_rtile, max_rec, d_rs);
    hipLaunchKernelGGL(k_route_scan, dim3(1), dim3(1024), 0, s, d_rtile, d_rs);
    hipLaunchKernelGGL(k_route_copy, dim3(2048), dim3(kRouteTile), 0, s, d_buf, d_rstart, d_rlen, d_rcls, d_rtile, d_rs, d_out);
    return hipGetLastError();
}


// The DICTIONARY variant (round 6): the table hands out the ids itself.  classify.cpp:52-56 gives a barcode its map entry at its first
// sighting; until round 5 that first sighting went to the host (a dictionary insert there, the id taught back to this table): a third
// of a read phase at 10M barcodes.  Here the lane that meets an unknown text claims its slot (compare-and-swap empty -> being written),
// takes the next id from one counter, writes text and id, files the text under its id for the host (text_of_id: read once, at the
// end, for printing) and publishes.  The same state machine as k_names_insert, for the same reason: two lanes of one wave may meet the
// same new barcode.  Left to the host: texts longer than 15 bytes and what arrives once `limit` ids are out (the host names those in an
// id range of its own, above `limit`).  Streams of several contexts of ONE GPU may run this on one table at the same time.
__global__ void __launch_bounds__(256) k_fq_name_claim(const uint32_t *text, uint32_t n, const FqState *st, NameEntry *tab, uint32_t mask, uint32_t *n_ids, uint32_t limit,
                                                       uint4 *text_of_id, uint32_t *h_ids, uint32_t *h_unknown, uint32_t *d_ids) {
    // (st != NULL: launched right behind the framing kernels, before the host knows the block's record count -- the count is the framer's,
    // at most n = what the pinned id arrays hold: a block with more records is named again, in full, once they have been regrown)
    if (st) n = st->n_rec < n ? st->n_rec : n;
    const uint32_t n_round = (n + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
    for (uint32_t r = 0; r < n_round; ++r) {
        const uint32_t i = r * gridDim.x * blockDim.x + blockIdx.x * blockDim.x + threadIdx.x;
        bool busy = i < n;
        uint4 t = make_uint4(0, 0, 0, 0);
        if (busy) t = reinterpret_cast<const uint4 *>(text)[i];
        const uint32_t k[4] = {t.x, t.y, t.z, t.w};
        uint32_t id = kNameUnknown;
        if (busy && (t.x & 0xFFu) == 0xFFu) busy = false;              // longer than a text record: the host's
        const bool mine = i < n;
        uint32_t at = busy ? name_hash(k) & mask : 0, probes = 0;
        while (__any(busy)) {
            if (busy) {
                NameEntry *e = &tab[at];
                uint32_t st = __hip_atomic_load(&e->state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (st == 0) {
                    if (__hip_atomic_load(n_ids, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= limit) busy = false;       // every id is out
                    else {
                        st = atomicCAS(&e->state, 0u, 1u);
                        if (st == 0) {                                    // claimed
                            const uint32_t got = atomicAdd(n_ids, 1u);
                            if (got >= limit) {                           // (the last ids went while this lane claimed: give the slot back)
                                __hip_atomic_store(&e->state, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                            } else {
                                e->key[0] = k[0]; e->key[1] = k[1]; e->key[2] = k[2]; e->key[3] = k[3];
                                e->id = got;
                                text_of_id[got] = t;
                                __hip_atomic_store(&e->state, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                                id = got;
                            }
                            busy = false;
                        }
                    }
                }
                if (busy && st == 2) {
                    if (e->key[0] == k[0] && e->key[1] == k[1] && e->key[2] == k[2] && e->key[3] == k[3]) { id = e->id; busy = false; }
                    else {
                        at = (at + 1) & mask;
                        if (++probes > mask) busy = false;
                    }
                }
                // st == 1: being written by another lane (of this wave, or of another stream's kernel) -- look again next round
            }
        }
        if (mine) {
            h_ids[i] = id;
            // (the same ids in device memory: a block the dictionary named completely is booked from there -- the bookkeeping kernel reading
            // 50 000 ids out of pinned host memory took as long as the classification of the block, round6_cli_c2_gz_kernel_stats_final.csv)
            if (d_ids) d_ids[i] = id;
            if (id == kNameUnknown) h_unknown[1 + atomicAdd(&h_unknown[0], 1u)] = i;             // (pinned host memory)
        }
    }
}
hipError_t launch_fq_name_claim(const uint32_t *d_text, uint32_t n, NameEntry *tab, uint32_t mask, uint32_t *d_n_ids, uint32_t limit, void *d_text_of_id,
                                uint32_t *h_ids, uint32_t *h_unknown, uint32_t *d_ids, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fq_name_claim, dim3((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), dim3(256), 0, s, d_text, n, (const FqState *)nullptr, tab, mask, d_n_ids,
                       limit, reinterpret_cast<uint4 *>(d_text_of_id), h_ids, h_unknown, d_ids);
    return hipGetLastError();
}
// the same behind the framing kernels of a block, on their stream: the record count is read from the block's state on the device
hipError_t launch_fq_name_claim_framed(const uint32_t *d_text, const FqState *d_st, uint32_t cap, NameEntry *tab, uint32_t mask, uint32_t *d_n_ids, uint32_t limit,
                                       void *d_text_of_id, uint32_t *h_ids, uint32_t *h_unknown, uint32_t *d_ids, hipStream_t s) {
    if (cap == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fq_name_claim, dim3((cap + 255) / 256 < 1024 ? (cap + 255) / 256 : 1024), dim3(256), 0, s, d_text, cap, d_st, tab, mask, d_n_ids, limit,
                       reinterpret_cast<uint4 *>(d_text_of_id), h_ids, h_unknown, d_ids);
    return hipGetLastError();
}
hipError_t launch_fq_name(const uint32_t *d_text, uint32_t n, const NameEntry *tab, uint32_t mask, uint32_t *h_ids, uint32_t *h_unknown, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fq_name, dim3((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), dim3(256), 0, s, d_text, n, tab, mask, h_ids, h_unknown);
    return hipGetLastError();
}
hipError_t launch_names_insert(const NamePub *pubs, uint32_t n, NameEntry *tab, uint32_t mask, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_names_insert, dim3((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), dim3(256), 0, s, pubs, n, tab, mask);
    return hipGetLastError();
}

// counters renumbered: record i < n_perm of `src` (4 x u64, three live words) is added to record perm[i] of `dst` (several may name one:
// atomics), the records from n_perm on keep their places.  Dictionaries of different GPUs number the barcodes in their own order:
// hast_counts_permute brings every context's counters into the first dictionary's numbering before the one all-reduce.  (Lives here,
// with the dictionary, not with the classification kernels.)
__global__ void __launch_bounds__(256) k_counts_permute(unsigned long long *dst, const unsigned long long *src, const uint32_t *perm, size_t n_perm, size_t n_old, size_t n_new) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_old; i += (size_t)gridDim.x * blockDim.x) {
        const size_t to = i < n_perm ? (size_t)perm[i] : i;
        if (to >= n_new) continue;
        for (int w = 0; w < 3; ++w) {
            const unsigned long long v = src[4 * i + w];
            if (v) atomicAdd(&dst[4 * to + w], v);
        }
    }
}
hipError_t launch_counts_permute(unsigned long long *d_dst, const unsigned long long *d_src, const uint32_t *d_perm, size_t n_perm, size_t n_old, size_t n_new, hipStream_t s) {
    if (n_old == 0) return hipSuccess;
    const size_t blocks = (n_old + 256 * 16 - 1) / (256 * 16);
    hipLaunchKernelGGL(k_counts_permute, dim3((unsigned)(blocks < 1 ? 1 : blocks > 65535 ? 65535 : blocks)), dim3(256), 0, s, d_dst, d_src, d_perm, n_perm, n_old, n_new);
    return hipGetLastError();
}

hipError_t launch_fq_block(uint8_t *d_buf, FqState *d_st, const uint8_t *d_prev_buf, const FqState *d_prev_st, uint64_t pad, uint64_t n_bytes,
                           uint32_t *d_tile_cnt, uint32_t *d_nl, uint64_t *d_off, uint32_t *d_len, uint32_t *d_bc_pos, uint32_t *d_bc_len,
                           uint32_t *h_bc, uint32_t *d_text, uint32_t h_cap, uint32_t k, int last, hipStream_t s) {
    hipError_t e = hipMemsetAsync(d_st, 0, sizeof(FqState), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fq_begin, dim3(1), dim3(256), 0, s, d_buf, d_st, d_prev_buf, d_prev_st, pad, n_bytes);
    const uint32_t n_tiles = (uint32_t)((pad + n_bytes + kFqTile - 1) / kFqTile);
    hipLaunchKernelGGL(k_fq_count, dim3(n_tiles), dim3(256), 0, s, d_buf, d_st, d_tile_cnt);
    hipLaunchKernelGGL(k_fq_scan, dim3(1), dim3(1024), 0, s, d_tile_cnt, n_tiles, d_st);
    hipLaunchKernelGGL(k_fq_index, dim3(n_tiles), dim3(256), 0, s, d_buf, d_st, d_tile_cnt, d_nl);
    hipLaunchKernelGGL(k_fq_records, dim3(2048), dim3(256), 0, s, d_buf, d_st, d_nl, d_off, d_len, d_bc_pos, d_bc_len, h_bc, d_text, h_cap, k, last);
    return hipGetLastError();
}

hipError_t launch_fq_count_own(const uint8_t *d_buf, FqState *d_st, uint64_t pad, uint64_t n_bytes, uint32_t *d_tile_cnt, hipStream_t s) {
    hipError_t e = hipMemsetAsync(d_st, 0, sizeof(FqState), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fq_view, dim3(1), dim3(1), 0, s, d_st, pad, pad + n_bytes, pad + n_bytes, 0u, 0u, 0u);
    const uint32_t n_tiles = (uint32_t)((pad + n_bytes + kFqTile - 1) / kFqTile);
    hipLaunchKernelGGL(k_fq_count, dim3(n_tiles), dim3(256), 0, s, d_buf, d_st, d_tile_cnt);
    hipLaunchKernelGGL(k_fq_scan, dim3(1), dim3(1024), 0, s, d_tile_cnt, n_tiles, d_st);
    return hipGetLastError();
}

hipError_t launch_fq_block_striped(const uint8_t *d_buf, FqState *d_st, uint64_t pad, uint64_t n_bytes, uint64_t n_over, uint32_t phase, int bol, int eof,
                                   uint32_t *d_tile_cnt, uint32_t *d_nl, uint64_t *d_off, uint32_t *d_len, uint32_t *d_bc_pos, uint32_t *d_bc_len,
                                   uint32_t *h_bc, uint32_t *d_text, uint32_t h_cap, uint32_t k, hipStream_t s) {
    hipError_t e = hipMemsetAsync(d_st, 0, sizeof(FqState), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fq_view, dim3(1), dim3(1), 0, s, d_st, pad, pad + n_bytes + n_over, pad + n_bytes, phase, bol ? 1u : 0u, eof ? 1u : 0u);
    const uint32_t n_tiles = (uint32_t)((pad + n_bytes + n_over + kFqTile - 1) / kFqTile);
    hipLaunchKernelGGL(k_fq_count, dim3(n_tiles), dim3(256), 0, s, d_buf, d_st, d_tile_cnt);
    hipLaunchKernelGGL(k_fq_scan, dim3(1), dim3(1024), 0, s, d_tile_cnt, n_tiles, d_st);
    hipLaunchKernelGGL(k_fq_index, dim3(n_tiles), dim3(256), 0, s, d_buf, d_st, d_tile_cnt, d_nl);
    hipLaunchKernelGGL(k_fq_records_striped, dim3(2048), dim3(256), 0, s, d_buf, d_st, d_nl, d_off, d_len, d_bc_pos, d_bc_len, h_bc, d_text, h_cap, k);
    return hipGetLastError();
}

}  // namespace hast

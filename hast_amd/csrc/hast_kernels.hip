// hast_kernels.hip -- gfx950 (MI355X, CDNA4) device code for HAST stage-01 read classification.
//
// Written for wave64 / 256 CUs / HBM3E directly: no CUDA-compat macros, no multi-backend paths.
// The workload is integer + random 64-B HBM reads (no MFMA anywhere by design):
//   K1  table build   : k_insert_keys / k_insert_text / k_erase_keys / k_count_tags
//   K3  classify      : k_classify   (round 1's hot kernel: every window probes the exact table; today the fallback when HBM
//                                       has no room for the filter, and HAST_CLASSIFY=exact -- the hot kernel is
//                                       k_classify_f in hast_filter.hip), k_commit_votes (per-barcode bookkeeping), the
//                                       segmentation of long reads
//   synthetic inputs  : k_synth_keys / k_synth_reads
//
// Reference semantics implemented (paths under /root/reference/01.classify_stlfr_reads/):
//   classify.cpp:30-46 (load_kmers), :182-209 (containN + process_reads), :314-339 (InitAdaptor),
//   kmer/kmer.h:11,153-166,169-194 (coding, str2Kmer, chopRead2Kmer).
#include "hast_common.h"
#include "hast_device.h"
#include "hast_devutil.h"

namespace hast {

// ------------------------------------------------------------------------------------------
// Table build.  slot = (key << 2) | tags, empty = all ones.  A key lives in the first bucket, in
// probe order from its home bucket, that had a free slot when it was inserted; slots never become
// empty again (erase only clears tag bits), so lookups may stop at the first bucket with an empty slot.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t *hap_table(uint64_t *slots, const TableGeom g, uint32_t hap) {
    return g.wide ? slots + (size_t)hap * g.nbuckets * kSlotsPerBucket : slots;
}
__device__ __forceinline__ bool slot_has_key(unsigned long long slot, uint64_t key, int wide) {
    return wide ? slot == key : (slot >> 2) == key;      // a canonical key never equals the empty/tombstone patterns
}

__device__ __forceinline__ bool table_insert(uint64_t *slots, const TableGeom g, uint64_t key, uint32_t hap) {
    const uint32_t nbuckets = g.nbuckets;
    const uint32_t tag = 1u << hap;
    const uint64_t want = g.wide ? key : ((key << 2) | tag);
    uint64_t *tab = hap_table(slots, g, hap);
    uint32_t b = home_bucket(key, g.k, g.m, nbuckets);
    for (uint32_t probe = 0; probe < nbuckets; ++probe) {
        unsigned long long *bs = reinterpret_cast<unsigned long long *>(tab) + (size_t)b * kSlotsPerBucket;
        for (int i = 0; i < kSlotsPerBucket; ++i) {
            unsigned long long cur = __hip_atomic_load(&bs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                if (cur == kEmptySlot) {
                    unsigned long long old = atomicCAS(&bs[i], (unsigned long long)kEmptySlot, (unsigned long long)want);
                    if (old == kEmptySlot) return true;
                    cur = old;               // someone else took the slot: look at what they wrote
                    continue;
                }
                if (slot_has_key(cur, key, g.wide)) {
                    if (!g.wide && (cur & tag) == 0) atomicOr(&bs[i], (unsigned long long)tag);
                    return true;
                }
                break;
            }
        }
        b = next_bucket(b, probe + 1, key, nbuckets);
    }
    return false;
}

// returns the address of the slot holding `key` in table `tab`, or nullptr
__device__ __forceinline__ unsigned long long *table_find(uint64_t *tab, const TableGeom g, uint64_t key) {
    const uint32_t nbuckets = g.nbuckets;
    uint32_t b = home_bucket(key, g.k, g.m, nbuckets);
    for (uint32_t probe = 0; probe < nbuckets; ++probe) {
        unsigned long long *bs = reinterpret_cast<unsigned long long *>(tab) + (size_t)b * kSlotsPerBucket;
        bool any_empty = false;
        for (int i = 0; i < kSlotsPerBucket; ++i) {
            unsigned long long cur = bs[i];
            if (cur == kEmptySlot) any_empty = true;
            else if (slot_has_key(cur, key, g.wide)) return &bs[i];
        }
        if (any_empty) return nullptr;
        b = next_bucket(b, probe + 1, key, nbuckets);
    }
    return nullptr;
}

__global__ void __launch_bounds__(256) k_insert_keys(uint64_t *slots, TableGeom g, const uint64_t *keys,
                                                     size_t n, uint32_t hap, uint32_t *err) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (!table_insert(slots, g, keys[i], hap)) atomicOr(&err[0], 1u);
}

// load_kmers (classify.cpp:30-46): line i = text[i*(K+1) .. +K), text[i*(K+1)+K] must be '\n'.
// err bit0: table full, bit1: a line is not exactly K bytes, bit2 (acgt_only: the stage-03 classifier compares k-mers as
// case-sensitive strings, S03/src_main/classify.cpp:59-65, so its key files must be what jellyfish / meryl dump): a byte
// of a line is not one of 'A','C','G','T'.
__global__ void __launch_bounds__(256) k_insert_text(uint64_t *slots, TableGeom g, const char *text,
                                                     size_t n_lines, uint32_t hap, int acgt_only, uint32_t *err) {
    const int k = g.k;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x) {
        const char *s = text + i * (size_t)(k + 1);
        uint64_t w = 0;
        bool bad = s[k] != '\n', other = false;
        for (int j = 0; j < k; ++j) {
            uint32_t c = (uint8_t)s[j];
            bad |= (c == '\n');
            other |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
            w = (w << 2) | base_code(c);
        }
        if (bad) { atomicOr(&err[0], 2u); continue; }
        if (acgt_only && other) { atomicOr(&err[0], 4u); continue; }
        if (!table_insert(slots, g, kmer_canon(w, k), hap)) atomicOr(&err[0], 1u);
    }
}

// InitAdaptor (classify.cpp:314-339): remove each key from both sets; report which sets had it.
__global__ void k_erase_keys(uint64_t *slots, TableGeom g, const uint64_t *keys, size_t n, uint8_t *hit) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!g.wide) {
        unsigned long long *p = table_find(slots, g, keys[i]);
        unsigned long long old = p ? atomicAnd(p, ~3ull) : 0ull;
        hit[i] = (uint8_t)(old & 3);
        return;
    }
    uint32_t h = 0;
    for (uint32_t hap = 0; hap < 2; ++hap) {
        unsigned long long *p = table_find(hap_table(slots, g, hap), g, keys[i]);
        if (p && atomicCAS(p, (unsigned long long)keys[i], (unsigned long long)kTombSlot) == keys[i]) h |= 1u << hap;
    }
    hit[i] = (uint8_t)h;
}

__global__ void k_lookup_keys(uint64_t *slots, TableGeom g, const uint64_t *keys, size_t n, uint8_t *tags) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!g.wide) {
        unsigned long long *p = table_find(slots, g, keys[i]);
        tags[i] = p ? (uint8_t)(*p & 3) : 0;
        return;
    }
    uint32_t t = 0;
    for (uint32_t hap = 0; hap < 2; ++hap)
        if (table_find(hap_table(slots, g, hap), g, keys[i])) t |= 1u << hap;
    tags[i] = (uint8_t)t;
}

// Table export/import (binary key-set cache): the live slots (key<<2|tags, tags != 0) are compacted
// into a dense array; importing re-inserts them (placement is recomputed, so K, not the geometry, must match).
__global__ void __launch_bounds__(256) k_export_slots(const uint64_t *slots, size_t nslots, uint64_t *out, size_t cap,
                                                      unsigned long long *counter) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nslots; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t s = slots[i];
        if (s != kEmptySlot && (s & 3)) {
            const unsigned long long at = atomicAdd(counter, 1ull);
            if (at < cap) out[at] = s;
        }
    }
}
__global__ void __launch_bounds__(256) k_import_slots(uint64_t *slots, TableGeom g, const uint64_t *in, size_t n, uint32_t *err) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t s = in[i];
        if ((s & 3) == 0 || (s >> 2) >> (2 * g.k)) { atomicOr(&err[0], 2u); continue; }     // not a slot image for this K
        bool ok = true;
        if (s & 1) ok = table_insert(slots, g, s >> 2, 0);
        if (ok && (s & 2)) ok = table_insert(slots, g, s >> 2, 1);
        if (!ok) atomicOr(&err[0], 1u);
    }
}

// g_kmers[h].size(): number of slots with tag bit h.  Streaming 16 B/lane.
__global__ void __launch_bounds__(256) k_count_tags(const uint64_t *slots, size_t nslots, unsigned long long *out, int wide) {
    unsigned long long c0 = 0, c1 = 0;
    const ulonglong2 *v = reinterpret_cast<const ulonglong2 *>(slots);
    size_t n2 = nslots / 2;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        ulonglong2 s = v[i];
        if (wide) {        // tag-less tables: hap 0 in the first half of the slots, hap 1 in the second; count live slots
            const unsigned long long live = (unsigned long long)(s.x < kTombSlot) + (unsigned long long)(s.y < kTombSlot);
            if (2 * i < nslots / 2) c0 += live; else c1 += live;
            continue;
        }
        if (s.x != kEmptySlot) { c0 += s.x & 1; c1 += (s.x >> 1) & 1; }
        if (s.y != kEmptySlot) { c0 += s.y & 1; c1 += (s.y >> 1) & 1; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        c0 += __shfl_down(c0, off, 64);
        c1 += __shfl_down(c1, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (c0) atomicAdd(&out[0], c0);
        if (c1) atomicAdd(&out[1], c1);
    }
}

// ------------------------------------------------------------------------------------------
// K3: classify.
//
// Workgroup = 256 threads = 4 wave64; it walks tiles of TR reads through LDS:
//   A  pack   : each lane turns 16 ASCII bases (aligned dword loads + v_alignbyte) into 32 bits of
//               2-bit codes ((c&6)>>1, kmer.h:11), first base most significant, and flags reads that
//               contain 'N' (classify.cpp:182-185).
//   M  m-mers : one lane per m-mer position: hash32(canonical m-mer) -> LDS.  The minimum over the
//               K-m+1 m-mers of a window is that window's minimizer hash = its home bucket
//               (hast_common.h), so consecutive windows mostly probe the same 64-B line.
//   B  probe  : (read, offset) windows are flattened; every LANE owns one window: funnel-shift it out
//               of two LDS words (no rolling state), canonicalise (v_bfrev), window-min of the m-mer
//               hashes.  Then kLPB rounds: in round j each GROUP of kLPB lanes (a pair by default, a quad
//               with -DHAST_LPB=4) takes the window of its lane j (DPP quad_perm broadcast) and its lanes
//               load the 4 x 16 B of that window's bucket between them, i.e. the 64-B line is fetched
//               once, by adjacent lanes, and adjacent groups = consecutive windows (same line when they
//               share a minimizer).  Blocks are software-pipelined: 4..8 loads in flight per lane.  A bucket is full iff its last
//               slot is taken (slots fill in order), so "no match and slot.y of some lane empty"
//               ends the probe; full buckets without a match (rare at load factor 0.2) take the
//               chain walk.  Hits (about 1 % of windows) go to per-read LDS counters.
//   C  votes  : one lane per read stores the read's {vote0, vote1} (8 B, coalesced).  The per-barcode bookkeeping
//               (classify.cpp:203-208) is a kernel of its own, k_commit_votes: random read-modify-writes that are
//               interleaved with the probes' random reads cost 4x what they cost on their own (DRAM bus turnaround;
//               tools/atomics_probe.hip: 16M updates beside 640M line reads +2.7 ms, alone 0.7 ms).
// ------------------------------------------------------------------------------------------
constexpr int kThreads = 256;
#ifndef HAST_MINWAVES
#define HAST_MINWAVES 5   // 96 VGPRs: 5 waves/SIMD = 5 workgroups per CU (measured best of 4/5/6/8 with the tile queue)
#endif
#ifndef HAST_LPB
#define HAST_LPB 2        // measured on C3: pairs 15.4 ms vs quads 16.2 ms per 16M reads (fewer DPP/compare rounds)
#endif
constexpr int kLPB = HAST_LPB;                // lanes that share one bucket (4: 16 B each, 2: 32 B each)
constexpr int kRounds = kLPB;                 // rounds per 64-window block (64/kLPB windows per round)
constexpr int kNLd = kPieces / kLPB;          // 16-B loads per lane per round
constexpr uint32_t kGrpMask = (1u << kLPB) - 1;
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// value of lane J of my group of kLPB lanes (DPP quad_perm: [J,J,J,J] for quads, [J,J,2+J,2+J] for pairs)
template <int J>
__device__ __forceinline__ uint32_t group_bcast(uint32_t v) {
    constexpr int ctrl = kLPB == 4 ? J * 0x55 : (J | (J << 2) | ((2 + J) << 4) | ((2 + J) << 6));
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, ctrl, 0xF, 0xF, true);
}

// WT  = number of m-mers per window (K-m+1) when known at compile time, 0 = runtime loop
// FAST = all index divisions are exact multiply-high (host-checked); false = plain division
// STRICT = per-window validity instead of the whole-read N skip: a window counts only when all its K bytes
//          are upper-case A/C/G/T (the string semantics of the stage-03 per-read classifier,
//          03.mkoutput_by_fabulous2.0/src_main/classify.cpp:209-214); rows may then be SEGMENTS of long reads
// WIDE = K == 32: bare 64-bit keys in two per-haplotype tables, probed one after the other
template <int WT, bool FAST, bool STRICT, bool WIDE>
__global__ void __launch_bounds__(kThreads, HAST_MINWAVES) k_classify(ClassifyArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t TR = a.tile_reads;
    const uint32_t WS = a.w64 + 1;                                   // LDS words per read incl. pad
    const uint32_t MS = a.mh_stride;                                 // m-mer positions per read (stride)
    unsigned long long *s_tile = reinterpret_cast<unsigned long long *>(smem);            // next tile of this workgroup
    unsigned long long *s_pack = s_tile + 2;                                               // [TR][WS]
    unsigned long long *s_vote = s_pack + (size_t)TR * WS;                                 // [TR]
    unsigned long long *s_off = s_vote + TR;                                               // [TR]
    uint32_t *s_len = reinterpret_cast<uint32_t *>(s_off + TR);                            // [TR]
    uint32_t *s_flag = s_len + TR;                                                         // [TR]
    uint32_t *s_mh = s_flag + TR;                                                          // [TR][MS] (+ W pad)
    const uint32_t IW = 2 * a.w64 + 1;                                                     // invalid-byte mask words per read
    uint32_t *s_inv = s_mh + (size_t)TR * MS + 16;                                         // [TR][IW], STRICT only

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wave = tid >> 6;
    const uint32_t sub = tid & (kLPB - 1);       // which 16-B piece(s) of a bucket this lane loads: sub, sub+kLPB, ..
    // lane -> window inside a 64-window block, so that in round j adjacent groups hold consecutive windows
    // (each half-wave owns 32 CONSECUTIVE windows, so its ds_read_b32 of the m-mer hashes hit 32 different banks)
    const uint32_t wofs = 32 * (lane >> 5) + (32 / kLPB) * sub + ((lane & 31) / kLPB);
    const uint32_t gsh = lane & ~(uint32_t)(kLPB - 1);               // first lane of my group
    const int K = a.k, M = a.m;
    const uint32_t W = WT ? (uint32_t)WT : (uint32_t)(K - M + 1);
    const uint32_t kshift = 64 - 2 * K, mshift = 64 - 2 * M;
    const uint32_t nb = a.nbuckets;
    const uint64_t n_rows = a.n_rows_ptr ? min((uint64_t)*a.n_rows_ptr, (uint64_t)a.n_reads) : a.n_reads;   // (never past the table the host sized)
    const uint64_t n_tiles = (n_rows + TR - 1) / TR;
    const uintptr_t base_addr = reinterpret_cast<uintptr_t>(a.bases);
    const uintptr_t end_addr = (base_addr + a.bases_bytes + 3) & ~(uintptr_t)3;
    const u64x2 *tab0 = reinterpret_cast<const u64x2 *>(a.slots) + sub;    // this lane's first 16-B piece of bucket 0

    // Tiles are handed out by a global queue (one atomic per tile), so the load stays balanced whatever the
    // residency of the grid is (the grid may be larger than what fits the chip at once).
    if (tid == 0) *s_tile = atomicAdd(a.tile_queue, 1ull);
    __syncthreads();
    for (;;) {
        const uint64_t tile = *s_tile;
        if (tile >= n_tiles) break;
        const uint64_t r0 = tile * TR;
        const uint32_t tra = (uint32_t)((n_rows - r0 < TR) ? (n_rows - r0) : TR);

        // ---- per-read header --------------------------------------------------------------
        if (tid < tra) {
            uint64_t off, len;
            if (a.offsets) { off = a.offsets[r0 + tid]; len = a.lens ? a.lens[r0 + tid] : a.offsets[r0 + tid + 1] - off; }
            else           { off = (r0 + tid) * (uint64_t)a.read_len; len = a.read_len; }
            if (len > a.read_len) len = a.read_len;          // contract: read_len bounds every read
            if (a.offsets && off > a.bases_bytes) len = 0;   // an offset outside the buffer (a caller's bug): no windows, no loads out of bounds
            s_off[tid] = off;
            s_len[tid] = (uint32_t)len;
            s_flag[tid] = 0;
            s_vote[tid] = 0;
        }
        __syncthreads();
        // everyone has read `tile`: fetch the next one now; the barriers below publish it before the loop top reads it
        if (tid == 0) *s_tile = atomicAdd(a.tile_queue, 1ull);

        // ---- A: pack ----------------------------------------------------------------------
        const uint32_t HW = a.w64 * 2;                                // 16-base half-words per read
        for (uint32_t t = tid; t < tra * HW; t += kThreads) {
            const uint32_t r = FAST ? __umulhi(t, a.div_hw) : (t / HW);
            const uint32_t j = t - r * HW;
            const uint32_t len = s_len[r];
            if (16 * j >= len) continue;
            const uint32_t nbases = (len - 16 * j < 16) ? (len - 16 * j) : 16;
            const uintptr_t addr = base_addr + s_off[r] + 16 * j;
            const uintptr_t a4 = addr & ~(uintptr_t)3;
            const uint32_t bsh = (uint32_t)(addr & 3);
            uint32_t d[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                uintptr_t p = a4 + 4 * i;
                d[i] = (p < end_addr) ? *reinterpret_cast<const uint32_t *>(p) : 0x41414141u;
            }
            uint32_t packed = 0, nflag = 0, invalid = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t x = __builtin_amdgcn_alignbyte(d[i + 1], d[i], bsh);   // bytes addr+4i .. +4i+3
                int vb = (int)nbases - 4 * i;                                    // valid bytes in x
                if (vb < 4) {
                    uint32_t m = (vb <= 0) ? 0u : ((1u << (8 * vb)) - 1u);
                    x = (x & m) | (0x41414141u & ~m);                            // pad with 'A'
                }
                if (STRICT) invalid = (invalid << 4) | not_acgt4(x);
                else nflag |= has_byte_N(x);
                packed = (packed << 8) | pack4(x);
            }
            // 64-bit LDS word w = bases 32w..32w+31, first base most significant: the even
            // half-word is the HIGH 32 bits (little-endian: +4 bytes)
            uint32_t *dst = reinterpret_cast<uint32_t *>(s_pack + (size_t)r * WS + (j >> 1)) + (1 - (j & 1));
            *dst = packed;
            if (STRICT) {
                // 32-bit mask word w = bases 32w..32w+31, first base most significant: even half-word = HIGH 16 bits
                uint16_t *di = reinterpret_cast<uint16_t *>(s_inv + (size_t)r * IW + (j >> 1)) + (1 - (j & 1));
                *di = (uint16_t)invalid;
            } else if (nflag) atomicOr(&s_flag[r], 1u);
        }
        __syncthreads();

        // ---- M: m-mer hashes; reads with 'N' get length 0 (whole-read skip, classify.cpp:190-193) ----
        for (uint32_t t = tid; t < tra * MS; t += kThreads) {
            const uint32_t r = FAST ? __umulhi(t, a.div_mh) : (t / MS);
            const uint32_t q = t - r * MS;
            const uint64_t mm = window_bits(s_pack + (size_t)r * WS, q, mshift);
            const uint32_t h = mmer_hash32(kmer_canon(mm, M));
            s_mh[t] = (q + M <= s_len[r]) ? h : 0xFFFFFFFFu;
        }
        __syncthreads();
        if (!STRICT) {
            if (tid < tra && s_flag[tid]) s_len[tid] = 0;
            __syncthreads();
        }

        // ---- B: probe.  Each wave walks the 64-window blocks wave, wave+4, ... -----------------------
        const uint32_t P = a.max_pos;                                 // windows per read (stride)
        const uint32_t Q = tra * P;
        const uint32_t nblk = (Q + 63) >> 6;
        // Software pipeline over the wave's blocks: the loads of block i+1 are issued (after its windows are
        // computed) BEFORE block i is compared, so a wave always has 4..8 bucket loads in flight while it does
        // VALU work, instead of alternating "compute with nothing in flight" and "wait".
        const u64x2 *tab = tab0;                        // table probed in this pass (WIDE: one pass per haplotype)
        uint32_t pass_tags = 0;                         // WIDE: the tag a match in this pass stands for
        struct Blk {
            uint32_t klo, khi, bkt, meta;               // this lane's own window
            u64x2 sl[kRounds][kNLd];                    // the bucket pieces this lane loaded, per round
        };
        auto start = [&](Blk &B, uint32_t blk) {
            const uint32_t q = blk * 64 + wofs;
            uint32_t r = FAST ? __umulhi(q, a.div_magic) : (q / P);
            const bool inq = q < Q;
            r = inq ? r : 0;
            const uint32_t p = inq ? q - r * P : 0;
            bool ok = inq && (p + K <= s_len[r]);
            if (STRICT) {           // any byte of [p, p+K) outside ACGT => the window cannot match a stored string
                const uint32_t *iw = s_inv + r * IW + (p >> 5);
                const unsigned long long bits = (((unsigned long long)iw[0] << 32) | iw[1]) << (p & 31);
                ok = ok && (bits >> (64 - K)) == 0;
            }
            const uint64_t ck = kmer_canon(window_bits(s_pack + (size_t)r * WS, p, kshift), K);
            const uint32_t *mh = s_mh + r * MS + p;
            uint32_t mn = mh[0];
            if (WT) {
#pragma unroll
                for (int j = 1; j < (WT ? WT : 1); ++j) mn = min(mn, mh[j]);
            } else {
                for (uint32_t j = 1; j < W; ++j) mn = min(mn, mh[j]);
            }
            B.klo = WIDE ? (uint32_t)ck : (uint32_t)(ck << 2);        // the slot value without tags (WIDE: the bare key)
            B.khi = WIDE ? (uint32_t)(ck >> 32) : (uint32_t)(ck >> 30);
            B.bkt = ok ? bucket_of(mn, ck, nb) : 0;                  // invalid windows read bucket 0 (harmless)
            B.meta = r | (ok ? 0x80000000u : 0u);
            uint32_t bk[kRounds];
            bk[0] = group_bcast<0>(B.bkt);
            bk[1] = group_bcast<1>(B.bkt);
            if (kRounds == 4) {
                bk[kRounds - 2] = group_bcast<kRounds == 4 ? 2 : 0>(B.bkt);
                bk[kRounds - 1] = group_bcast<kRounds == 4 ? 3 : 1>(B.bkt);
            }
#pragma unroll
            for (int j = 0; j < kRounds; ++j)
#pragma unroll
                for (int l = 0; l < kNLd; ++l) B.sl[j][l] = tab[(size_t)bk[j] * kPieces + l * kLPB];
        };
        auto finish = [&](Blk &B) {
            // the window's key and read are re-broadcast here rather than kept live across the loads
            uint32_t kl[kRounds], kh[kRounds], mt[kRounds];
            kl[0] = group_bcast<0>(B.klo); kh[0] = group_bcast<0>(B.khi); mt[0] = group_bcast<0>(B.meta);
            kl[1] = group_bcast<1>(B.klo); kh[1] = group_bcast<1>(B.khi); mt[1] = group_bcast<1>(B.meta);
            if (kRounds == 4) {
                constexpr int J2 = kRounds == 4 ? 2 : 0, J3 = kRounds == 4 ? 3 : 1;
                kl[kRounds - 2] = group_bcast<J2>(B.klo); kh[kRounds - 2] = group_bcast<J2>(B.khi); mt[kRounds - 2] = group_bcast<J2>(B.meta);
                kl[kRounds - 1] = group_bcast<J3>(B.klo); kh[kRounds - 1] = group_bcast<J3>(B.khi); mt[kRounds - 1] = group_bcast<J3>(B.meta);
            }
            uint32_t hitmask = 0, fullmask = 0;
#pragma unroll
            for (int j = 0; j < kRounds; ++j) {
                const unsigned long long kq = ((unsigned long long)kh[j] << 32) | kl[j];
                const bool valid = (int)mt[j] < 0;
                bool m = false;
#pragma unroll
                for (int l = 0; l < kNLd; ++l) m = m || (WIDE ? B.sl[j][l].x : (B.sl[j][l].x & ~3ull)) == kq || (WIDE ? B.sl[j][l].y : (B.sl[j][l].y & ~3ull)) == kq;
                if (valid && m) hitmask |= 1u << j;
                // bucket full <=> its last slot (last lane of the group, last piece, .y) is taken: slots fill in order
                if (valid && B.sl[j][kNLd - 1].y != kEmptySlot && sub == kLPB - 1) fullmask |= 1u << j;
            }
            if (hitmask) {                                      // ~1 % of windows
#pragma unroll
                for (int j = 0; j < kRounds; ++j)
                    if (hitmask & (1u << j)) {
                        const unsigned long long kq = ((unsigned long long)kh[j] << 32) | kl[j];
                        unsigned long long hit_slot = 0;
#pragma unroll
                        for (int l = 0; l < kNLd; ++l) {
                            if ((WIDE ? B.sl[j][l].x : (B.sl[j][l].x & ~3ull)) == kq) hit_slot = B.sl[j][l].x;
                            if ((WIDE ? B.sl[j][l].y : (B.sl[j][l].y & ~3ull)) == kq) hit_slot = B.sl[j][l].y;
                        }
                        const uint32_t tags = WIDE ? pass_tags : (uint32_t)(hit_slot & 3);
                        atomicAdd(&s_vote[mt[j] & 0xFFFF], (unsigned long long)(tags & 1) | ((unsigned long long)(tags >> 1) << 32));
                    }
            }
            if (__any(fullmask != 0)) {                         // some bucket of this block is full (rare at LF 0.2)
                // group-uniform masks: a full bucket needs the chain walk unless some lane of the group hit
                uint32_t moremask = 0;
#pragma unroll
                for (int j = 0; j < kRounds; ++j) {
                    const unsigned long long f = __ballot((fullmask >> j) & 1), h = __ballot((hitmask >> j) & 1);
                    if (((f >> gsh) & kGrpMask) != 0 && ((h >> gsh) & kGrpMask) == 0) moremask |= 1u << j;
                }
                while (__any(moremask != 0)) {
                    const bool act = moremask != 0;
                    const uint32_t j = act ? (uint32_t)__ffs(moremask) - 1 : 0;
                    const int src = (int)(gsh | j);
                    const unsigned long long kq = ((unsigned long long)__shfl((int)B.khi, src) << 32) | (uint32_t)__shfl((int)B.klo, src);
                    uint32_t b = (uint32_t)__shfl((int)B.bkt, src);
                    const uint32_t rd = (uint32_t)__shfl((int)B.meta, src) & 0xFFFF;
                    bool pending = act;
                    uint32_t guard = 0;
                    while (__any(pending)) {
                        u64x2 s2[kNLd];
#pragma unroll
                        for (int l = 0; l < kNLd; ++l) s2[l] = u64x2{kEmptySlot, kEmptySlot};
                        if (pending) {
                            b = next_bucket(b, guard + 1, WIDE ? kq : (kq >> 2), nb);
#pragma unroll
                            for (int l = 0; l < kNLd; ++l) s2[l] = tab[(size_t)b * kPieces + l * kLPB];
                        }
                        unsigned long long hit_slot = 0;
                        bool h2 = false;
#pragma unroll
                        for (int l = 0; l < kNLd; ++l) {
                            if ((WIDE ? s2[l].x : (s2[l].x & ~3ull)) == kq) { hit_slot = s2[l].x; h2 = true; }
                            if ((WIDE ? s2[l].y : (s2[l].y & ~3ull)) == kq) { hit_slot = s2[l].y; h2 = true; }
                        }
                        h2 = h2 && pending;
                        // the bucket is not full iff its last slot is empty; any empty .y of the last piece implies it
                        const unsigned long long m2 = __ballot(h2 || (pending && s2[kNLd - 1].y == kEmptySlot));
                        if (h2) {
                            const uint32_t tags = WIDE ? pass_tags : (uint32_t)(hit_slot & 3);
                            atomicAdd(&s_vote[rd], (unsigned long long)(tags & 1) | ((unsigned long long)(tags >> 1) << 32));
                        }
                        if (((m2 >> gsh) & kGrpMask) != 0 || ++guard >= nb) pending = false;
                    }
                    moremask &= moremask - 1;
                }
            }
        };
        for (int pass = 0; pass < (WIDE ? 2 : 1); ++pass) {
            if (WIDE) {
                tab = tab0 + (size_t)pass * nb * kPieces;
                pass_tags = 1u << pass;
            }
            Blk A, B;
            uint32_t blk = wave;
            bool va = blk < nblk;
            if (va) start(A, blk);
            blk += 4;
            while (va) {
                const bool vb = blk < nblk;
                if (vb) start(B, blk);
                blk += 4;
                finish(A);
                if (!vb) break;
                va = blk < nblk;
                if (va) start(A, blk);
                blk += 4;
                finish(B);
            }
        }
        __syncthreads();

        // ---- C: the read's votes out (the per-barcode bookkeeping is k_commit_votes' job) ----------------
        if (tid < tra) {
            const unsigned long long v = s_vote[tid];
            if (a.votes) {
                if (a.seg_read) {                 // rows are segments of long reads: add into the read's (zeroed) row
                    uint32_t *row = a.votes + 2 * (size_t)a.seg_read[r0 + tid];
                    if ((uint32_t)v) atomicAdd(row, (uint32_t)v);
                    if ((uint32_t)(v >> 32)) atomicAdd(row + 1, (uint32_t)(v >> 32));
                } else {
                    reinterpret_cast<unsigned long long *>(a.votes)[r0 + tid] = v;       // {vote0, vote1}: one coalesced 8-byte store
                }
            }
        }
        // no barrier needed here: the next tile's header only touches this lane's own s_* entries
        // and is followed by a barrier before anyone else reads them.
    }
}

// ------------------------------------------------------------------------------------------
// synthetic workload; same integer functions as the host generator
// ------------------------------------------------------------------------------------------
__global__ void k_synth_keys(SynthParams p, int hap, uint64_t first, size_t n, uint64_t *out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = synth_key(p, hap, first + i);
}

__global__ void k_synth_reads(SynthParams p, uint64_t first, size_t n, uint8_t *bases, uint32_t *barcodes) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t bc;
        synth_read(p, first + i, bases + i * (size_t)p.read_len, &bc);
        if (barcodes) barcodes[i] = bc;
    }
}

// ------------------------------------------------------------------------------------------
// launch wrappers (host side, called from hast_api.cpp)
// ------------------------------------------------------------------------------------------
static inline int grid_for(size_t n, int block, int cap) {
    size_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g > (size_t)cap ? (size_t)cap : g);
}

hipError_t launch_insert_keys(uint64_t *slots, TableGeom g, const uint64_t *d_keys, size_t n, uint32_t hap,
                              uint32_t *d_err, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_insert_keys, dim3(grid_for(n, 256, 256 * 32)), dim3(256), 0, s, slots, g, d_keys, n, hap, d_err);
    return hipGetLastError();
}
hipError_t launch_insert_text(uint64_t *slots, TableGeom g, const char *d_text, size_t n_lines,
                              uint32_t hap, int acgt_only, uint32_t *d_err, hipStream_t s) {
    if (n_lines == 0) return hipSuccess;
    hipLaunchKernelGGL(k_insert_text, dim3(grid_for(n_lines, 256, 256 * 32)), dim3(256), 0, s, slots, g, d_text, n_lines, hap, acgt_only, d_err);
    return hipGetLastError();
}
hipError_t launch_erase_keys(uint64_t *slots, TableGeom g, const uint64_t *d_keys, size_t n, uint8_t *d_hit, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_erase_keys, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, slots, g, d_keys, n, d_hit);
    return hipGetLastError();
}
hipError_t launch_lookup_keys(uint64_t *slots, TableGeom g, const uint64_t *d_keys, size_t n, uint8_t *d_tags, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_lookup_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slots, g, d_keys, n, d_tags);
    return hipGetLastError();
}
hipError_t launch_export_slots(const uint64_t *slots, size_t nslots, uint64_t *d_out, size_t cap, unsigned long long *d_counter, hipStream_t s) {
    hipLaunchKernelGGL(k_export_slots, dim3(grid_for(nslots, 256, 256 * 16)), dim3(256), 0, s, slots, nslots, d_out, cap, d_counter);
    return hipGetLastError();
}
hipError_t launch_import_slots(uint64_t *slots, TableGeom g, const uint64_t *d_in, size_t n, uint32_t *d_err, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_import_slots, dim3(grid_for(n, 256, 256 * 32)), dim3(256), 0, s, slots, g, d_in, n, d_err);
    return hipGetLastError();
}
hipError_t launch_count_tags(const uint64_t *slots, TableGeom g, unsigned long long *d_out, hipStream_t s) {
    const size_t nslots = (size_t)g.nbuckets * kSlotsPerBucket * (g.wide ? 2 : 1);
    hipLaunchKernelGGL(k_count_tags, dim3(grid_for(nslots / 2, 256, 256 * 8)), dim3(256), 0, s, slots, nslots, d_out, g.wide);
    return hipGetLastError();
}
template <int WT, bool FAST, bool STRICT, bool WIDE = false>
static hipError_t launch_classify_t(const ClassifyArgs &a, int grid, size_t smem, hipStream_t s) {
    if (smem > (48u << 10)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_classify<WT, FAST, STRICT, WIDE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_classify<WT, FAST, STRICT, WIDE>), dim3(grid), dim3(kThreads), smem, s, a);
    return hipGetLastError();
}

template <bool STRICT>
static hipError_t launch_classify_s(const ClassifyArgs &a, int grid, size_t smem, hipStream_t s) {
    const int w = a.k - a.m + 1;
    const bool fast = a.div_magic && a.div_mh && a.div_hw;
    if (a.wide) return fast ? launch_classify_t<0, true, STRICT, true>(a, grid, smem, s) : launch_classify_t<0, false, STRICT, true>(a, grid, smem, s);
    if (!fast) return launch_classify_t<0, false, STRICT>(a, grid, smem, s);
    switch (w) {
    case 1: return launch_classify_t<1, true, STRICT>(a, grid, smem, s);
    case 2: return launch_classify_t<2, true, STRICT>(a, grid, smem, s);
    case 3: return launch_classify_t<3, true, STRICT>(a, grid, smem, s);
    case 4: return launch_classify_t<4, true, STRICT>(a, grid, smem, s);
    case 5: return launch_classify_t<5, true, STRICT>(a, grid, smem, s);
    case 6: return launch_classify_t<6, true, STRICT>(a, grid, smem, s);
    case 7: return launch_classify_t<7, true, STRICT>(a, grid, smem, s);
    case 8: return launch_classify_t<8, true, STRICT>(a, grid, smem, s);
    case 9: return launch_classify_t<9, true, STRICT>(a, grid, smem, s);
    default: return launch_classify_t<0, true, STRICT>(a, grid, smem, s);
    }
}

hipError_t launch_classify(const ClassifyArgs &a, int grid, size_t smem, hipStream_t s) {
    if (a.n_reads == 0) return hipSuccess;
    return a.strict ? launch_classify_s<true>(a, grid, smem, s) : launch_classify_s<false>(a, grid, smem, s);
}

// Long reads -> segments of at most seg_windows windows (consecutive segments overlap by K-1 bases), so that a
// row of the classify kernel always fits LDS.  Segment rows are handed out with one atomic per read.
// A workgroup takes 256 reads: every thread sizes one read, the workgroup reserves all their rows with ONE atomic (one atomic per
// read on the one counter word serialises at the memory side: 360k of them = 0.65-1.1 ms, 2-4 % of a config-5 step), then 16
// lanes write a read's rows side by side (a 20-kb read has 42).
__global__ void __launch_bounds__(256) k_build_segments(const uint64_t *offsets, const uint32_t *lens, size_t n_reads, int k, uint32_t seg_windows,
                                                        uint64_t *seg_off, uint32_t *seg_len, uint32_t *seg_read, unsigned long long *counter,
                                                        const uint8_t *skip, uint64_t fixed_len, uint64_t cap, uint64_t bases_bytes, uint32_t *err) {
    __shared__ unsigned long long s_off[256], s_len[256], s_first[256];
    __shared__ uint32_t s_nseg[256], s_wave[4];
    __shared__ unsigned long long s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const size_t i = blockIdx.x * (size_t)256 + tid;
    // offsets == nullptr: reads of fixed_len back to back; lens == nullptr: offsets holds n_reads+1 boundaries; else n_reads
    // starts + their lengths (reads framed out of a raw FASTQ block)
    uint64_t off = 0, len = 0;
    uint32_t nseg = 0;
    if (i < n_reads) {
        off = offsets ? offsets[i] : i * fixed_len;
        len = (skip && skip[i]) ? 0 : !offsets ? fixed_len : (lens ? (uint64_t)lens[i] : offsets[i + 1] - off);   // skipped read: no windows
        // offsets that run backwards or past the buffer (a caller's bug) must not become rows: the read gets none and the context's
        // error word says so (hast_stream_sync / hast_counts_read / the synchronous calls report it)
        if (off > bases_bytes || len > bases_bytes - off) { atomicOr(err, 1u); len = 0; }
        const uint64_t nwin = len >= (uint64_t)k ? len - k + 1 : 0;
        nseg = (uint32_t)(nwin ? (nwin + seg_windows - 1) / seg_windows : 1);
    }
    uint32_t incl = nseg;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= (uint32_t)o) incl += t;
    }
    if (lane == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = incl - nseg;
    for (uint32_t w = 0; w < (tid >> 6); ++w) before += s_wave[w];
    if (tid == 255) s_base = atomicAdd(counter, (unsigned long long)(before + nseg));
    __syncthreads();
    s_off[tid] = off;
    s_len[tid] = len;
    s_nseg[tid] = nseg;
    s_first[tid] = s_base + before;
    __syncthreads();
    const uint32_t grp = tid >> 4, gl = tid & 15;
    for (uint32_t r = grp; r < 256; r += 16) {
        const uint32_t ns = s_nseg[r];
        const uint64_t o = s_off[r], l = s_len[r], first = s_first[r];
        const uint32_t read = (uint32_t)(blockIdx.x * (size_t)256 + r);
        for (uint32_t j = gl; j < ns; j += 16) {
            const uint64_t start = (uint64_t)j * seg_windows;
            const uint64_t rest = l - start;
            const uint64_t sl = rest < (uint64_t)seg_windows + k - 1 ? rest : (uint64_t)seg_windows + k - 1;
            // reads that overlap make the lengths add up to more than the table was sized for: such rows are dropped, with the
            // same report
            if (first + j >= cap) { atomicOr(err, 1u); break; }
            seg_off[first + j] = o + start;
            seg_len[first + j] = (uint32_t)sl;
            seg_read[first + j] = read;
        }
    }
}

// containN (classify.cpp:182-185) for reads too long for one kernel row: one wave per read scans its bytes.
__global__ void __launch_bounds__(256) k_scan_n(const uint8_t *bases, const uint64_t *offsets, const uint32_t *lens, size_t n_reads, uint8_t *has_n, uint64_t fixed_len) {
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (wave >= n_reads) return;
    const uint64_t off = offsets ? offsets[wave] : wave * fixed_len, len = !offsets ? fixed_len : lens ? (uint64_t)lens[wave] : offsets[wave + 1] - off;
    bool found = false;
    for (uint64_t i = lane; i < len && !found; i += 64) found = bases[off + i] == 'N';
    const bool any = __any(found);
    if (lane == 0) has_n[wave] = any ? 1 : 0;
}
// process_reads' bookkeeping (classify.cpp:203-208) from per-read votes: barcode[0] += vote0, barcode[1] += vote1, or
// barcode[-1] += 1 when both are zero.  Counters are 64-bit words (u64 counts[n][4] = {c0, c1, neg, reserved}): the reference
// counts in `int` (classify.cpp:51), which "0_0_0" -- 10-20 % of real stLFR reads -- can overflow; here nothing wraps and
// nothing carries from one counter into the next, the narrowing is the caller's ("Hot barcode").
// A workgroup takes a span of kCommitSpan consecutive reads.  Agent-scope atomics execute at the memory side on this part,
// so a million adds to ONE record serialise (measured inside k_classify before the split: 16 barcodes -> 3x, 1 barcode ->
// 12x the kernel time).  Hence a small LDS cache keyed by barcode id: a barcode that holds one of its two candidate slots
// is summed in LDS and written out at the end of every kCommitFlush reads; everyone else does its own global atomics.
constexpr int kCommitSlots = 128;              // power of two
constexpr uint32_t kCommitSpan = 8192, kCommitFlush = 2048, kNoBarcode = 0xFFFFFFFFu;
// n_ptr != nullptr: the list's length is only known on the device (the overflow list of the partitioned commit below)
__global__ void __launch_bounds__(256) k_commit_votes(const uint32_t *votes, const uint32_t *barcode_ids, unsigned long long *counts,
                                                      uint32_t *votes_out, size_t n_reads_arg, const unsigned long long *n_ptr) {
    const size_t n_reads = n_ptr ? (size_t)*n_ptr : n_reads_arg;
    if ((size_t)blockIdx.x * kCommitSpan >= n_reads) return;                     // (wave-uniform)
    __shared__ unsigned long long s_v0[kCommitSlots], s_v1[kCommitSlots];
    __shared__ uint32_t s_neg[kCommitSlots], s_id[kCommitSlots];
    const uint32_t tid = threadIdx.x;
    auto flush = [&]() {
        if (tid < kCommitSlots) {
            if (s_id[tid] != kNoBarcode) {
                unsigned long long *rec = counts + 4 * (size_t)s_id[tid];
                if (s_v0[tid]) atomicAdd(rec, s_v0[tid]);
                if (s_v1[tid]) atomicAdd(rec + 1, s_v1[tid]);
                if (s_neg[tid]) atomicAdd(rec + 2, (unsigned long long)s_neg[tid]);
            }
            s_id[tid] = kNoBarcode;
            s_v0[tid] = 0;
            s_v1[tid] = 0;
            s_neg[tid] = 0;
        }
    };
    if (tid < kCommitSlots) {                  // (LDS starts out undefined: never flush before this)
        s_id[tid] = kNoBarcode;
        s_v0[tid] = 0;
        s_v1[tid] = 0;
        s_neg[tid] = 0;
    }
    __syncthreads();
    const size_t span0 = (size_t)blockIdx.x * kCommitSpan;
    for (uint32_t j = 0; j < kCommitSpan; j += 256) {
        const size_t i = span0 + j + tid;
        if (i < n_reads) {
            const unsigned long long v = reinterpret_cast<const unsigned long long *>(votes)[i];
            if (votes_out) reinterpret_cast<unsigned long long *>(votes_out)[i] = v;
            if (barcode_ids) {
                const uint32_t id = barcode_ids[i];
                const unsigned long long v0 = (uint32_t)v, v1 = v >> 32;
                const uint32_t h = id * 0x9E3779B1u;
                uint32_t slot = h >> (32 - 7);                                   // log2(kCommitSlots) = 7
                uint32_t owner = atomicCAS(&s_id[slot], kNoBarcode, id);
                if (owner != kNoBarcode && owner != id) {                        // second candidate
                    slot = (h >> 11) & (kCommitSlots - 1);
                    owner = atomicCAS(&s_id[slot], kNoBarcode, id);
                }
                if (owner == kNoBarcode || owner == id) {
                    if (v0) atomicAdd(&s_v0[slot], v0);
                    if (v1) atomicAdd(&s_v1[slot], v1);
                    if (!v) atomicAdd(&s_neg[slot], 1u);
                } else {
                    unsigned long long *rec = counts + 4 * (size_t)id;
                    if (v0) atomicAdd(rec, v0);
                    if (v1) atomicAdd(rec + 1, v1);
                    if (!v) atomicAdd(rec + 2, 1ull);
                }
            }
        }
        if (barcode_ids && (j + 256) % kCommitFlush == 0) {                      // give other barcodes a chance at the slots
            __syncthreads();
            flush();
            __syncthreads();
        }
    }
}
// ------------------------------------------------------------------------------------------
// Partitioned commit (large batches over many barcodes).  k_commit_votes does one memory-side atomic per read, and this part
// executes 27 G of them per second whatever their scope or footprint (tools/l2_atomics_probe): 48M reads = 2.0 ms, 7 % of a
// bench step.  Here the (barcode, votes) pairs are first PARTITIONED by barcode range -- in LDS, so that every bin receives
// whole runs of records, and compressed to 4 bytes (13 bits of barcode inside the bin, 8 + 8 bits of votes: a read of up to
// 255 windows) -- and then every bin is summed in LDS and added to its counters with plain coalesced read-modify-writes: the
// workgroup of a bin is the only one that touches those counters.  No atomic leaves the chip except one reservation per
// (workgroup, bin).  Bins have a fixed capacity; what does not fit (a barcode that owns a large share of the reads) goes to an
// overflow list that k_commit_pairs adds with atomics afterwards.  Same integer sums, any order.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kPartMaxSpanBits = 13;                                        // at most 8192 barcodes per bin (13 + 8 + 8 bits per record)
constexpr uint32_t kPartRecs = 16384;                                            // records a workgroup partitions at a time
constexpr int kPartThreads = 1024;
constexpr uint32_t kPartMaxBins = 4096;
__global__ void __launch_bounds__(kPartThreads) k_commit_partition(const unsigned long long *votes, const uint32_t *ids, size_t n, uint32_t n_bins,
                                                                  uint32_t span_bits, uint32_t cap, uint32_t *bin_fill, uint32_t *bin_valid, uint32_t *bin_recs,
                                                                  unsigned long long *over_n, uint32_t *over_ids, unsigned long long *over_votes) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(smem);                       // [n_bins] records of this span per bin
    uint32_t *s_off = s_cnt + n_bins;                                            // [n_bins] where the bin starts in s_rec
    uint32_t *s_dst = s_off + n_bins;                                            // [n_bins] where the bin's run goes (bit 31: overflow list)
    uint32_t *s_rec = s_dst + n_bins;                                            // [kPartRecs] compressed records, grouped by bin
    __shared__ uint32_t s_scan[kPartThreads / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t kPartSpanBits = span_bits, kPartSpan = 1u << span_bits;
    const size_t r0 = (size_t)blockIdx.x * kPartRecs;
    const uint32_t nr = (uint32_t)(n - r0 < kPartRecs ? n - r0 : kPartRecs);
    for (uint32_t b = tid; b < n_bins; b += kPartThreads) s_cnt[b] = 0;
    __syncthreads();
    constexpr int PER = kPartRecs / kPartThreads;                                // 16 records per thread
    uint32_t rec[PER], rank[PER];                                                // rec: bin << 29-bit payload is not stored: bin is id >> 13
    uint32_t bin_of[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const uint32_t i = (uint32_t)q * kPartThreads + tid;                     // coalesced
        bin_of[q] = 0xFFFFFFFFu;
        if (i < nr) {
            const uint32_t id = ids[r0 + i];
            const unsigned long long v = votes[r0 + i];
            bin_of[q] = id >> kPartSpanBits;
            if (bin_of[q] >= n_bins) { bin_of[q] = 0xFFFFFFFFu; continue; }       // an id outside the counters (the caller's contract): dropped
            rec[q] = (id & (kPartSpan - 1)) | ((uint32_t)v & 0xFFu) << 13 | ((uint32_t)(v >> 32) & 0xFFu) << 21;
            rank[q] = atomicAdd(&s_cnt[bin_of[q]], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of s_cnt over the bins (n_bins <= 4096 = 4 per thread)
    {
        uint32_t c[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t b = tid * 4 + q;
            c[q] = b < n_bins ? s_cnt[b] : 0;
            sum += c[q];
        }
        uint32_t incl = sum;
        const uint32_t lane = tid & 63;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= (uint32_t)off) incl += t;
        }
        if (lane == 63) s_scan[tid >> 6] = incl;
        __syncthreads();
        uint32_t base = incl - sum;
        for (uint32_t w = 0; w < (tid >> 6); ++w) base += s_scan[w];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t b = tid * 4 + q;
            if (b < n_bins) {
                s_off[b] = base;
                base += c[q];
                // one reservation per (workgroup, bin): a run of c[q] records in the bin, or -- when the bin is full -- in the overflow list
                uint32_t dst = 0;
                if (c[q]) {
                    const uint32_t g = atomicAdd(&bin_fill[b], c[q]);
                    if (g + c[q] <= cap) dst = g;
                    else {
                        atomicMin(&bin_valid[b], g);                             // records [0, first failed reservation) of a bin are real
                        dst = 0x80000000u | (uint32_t)atomicAdd(over_n, (unsigned long long)c[q]);
                    }
                }
                s_dst[b] = dst;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (bin_of[q] != 0xFFFFFFFFu) s_rec[s_off[bin_of[q]] + rank[q]] = rec[q] | 0u;
    __syncthreads();
    // runs out: a quarter of a wave per bin (a run holds ~13 records: 16384 records over ~1200 bins), 16 lanes over the run
    const uint32_t grp = tid >> 4, gl = tid & 15;
    for (uint32_t b = grp; b < n_bins; b += kPartThreads / 16) {
        const uint32_t cnt = s_cnt[b];
        if (!cnt) continue;
        const uint32_t dst = s_dst[b], off = s_off[b];
        if (!(dst & 0x80000000u)) {
            uint32_t *out = bin_recs + (size_t)b * cap + dst;
            for (uint32_t j = gl; j < cnt; j += 16) out[j] = s_rec[off + j];
        } else {
            const uint32_t o = dst & 0x7FFFFFFFu;
            for (uint32_t j = gl; j < cnt; j += 16) {
                const uint32_t r = s_rec[off + j];
                over_ids[o + j] = (b << kPartSpanBits) | (r & (kPartSpan - 1));
                over_votes[o + j] = (unsigned long long)((r >> 13) & 0xFFu) | ((unsigned long long)((r >> 21) & 0xFFu) << 32);
            }
        }
    }
}

// one workgroup per bin: sum its records in LDS, then add the sums to the bin's counters (nobody else touches them in this kernel)
constexpr int kBinThreads = 1024;
__global__ void __launch_bounds__(kBinThreads) k_commit_bins(const uint32_t *bin_recs, const uint32_t *bin_fill, const uint32_t *bin_valid, uint32_t cap,
                                                     uint32_t span_bits, unsigned long long *counts, size_t n_barcodes) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t kPartSpanBits = span_bits, kPartSpan = 1u << span_bits;
    unsigned long long *s_vote = reinterpret_cast<unsigned long long *>(smem);   // [kPartSpan] {c0, c1}: a bin holds <= cap records of <= 255 votes, cap * 255 < 2^32 (commit_partition_usable)
    uint32_t *s_neg = reinterpret_cast<uint32_t *>(s_vote + kPartSpan);          // [kPartSpan]
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t fill = bin_fill[b], valid = bin_valid[b];
    const uint32_t n = fill < valid ? (fill < cap ? fill : cap) : valid;
    if (n == 0) return;
    for (uint32_t j = tid; j < kPartSpan; j += kBinThreads) {
        s_vote[j] = 0;
        s_neg[j] = 0;
    }
    __syncthreads();
    const uint32_t *recs = bin_recs + (size_t)b * cap;
    for (uint32_t i = tid; i < n; i += kBinThreads) {
        const uint32_t r = recs[i];
        const uint32_t id = r & (kPartSpan - 1), v0 = (r >> 13) & 0xFFu, v1 = (r >> 21) & 0xFFu;
        if (v0 | v1) atomicAdd(&s_vote[id], (unsigned long long)v0 | ((unsigned long long)v1 << 32));
        else atomicAdd(&s_neg[id], 1u);
    }
    __syncthreads();
    const size_t first = (size_t)b << kPartSpanBits;
    for (uint32_t j = tid; j < kPartSpan && first + j < n_barcodes; j += kBinThreads) {
        const unsigned long long v = s_vote[j];
        const uint32_t g = s_neg[j];
        if (v | g) {                                                             // 64-bit counters: no wrap, no carry between them
            unsigned long long *rec = counts + 4 * (first + j);
            if (v) {
                ulonglong2 c = *reinterpret_cast<ulonglong2 *>(rec);
                c.x += (uint32_t)v;
                c.y += v >> 32;
                *reinterpret_cast<ulonglong2 *>(rec) = c;
            }
            if (g) rec[2] += g;
        }
    }
}

static uint32_t part_span_bits(size_t n_barcodes) {               // ~1000 bins where the barcodes allow it, 256 .. 8192 barcodes per bin
    uint32_t lg = 0;
    while (((size_t)1 << lg) < n_barcodes) ++lg;
    const uint32_t want = lg > 10 ? lg - 10 : 0;
    return want < 8 ? 8u : (want > kPartMaxSpanBits ? kPartMaxSpanBits : want);
}
size_t commit_partition_scratch_bytes(size_t n_reads, size_t n_barcodes, uint32_t *n_bins_out, uint32_t *cap_out) {
    const uint32_t sb = part_span_bits(n_barcodes);
    const uint32_t n_bins = (uint32_t)((n_barcodes + ((size_t)1 << sb) - 1) >> sb);
    const uint64_t mean = n_bins ? (n_reads + n_bins - 1) / n_bins : 0;
    const uint32_t cap = (uint32_t)std::min<uint64_t>(0x7FFFFFFFull, mean + mean / 2 + 2048);
    if (n_bins_out) *n_bins_out = n_bins;
    if (cap_out) *cap_out = cap;
    // [over_n u64 | pad][bin_fill][bin_valid][bin_recs n_bins x cap][over_ids n][over_votes n]
    return 256 + (size_t)n_bins * 8 + 256 + (size_t)n_bins * cap * 4 + 256 + n_reads * 4 + 256 + n_reads * 8;
}
// worth it (and possible) for large batches over many barcodes: enough bins to fill the GPU, votes that fit a byte
bool commit_partition_usable(size_t n_reads, size_t n_barcodes, uint32_t max_votes, bool forced) {
    uint32_t n_bins, cap;
    (void)commit_partition_scratch_bytes(n_reads, n_barcodes, &n_bins, &cap);
    if (max_votes > 255 || n_bins < 1 || n_bins > kPartMaxBins || n_reads < 1 || n_reads >= (1ull << 31)) return false;
    if ((uint64_t)cap * 255u >= (1ull << 32)) return false;                  // a bin's sums are 32 + 32 bits in LDS
    return forced || (n_bins >= 128 && n_reads >= (1u << 21));
}
hipError_t launch_commit_partitioned(const uint32_t *d_votes, const uint32_t *d_barcode_ids, unsigned long long *d_counts, size_t n_barcodes, size_t n_reads,
                                     void *d_scratch, hipStream_t s) {
    uint32_t n_bins, cap;
    (void)commit_partition_scratch_bytes(n_reads, n_barcodes, &n_bins, &cap);
    const uint32_t sb = part_span_bits(n_barcodes);
    unsigned char *p = reinterpret_cast<unsigned char *>(d_scratch);
    unsigned long long *over_n = reinterpret_cast<unsigned long long *>(p);
    uint32_t *bin_fill = reinterpret_cast<uint32_t *>(p + 256);
    uint32_t *bin_valid = bin_fill + n_bins;
    size_t at = 256 + (size_t)n_bins * 8;
    at = (at + 255) & ~(size_t)255;
    uint32_t *bin_recs = reinterpret_cast<uint32_t *>(p + at);
    at += (size_t)n_bins * cap * 4;
    at = (at + 255) & ~(size_t)255;
    uint32_t *over_ids = reinterpret_cast<uint32_t *>(p + at);
    at += n_reads * 4;
    at = (at + 255) & ~(size_t)255;
    unsigned long long *over_votes = reinterpret_cast<unsigned long long *>(p + at);
    hipError_t e = hipMemsetAsync(p, 0, 256 + (size_t)n_bins * 4, s);                        // over_n, bin_fill
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(bin_valid, 0xFF, (size_t)n_bins * 4, s);
    if (e != hipSuccess) return e;
    const size_t lds1 = (size_t)n_bins * 12 + (size_t)kPartRecs * 4, lds2 = ((size_t)12 << sb);
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_commit_partition), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_commit_bins), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_commit_partition, dim3((unsigned)((n_reads + kPartRecs - 1) / kPartRecs)), dim3(kPartThreads), lds1, s,
                       reinterpret_cast<const unsigned long long *>(d_votes), d_barcode_ids, n_reads, n_bins, sb, cap, bin_fill, bin_valid, bin_recs,
                       over_n, over_ids, over_votes);
    hipLaunchKernelGGL(k_commit_bins, dim3(n_bins), dim3(kBinThreads), lds2, s, bin_recs, bin_fill, bin_valid, cap, sb, d_counts, n_barcodes);
    // what found no room in its bin (a barcode that owns a large share of the reads): the atomic kernel with its LDS cache for such barcodes
    hipLaunchKernelGGL(k_commit_votes, dim3((unsigned)((n_reads + kCommitSpan - 1) / kCommitSpan)), dim3(256), 0, s,
                       reinterpret_cast<const uint32_t *>(over_votes), over_ids, d_counts, (uint32_t *)nullptr, (size_t)0, over_n);
    return hipGetLastError();
}

hipError_t launch_scan_n(const uint8_t *d_bases, const uint64_t *d_offsets, const uint32_t *d_lens, uint64_t fixed_len, size_t n_reads, uint8_t *d_has_n, hipStream_t s) {
    if (n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_n, dim3((unsigned)((n_reads * 64 + 255) / 256)), dim3(256), 0, s, d_bases, d_offsets, d_lens, n_reads, d_has_n, fixed_len);
    return hipGetLastError();
}
hipError_t launch_commit_votes(const uint32_t *d_votes, const uint32_t *d_barcode_ids, unsigned long long *d_counts, uint32_t *d_votes_out,
                               size_t n_reads, hipStream_t s) {
    if (n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(k_commit_votes, dim3((unsigned)((n_reads + kCommitSpan - 1) / kCommitSpan)), dim3(256), 0, s, d_votes, d_barcode_ids,
                       d_counts, d_votes_out, n_reads, (const unsigned long long *)nullptr);
    return hipGetLastError();
}

hipError_t launch_build_segments(const uint64_t *d_offsets, const uint32_t *d_lens, uint64_t fixed_len, size_t n_reads, int k, uint32_t seg_windows, uint64_t *seg_off,
                                 uint32_t *seg_len, uint32_t *seg_read, unsigned long long *d_counter, const uint8_t *d_skip,
                                 uint64_t cap, uint64_t bases_bytes, uint32_t *d_err, hipStream_t s) {
    if (n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(k_build_segments, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, s, d_offsets, d_lens, n_reads, k,
                       seg_windows, seg_off, seg_len, seg_read, d_counter, d_skip, fixed_len, cap, bases_bytes, d_err);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_add_u64(unsigned long long *dst, const unsigned long long *src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
hipError_t launch_add_u64(unsigned long long *d_dst, const unsigned long long *d_src, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_add_u64, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, d_dst, d_src, n);
    return hipGetLastError();
}

// the three live words of the 32-byte counter records as three arrays (c0[n], c1[n], neg[n]) and back: what crosses xGMI in the
// one all-reduce and PCIe on the way to the host is 24 bytes per barcode, not 32 (the fourth word only pads the records the commit
// kernels update to one 32-byte sector)
__global__ void __launch_bounds__(256) k_counts_pack(const unsigned long long *counts, unsigned long long *packed, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(counts + 4 * i);
        packed[i] = a.x;
        packed[n + i] = a.y;
        packed[2 * n + i] = counts[4 * i + 2];
    }
}
__global__ void __launch_bounds__(256) k_counts_unpack(unsigned long long *counts, const unsigned long long *packed, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        ulonglong2 a;
        a.x = packed[i];
        a.y = packed[n + i];
        *reinterpret_cast<ulonglong2 *>(counts + 4 * i) = a;
        counts[4 * i + 2] = packed[2 * n + i];
    }
}
hipError_t launch_counts_pack(const unsigned long long *d_counts, unsigned long long *d_packed, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_counts_pack, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, d_counts, d_packed, n);
    return hipGetLastError();
}
hipError_t launch_counts_unpack(unsigned long long *d_counts, const unsigned long long *d_packed, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_counts_unpack, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, d_counts, d_packed, n);
    return hipGetLastError();
}

hipError_t launch_synth_keys(const SynthParams &p, int hap, uint64_t first, size_t n, uint64_t *d_out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_synth_keys, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, p, hap, first, n, d_out);
    return hipGetLastError();
}
hipError_t launch_synth_reads(const SynthParams &p, uint64_t first, size_t n, uint8_t *d_bases, uint32_t *d_bc, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_synth_reads, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, p, first, n, d_bases, d_bc);
    return hipGetLastError();
}

}  // namespace hast

// hast_filter.hip -- gfx950 (MI355X, CDNA4) device code: the filter in front of the exact k-mer table and the classify
// kernel that probes it (k_classify_f).  See hast_common.h ("fingerprint filter", "exact entries") for the structure and why
// it exists: the probe is bound by the HBM random-request rate, and this front end needs 25 instead of 40 requests per
// 150-bp read while doing less arithmetic per window.  With prints, results are decided by the exact table alone (every
// positive of the filter is looked up there); with exact entries a match in the filter is the filed string itself with its
// tag bits, and the table is asked only where a key may have found no room.  Either way the reference semantics implemented
// are exactly those of hast_kernels.hip:
//   classify.cpp:182-209 (containN + process_reads), kmer/kmer.h:11,153-166,169-194 (coding, canonical k-mers).
#include "hast_common.h"
#include "hast_device.h"
#include "hast_devutil.h"

namespace hast {

// ------------------------------------------------------------------------------------------
// Filter build: every live key of the exact table is filed under the block of its own string and under the block of its
// reverse complement (a read window is looked up under the block of the window as it stands).  A sub-bucket holds 8
// 16-bit prints that fill in order (0 = free); a key that finds its sub-bucket full is simply not filed: lookups treat a
// full sub-bucket as "ask the exact table".
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sub_load(const uint32_t *w) {                 // prints in a sub-bucket (they fill in order)
    uint32_t n = 0;
    for (int i = 0; i < kFilterPrints / 2; ++i) {
        const uint32_t v = __hip_atomic_load(&w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        n += (v & 0xFFFFu) != 0;
        n += (v >> 16) != 0;
    }
    return n;
}
// true: the print is in the sub-bucket now (it was there, or it found a free slot); false: the sub-bucket is full
__device__ __forceinline__ bool sub_insert(uint32_t *w, uint32_t fp) {
    for (int i = 0; i < kFilterPrints / 2;) {
        const uint32_t v = __hip_atomic_load(&w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t lo = v & 0xFFFFu, hi = v >> 16;
        if (lo == fp || hi == fp) return true;                       // an equal print is already there
        uint32_t nv;
        if (lo == 0) nv = v | fp;
        else if (hi == 0) nv = v | (fp << 16);
        else { ++i; continue; }
        if (atomicCAS(&w[i], v, nv) == v) return true;               // else: someone else wrote this word, look again
    }
    return false;
}
__device__ __forceinline__ void filter_insert(uint32_t *filt, const FilterGeom g, uint64_t key, uint32_t tags) {
    for (int o = 0; o < 2; ++o) {
        const uint64_t s = o ? kmer_revcomp(key, g.k) : key;
        if (o && s == key) break;                                    // its own reverse complement
        if (g.exact) {
            // exact entry (hast_common.h): the string's code picks the one sub-bucket it can sit in and is stored with the tags
            const uint32_t pm = filter_sample_pos(s, g);
            const uint32_t blk = filter_block_of((uint32_t)(s >> (2 * (g.k - g.m - (int)pm))) & (uint32_t)kmer_mask(g.m), g.m);
            const uint32_t c17 = filter_exact_code(s, pm, g);
            uint32_t *w = filt + (size_t)blk * (kFilterSubs * kFilterPrints / 2) + filter_exact_sub(c17) * (kFilterPrints / 2);
            (void)sub_insert(w, filter_exact_entry(c17, tags));      // full: not filed; windows that land there ask the table
            continue;
        }
        // block, sub-buckets and print all come from the string AS A READ WOULD SHOW IT: the probe never canonicalises
        const uint32_t blk = filter_block_of_string(s, g);
        const uint32_t h = filter_keyhash(s), fp = filter_print_of(h);
        uint32_t *b = filt + (size_t)blk * (kFilterSubs * kFilterPrints / 2);
        uint32_t *w1 = b + filter_sub_of(h) * (kFilterPrints / 2), *w2 = b + filter_sub2_of(h) * (kFilterPrints / 2);
        if (g.choices == 1) {                                        // lightly loaded blocks: the one sub-bucket of the print
            (void)sub_insert(w1, fp);
            continue;
        }
        // the less loaded of the two first; a print that is already in either one is not filed again
        bool there = false;
        for (int i = 0; i < kFilterPrints / 2 && !there; ++i) {
            const uint32_t v = __hip_atomic_load(&w2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            there = (v & 0xFFFFu) == fp || (v >> 16) == fp;
        }
        if (there) continue;
        if (w1 != w2 && sub_load(w2) < sub_load(w1)) { uint32_t *t = w1; w1 = w2; w2 = t; }
        if (!sub_insert(w1, fp) && w1 != w2) (void)sub_insert(w2, fp);     // both full: not filed; such windows ask the table
    }
}

__global__ void __launch_bounds__(256) k_filter_build(const uint64_t *slots, size_t nslots, uint32_t *filt, FilterGeom g, int wide) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nslots; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t s = slots[i];
        if (wide) {
            if (s < kTombSlot) filter_insert(filt, g, s, 0);         // (wide keys carry no tags and are never filed exactly)
        } else if (s != kEmptySlot && (s & 3)) filter_insert(filt, g, s >> 2, (uint32_t)(s & 3));
    }
}

hipError_t launch_filter_build(const uint64_t *slots, TableGeom tg, void *filter, FilterGeom fg, hipStream_t s) {
    const size_t nslots = (size_t)tg.nbuckets * kSlotsPerBucket * (tg.wide ? 2 : 1);
    size_t grid = (nslots + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(k_filter_build, dim3((unsigned)grid), dim3(256), 0, s, slots, nslots, (uint32_t *)filter, fg, tg.wide);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// k_classify_f.  Workgroup = 256 threads = 4 wave64; tiles of TR reads go through LDS:
//   A  pack   : as hast_kernels.hip: 16 ASCII bases per lane -> 32 bits of 2-bit codes; 'N' flag / invalid-byte mask.
//   M  order  : e[q] = tmer_order(t-mer at q, q) for every position, and the first level of the sliding minimum,
//               L1[q] = min(e[q .. q+3]): a lane takes 4 consecutive positions (one funnel shift, four hashes) and needs
//               the prefix minima of the next lane's four (three DPP moves).  A window's smallest t-mer (leftmost on
//               ties) is then the minimum of ceil((kp-t+1)/4) L1 entries.
//   B  probe  : each wave walks a contiguous quarter of the tile's reads, 64 windows per instruction; every LANE owns one
//               window: its K-mer by funnel shift out of the packed LDS words, smallest t-mer -> position x -> the m-mer at
//               x mod W names the 128-B block (no canonical form anywhere in the probe: every key was filed once per
//               strand); consecutive windows (adjacent lanes) mostly name the same block, which the memory system fetches
//               once.  Instructions are software-pipelined: the loads of instruction i+1 are in flight while i is compared.
//               prints (K > 21, small tables): a hash of the K-mer as it stands names two 16-B sub-buckets and a 16-bit
//                 print; two 16-B loads; compare = 8 xor + 7 v_pk_min_u16 + has-zero-halfword.  Positives (print found, or
//                 both sub-buckets full) -- the real hits plus a few in 10^5 false ones -- go to the wave's queue in LDS.
//               exact entries (EXACT): the window's 17-bit code names ONE sub-bucket and 14 stored bits; one 16-B load;
//                 compare = 4 xor + 3 v_pk_max_u16; a match carries the tag bits and is added to the read's votes at once,
//                 only "no match in a full sub-bucket" is queued.
//   V  verify : when a wave's queue holds 64 entries (and at the end of the tile) each lane takes one, canonicalises it
//               (v_bfrev) and finds it in the exact table (home bucket by the table's own minimizer, then the chain)
//               and adds its tag bits to the read's votes in LDS.
//   C  votes  : one lane per read stores {vote0, vote1}; k_commit_votes does the per-barcode bookkeeping.
// ------------------------------------------------------------------------------------------
constexpr int kThreadsF = 256;
constexpr int kQCap = 128;                                    // queue entries per wave: < 64 waiting + <= 64 new
#ifndef HAST_F_MINWAVES
#define HAST_F_MINWAVES 5      // 96 VGPRs: measured best of 4/5/6/8 (242 vs 226-228 Gbp/s with prints; exact entries: 4, 5, 6 within 1.5 %)
#endif
typedef unsigned long long u64x2f __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4f __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2f __attribute__((ext_vector_type(2)));

size_t classify_f_queue_bytes() { return (size_t)4 * kQCap * 3 * sizeof(uint32_t); }

__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    const u16x2f r = __builtin_elementwise_min(__builtin_bit_cast(u16x2f, a), __builtin_bit_cast(u16x2f, b));
    return __builtin_bit_cast(uint32_t, r);
}
// wave-wide vote as an SGPR pair: the builtin is one s_and of the compare's result with exec (HIP's __ballot goes
// through a v_cndmask + v_cmp pair)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    const u16x2f r = __builtin_elementwise_max(__builtin_bit_cast(u16x2f, a), __builtin_bit_cast(u16x2f, b));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {          // set bits of mask below my lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// NTC = first-level minima per window, ceil((K-t+1)/g), when known at compile time (0 = runtime loop)
// EXACT: the filter holds exact entries (hast_common.h): one 16-B load per window, a match IS a hit with its tag bits
// TWO: prints may sit in either of two sub-buckets (FilterGeom::choices == 2): two loads and compares per window
// GEO: 0 = the filter's geometry is read from the arguments; 1 / 2 = the two BASELINE geometries as compile-time constants
// (1: K = 21, m = 14, t = 6, kp = 21 -- config 1-4 with exact entries; 2: K = 31, m = 15, t = 6, kp = 23 -- config 5).  The kernel
// keeps more wave-uniform values alive than a wave has SGPRs; with the geometry folded into immediates the shifts, masks and the
// x mod W of the probe loop need no registers at all and the spill reloads (v_readlane) leave the loop: 1612 -> 1453 static VALU
// instructions (GEO 1), 1716 -> 1596 (GEO 2); measured on one box, both ways twice (profiles/round3_ab_geo.log): 1-2 % less
// kernel time on C3, config 5 and clustered keys.
template <int GEO> struct GeoConst { static constexpr int k = 0, m = 0, t = 0, kp = 0; };
template <> struct GeoConst<1> { static constexpr int k = 21, m = 14, t = 6, kp = 21; };
template <> struct GeoConst<2> { static constexpr int k = 31, m = 15, t = 6, kp = 23; };

// RL (only with GEO): the rows' length as a compile-time constant too (150- and 100-bp reads; the 512-base segment rows of long reads):
// words per row, the L1 stride and the windows per row become immediates and the row/position divisions constant divisions.
template <int NTC, bool FAST, bool STRICT, bool WIDE, bool EXACT, bool TWO, int GEO = 0, int RL = 0>
__global__ void __launch_bounds__(kThreadsF, HAST_F_MINWAVES) k_classify_f(ClassifyArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    typedef GeoConst<GEO> GC0;
    constexpr bool RLC = RL != 0 && GEO != 0;                        // row shape known at compile time
    const uint32_t TR = a.tile_reads;
    const uint32_t W64 = RLC ? (uint32_t)((RL + 31) / 32) : a.w64;
    const uint32_t WS = W64 + 1;                                     // LDS words per read incl. pad
    const uint32_t L1S = RLC ? (uint32_t)((RL - GC0::t + 1 + 1 + 3) & ~3) : a.l1_stride;
    unsigned long long *s_tile = reinterpret_cast<unsigned long long *>(smem);            // next tile of this workgroup
    uint32_t *s_dirty = reinterpret_cast<uint32_t *>(s_tile + 1);                          // STRICT: some row of the tile holds a byte outside ACGT
    uint32_t *s_l1 = reinterpret_cast<uint32_t *>(s_tile + 2);                             // [TR][L1S] (+ 64 pad), 16-B aligned rows
    unsigned long long *s_pack = reinterpret_cast<unsigned long long *>(s_l1 + (size_t)TR * L1S + 64);   // [TR][WS]
    unsigned long long *s_vote = s_pack + (size_t)TR * WS;                                 // [TR]
    unsigned long long *s_off = s_vote + TR;                                               // [TR]
    uint32_t *s_len = reinterpret_cast<uint32_t *>(s_off + TR);                            // [TR]
    uint32_t *s_flag = s_len + TR;                                                         // [TR]
    uint32_t *s_q = s_flag + TR;                                                           // [4][3][kQCap]
    const uint32_t IW = 2 * W64 + 1;                                                       // invalid-byte mask words per read
    uint32_t *s_inv = s_q + 4 * 3 * kQCap;                                                 // [TR][IW], STRICT only

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));      // wave-uniform: loop control stays scalar
    uint32_t *q_lo = s_q + wave * 3 * kQCap, *q_hi = q_lo + kQCap, *q_rd = q_hi + kQCap;
    typedef GeoConst<GEO> GC;
    FilterGeom fg = a.fg;
    if (GEO) {                       // (the launcher has checked that the arguments say the same)
        fg.k = GC::k; fg.m = GC::m; fg.t = GC::t; fg.kp = GC::kp;
        fg.g = 4;
        fg.wdiv = 65536u / (uint32_t)(GC::kp - GC::m + 1) + 1u;
    }
    const int K = GEO ? GC::k : a.k, M = fg.m, T = fg.t;
    const uint32_t G = (uint32_t)fg.g, W = filter_w(fg), NT = filter_nt(fg);
    const uint32_t ntc = NTC ? (uint32_t)NTC : (NT + G - 1) / G;
    const uint32_t kshift = 64 - 2 * K, tshift = 64 - 2 * T;
    const uint32_t mmask = (uint32_t)kmer_mask(M);
    const uint32_t nb = a.nbuckets;
    const uint64_t n_rows = a.n_rows_ptr ? min((uint64_t)*a.n_rows_ptr, (uint64_t)a.n_reads) : a.n_reads;   // (never past the table the host sized)
    const uint64_t n_tiles = (n_rows + TR - 1) / TR;
    const uintptr_t base_addr = reinterpret_cast<uintptr_t>(a.bases);
    const uintptr_t end_addr = (base_addr + a.bases_bytes + 3) & ~(uintptr_t)3;
    const u32x4f *filt = reinterpret_cast<const u32x4f *>(a.filter);
    const u64x2f *tab0 = reinterpret_cast<const u64x2f *>(a.slots);

    if (tid == 0) *s_tile = atomicAdd(a.tile_queue, 1ull);
    __syncthreads();
    for (;;) {
        const uint64_t tile = *s_tile;
        if (tile >= n_tiles) break;
        const uint64_t r0 = tile * TR;
        const uint32_t tra = (uint32_t)((n_rows - r0 < TR) ? (n_rows - r0) : TR);

        // ---- per-read header --------------------------------------------------------------
        if (STRICT && tid == 0) *s_dirty = 0;           // (phase A, behind the barrier below, raises it; everyone is past the previous tile's probes)
        if (tid < tra) {
            uint64_t off, len;
            const uint32_t rlen = RLC ? (uint32_t)RL : a.read_len;
            if (a.offsets) { off = a.offsets[r0 + tid]; len = a.lens ? a.lens[r0 + tid] : a.offsets[r0 + tid + 1] - off; }
            else           { off = (r0 + tid) * (uint64_t)rlen; len = rlen; }
            if (len > rlen) len = rlen;                      // contract: read_len bounds every read
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
        const uint32_t HW = W64 * 2;                                  // 16-base half-words per read
        for (uint32_t t = tid; t < tra * HW; t += kThreadsF) {
            const uint32_t r = RLC ? (t / HW) : FAST ? __umulhi(t, a.div_hw) : (t / HW);
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
            // 64-bit LDS word w = bases 32w..32w+31, first base most significant: even half-word = HIGH 32 bits
            uint32_t *dst = reinterpret_cast<uint32_t *>(s_pack + (size_t)r * WS + (j >> 1)) + (1 - (j & 1));
            *dst = packed;
            if (STRICT) {
                uint16_t *di = reinterpret_cast<uint16_t *>(s_inv + (size_t)r * IW + (j >> 1)) + (1 - (j & 1));
                *di = (uint16_t)invalid;
                if (invalid) {
                    atomicOr(&s_flag[r], 1u);                                   // the row holds a byte outside ACGT somewhere
                    *s_dirty = 1;                                                // ... and so does the tile
                }
            } else if (nflag) atomicOr(&s_flag[r], 1u);
        }
        __syncthreads();

        // ---- M: t-mer order and the first level of the sliding minimum, L1[q] = min(e[q .. q+g-1]) ---------------------
        if (G == 4) {
            // a lane takes 4 consecutive positions (one funnel shift, four hashes); L1 of its positions needs its own
            // suffix minima and the prefix minima of the next lane's four (three shuffles); lane 63 only serves lane 62
            const uint32_t gpr = L1S >> 2, total = tra * gpr;          // groups of 4 positions per read (L1S % 4 == 0)
            const uint32_t t3shift = 64 - 2 * ((uint32_t)T + 3), tmask = (uint32_t)kmer_mask(T);
            for (uint32_t b0 = wave * 63; b0 < total; b0 += 4 * 63) {
                const uint32_t gi = b0 + lane;
                const bool in = gi < total;
                uint32_t r = RLC ? (gi / gpr) : FAST ? __umulhi(gi, a.div_l1g) : (gi / gpr);
                r = in ? r : 0;
                const uint32_t q0 = in ? 4 * (gi - r * gpr) : 0;
                const int nv = in ? (int)s_len[r] - T + 1 - (int)q0 : 0;                  // e[q0 + i] exists iff i < nv
                const unsigned long long bits = window_bits(s_pack + (size_t)r * WS, nv > 0 ? q0 : 0, t3shift);   // T+3 bases
                uint32_t e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    e[i] = i < nv ? tmer_order((uint32_t)(bits >> (2 * (3 - i))) & tmask, q0 + i) : 0xFFFFFFFFu;
                const uint32_t s2 = min(e[2], e[3]), s1 = min(e[1], s2), s0 = min(e[0], s1);
                const uint32_t p1 = min(e[0], e[1]), p2 = min(p1, e[2]);
                // the next lane's values: wave_shl:1 in the VALU's own data path (a ds_bpermute would queue behind the LDS traffic)
                const uint32_t n0 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)e[0], 0x130, 0xF, 0xF, false);
                const uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)p1, 0x130, 0xF, 0xF, false);
                const uint32_t n2 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)p2, 0x130, 0xF, 0xF, false);
                if (lane < 63 && in)
                    *reinterpret_cast<u32x4f *>(s_l1 + (size_t)r * L1S + q0) = u32x4f{s0, min(s1, n0), min(s2, n1), min(e[3], n2)};
            }
        } else {
            const uint32_t total = tra * L1S, step = 64 - (G - 1);
            for (uint32_t b0 = wave * step; b0 < total; b0 += 4 * step) {
                const uint32_t idx = b0 + lane;
                uint32_t r = idx / L1S;
                uint32_t q = idx - r * L1S;
                const bool in = idx < total;
                r = in ? r : 0;
                const bool valid = in && (q + T <= s_len[r]);
                q = valid ? q : 0;
                const uint32_t tm = (uint32_t)window_bits(s_pack + (size_t)r * WS, q, tshift);
                const uint32_t e = valid ? tmer_order(tm, q) : 0xFFFFFFFFu;
                uint32_t mn = e;
                if (G >= 2) mn = min(mn, (uint32_t)__shfl_down((int)e, 1));
                if (G >= 3) mn = min(mn, (uint32_t)__shfl_down((int)e, 2));
                if (lane < step && in) s_l1[idx] = mn;
            }
        }
        __syncthreads();
        if (!STRICT) {          // reads with 'N' get length 0 (whole-read skip, classify.cpp:190-193)
            if (tid < tra && s_flag[tid]) s_len[tid] = 0;
            __syncthreads();
        }

        // ---- B: probe ---------------------------------------------------------------------------------------------
        const uint32_t P = RLC ? (uint32_t)(RL - GC0::k + 1) : a.max_pos;   // windows per read (stride)
        // STRICT: nearly every tile is clean (long reads hold an 'N' or a lower-case run every few hundred kb): one scalar test per
        // probe instruction instead of a row-flag read + compare + ballot per window
        const bool tile_dirty = STRICT && __builtin_amdgcn_readfirstlane((int)*s_dirty) != 0;
        uint32_t qn = 0;                                              // positives waiting in this wave's queue
        // V: the last `cnt` queue entries, one per lane, against the exact table
        auto drain = [&](uint32_t cnt) {
            const bool act = lane < cnt;
            const uint32_t e = act ? qn - cnt + lane : 0;
            const unsigned long long key = kmer_canon(((unsigned long long)q_hi[e] << 32) | q_lo[e], K);   // queued: the window as read
            const uint32_t rd = q_rd[e];
            qn -= cnt;
            for (int pass = 0; pass < (WIDE ? 2 : 1); ++pass) {
                const u64x2f *tab = tab0 + (WIDE ? (size_t)pass * nb * kPieces : 0);
                const unsigned long long want = WIDE ? key : (key << 2);
                uint32_t b = act ? home_bucket(key, K, a.m, nb) : 0;
                bool pending = act;
                uint32_t step = 0;
                while (__any(pending)) {
                    u64x2f s[kPieces];
#pragma unroll
                    for (int l = 0; l < kPieces; ++l) s[l] = u64x2f{kEmptySlot, kEmptySlot};
                    if (pending) {
#pragma unroll
                        for (int l = 0; l < kPieces; ++l) s[l] = tab[(size_t)b * kPieces + l];
                    }
                    unsigned long long hit_slot = 0;
                    bool hit = false;
#pragma unroll
                    for (int l = 0; l < kPieces; ++l) {
                        if ((WIDE ? s[l].x : (s[l].x & ~3ull)) == want) { hit_slot = s[l].x; hit = true; }
                        if ((WIDE ? s[l].y : (s[l].y & ~3ull)) == want) { hit_slot = s[l].y; hit = true; }
                    }
                    hit = hit && pending;
                    if (hit) {
                        const uint32_t tags = WIDE ? (1u << pass) : (uint32_t)(hit_slot & 3);
                        if (tags) atomicAdd(&s_vote[rd], (unsigned long long)(tags & 1) | ((unsigned long long)(tags >> 1) << 32));
                    }
                    // slots fill in order: the bucket is full iff its last slot is taken; only then the chain goes on
                    if (pending && (hit || s[kPieces - 1].y == kEmptySlot || ++step >= nb)) pending = false;
                    if (pending) b = next_bucket(b, step, key, nb);
                }
            }
        };
        struct Blk {
            u32x4f v, v2;                               // the 2 x 8 prints of this lane's window's two sub-buckets
            uint32_t klo, khi, fpw, meta;               // the window's K-mer, print in both halves, read | valid << 31
        };
        // One probe instruction = the next 64 windows of this wave's range, one per lane, starting at `pos`.  Lanes without a
        // window to probe (read shorter than the stride, read with 'N', range end) ask for the block of a lane that has one:
        // same 128-B block, no request of their own -- they used to fetch a random block each (26.9 -> 26.0 requests per read
        // on the bench's reads, where one read in 200 holds an 'N'; ragged reads gain more).
        // Tried and dropped: cutting the instruction at the first lane of its last run, so that the run which straddles two
        // instructions asks only once.  TCC_EA0_RDREQ stayed at 26.006 per read: the second request for a block already on
        // its way never reaches the fabric, and the lanes a cut idles cost 4 % of the time (the kernel is bound by VALU issue
        // as much as by requests).
        uint32_t pos = 0, q_end = 0, last_fb = 0;                     // wave-uniform
        uint32_t cr = 0, cp = 0;                                      // this lane's window in the next instruction: read, position
        // (re)derive cr, cp from pos: once per range, and per instruction when reads have fewer than 64 windows
        auto locate = [&]() {
            const uint32_t q = pos + lane;
            cr = RLC ? (q / P) : FAST ? __umulhi(q, a.div_magic) : (q / P);
            cp = q - cr * P;
        };
        auto start = [&](Blk &B) {
            const bool inq = lane < q_end - pos;                      // pos <= q_end
            const uint32_t r = min(cr, TR - 1), p = cp;
            const uint32_t len = s_len[r];
            const bool fits = p + K <= len;
            bool ok = inq & fits;
            // the wave-wide mask of `ok`: a ballot per compare (each is the compare itself, written to an SGPR pair) and one
            // scalar AND -- a ballot of the combined condition goes through v_cndmask + v_cmp
            unsigned long long okm = ballot64(inq) & ballot64(fits);
            if (STRICT) {           // any byte of [p, p+K) outside ACGT => the window cannot match a stored string
                // (rows without such a byte -- nearly all of them -- are flagged clean by phase A: the wave skips the masks)
                if (tile_dirty && ballot64(s_flag[r] != 0)) {
                    const uint32_t *iw = s_inv + mul24(r, IW) + (p >> 5);
                    const unsigned long long bits = (((unsigned long long)iw[0] << 32) | iw[1]) << (p & 31);
                    const bool clean = (bits >> (64 - K)) == 0;
                    ok = ok & clean;
                    okm &= ballot64(clean);
                }
            }
            const unsigned long long fwd = window_bits(s_pack + mul24(r, WS), p, kshift);
            const uint32_t *l1 = s_l1 + mul24(r, L1S) + p;
            uint32_t x = l1[0];
            if (NTC) {
#pragma unroll
                for (int c = 1; c + 1 < (NTC ? NTC : 1); ++c) x = min(x, l1[c * 4]);
            } else {
                for (uint32_t c = 1; c + 1 < ntc; ++c) x = min(x, l1[c * G]);
            }
            if (ntc > 1) x = min(x, l1[NT - G]);
            // lanes without a window compute on whatever LDS holds: in range by construction, and never used
            const uint32_t xr = ((x & 0xFFFu) - p) & 63u;             // position of the smallest t-mer inside the window
            // ... mod W = position of the sampled m-mer (a mask when W is a known power of two)
            const uint32_t pm = (GEO && (W & (W - 1)) == 0) ? (xr & (W - 1)) : xr - mul24(mul24(xr, fg.wdiv) >> 16, W);
            const uint32_t mm = (uint32_t)(fwd >> (2 * ((uint32_t)(K - M) - pm))) & mmask;
            uint32_t fb = ok ? filter_block_of(mm, M) : 0xFFFFFFFFu;  // real blocks are < 4^15
            // of the window as it stands (no canonical form in the probe): a hash for two sub-buckets and a print, or the
            // window's exact code, whose top bits are its one sub-bucket
            const uint32_t h = EXACT ? filter_exact_code(fwd, pm, fg) : filter_keyhash(fwd);
            const uint32_t anchor = okm ? (uint32_t)__builtin_amdgcn_readlane((int)fb, (int)__builtin_ctzll(okm)) : last_fb;
            last_fb = anchor;
            fb = ok ? fb : anchor;
            if (EXACT) {
                // exact entries mean m <= 14 (filter_exact_fits: 2(K-m) <= 17): block * 8 + sub-bucket fits 32 bits, one shift-or
                // and one 64-bit shift-add make the address
                B.v = filt[(fb << 3) | filter_exact_sub(h)];                  // (any sub-bucket will do for a lane without a window)
                B.fpw = mul24((h & 0x3FFFu) ^ 0x3FFFu, 0x00040004u);          // the complement of the stored bits, in both halves
            } else {
                B.v = filt[(size_t)fb * kFilterSubs + filter_sub_of(h)];
                if (TWO) B.v2 = filt[(size_t)fb * kFilterSubs + filter_sub2_of(h)];
                const uint32_t fp = filter_print_of(h);
                B.fpw = fp | (fp << 16);
            }
            B.klo = (uint32_t)fwd;
            B.khi = (uint32_t)(fwd >> 32);
            B.meta = r | (ok ? 0x80000000u : 0u);
            pos = pos + 64 < q_end ? pos + 64 : q_end;
            if (P >= 64) {                                            // the lane's next window: 64 <= P, one wrap at most
                cp += 64;
                const bool wrap = cp >= P;
                cp -= wrap ? P : 0u;
                cr += wrap ? 1u : 0u;
            } else locate();
        };
        auto finish = [&](Blk &B) {
            bool pos;
            if (EXACT) {
                // entry ^ complement(stored bits): the 14 code bits of a matching entry come out all ones, its tags (1..3) below
                // them -- the largest halfword of the sub-bucket is >= 0xFFFD iff the window's string is filed here (an empty
                // slot gives at most 0xFFFC)
                const uint32_t mx = pk_max_u16(pk_max_u16(B.v.x ^ B.fpw, B.v.y ^ B.fpw), pk_max_u16(B.v.z ^ B.fpw, B.v.w ^ B.fpw));
                const uint32_t m16 = max(mx >> 16, mx & 0xFFFFu);
                const bool valid = (int)B.meta < 0;
                const bool hit = valid && m16 >= 0xFFFDu;
                if (hit) atomicAdd(&s_vote[B.meta & 0xFFFFu], (unsigned long long)(m16 & 1u) | ((unsigned long long)((m16 >> 1) & 1u) << 32));
                pos = valid && !hit && (B.v.w >> 16) != 0;               // no match in a FULL sub-bucket: the key may not have found room
            } else {
            uint32_t acc = pk_min_u16(pk_min_u16(B.v.x ^ B.fpw, B.v.y ^ B.fpw), pk_min_u16(B.v.z ^ B.fpw, B.v.w ^ B.fpw));
            bool full = (B.v.w >> 16) != 0;
            if (TWO) {
                acc = pk_min_u16(acc, pk_min_u16(pk_min_u16(B.v2.x ^ B.fpw, B.v2.y ^ B.fpw), pk_min_u16(B.v2.z ^ B.fpw, B.v2.w ^ B.fpw)));
                full = full && (B.v2.w >> 16) != 0;                                  // a key finds no room only when BOTH are full
            }
            const bool match = ((acc - 0x00010001u) & ~acc & 0x80008000u) != 0;      // some halfword of acc is zero
            pos = (int)B.meta < 0 && (match || full);
            }
            const unsigned long long pmask = ballot64(pos);
            if (pmask) {
                const uint32_t at = qn + lanes_below(pmask);
                if (pos) {
                    q_lo[at] = B.klo;
                    q_hi[at] = B.khi;
                    q_rd[at] = B.meta & 0xFFFFu;
                }
                qn += (uint32_t)__popcll(pmask);
                if (qn >= 64) drain(64);
            }
        };
        {
            // Software pipeline, two instructions deep.  Every start() sits in straight-line code (no branch around it), so
            // the compiler can count the loads in flight and wait for instruction i with vmcnt(2) while i+1's loads are
            // still out; a start() under an `if` makes it fall back to vmcnt(0), i.e. no overlap at all.  Hence the loop
            // may issue one start() past the end of the range: all its lanes re-ask for the last block (an L1/L2 hit).
            // A wave takes a contiguous quarter of the tile's READS: range borders are read borders, where runs end anyway.
            const uint32_t rpw = (tra + 3) >> 2;
            const uint32_t r_first = wave * rpw < tra ? wave * rpw : tra, r_last = r_first + rpw < tra ? r_first + rpw : tra;
            pos = r_first * P;
            q_end = r_last * P;
            locate();
            if (pos < q_end) {
                Blk A, B;
                start(A);
                while (pos < q_end) {
                    start(B);
                    finish(A);
                    start(A);
                    finish(B);
                }
                finish(A);
            }
            while (qn) drain(qn < 64 ? qn : 64);
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
// Measurement entry (hast_filter_request_ceiling): uniformly random 128-B blocks of the filter itself, read in the probe's
// access shape (a group of 8 lanes takes one block, 16 B per lane, four blocks in flight per lane) with next to no
// arithmetic -- what the memory system serves over THIS allocation on THIS box, the ceiling k_classify_f's request rate is
// priced against in the same run.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_request_ceiling(const u32x4f *filt, uint64_t nblocks, uint32_t iters, uint32_t *sink) {
    const uint64_t gid = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 3;
    const uint32_t sub = threadIdx.x & 7;
    u32x4f acc = {0, 0, 0, 0};
    for (uint32_t it = 0; it < iters; ++it) {
        u32x4f v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t h = splitmix64(gid * 0x10001ull + (uint64_t)it * 4 + (uint64_t)u);
            const uint64_t blk = (uint64_t)(((unsigned __int128)h * nblocks) >> 64);
            v[u] = filt[blk * kFilterSubs + sub];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) sink[0] = 1;       // keeps the loads alive
}
hipError_t launch_request_ceiling(const void *filter, uint64_t nblocks, uint32_t iters, int grid, uint32_t *d_sink, hipStream_t s) {
    hipLaunchKernelGGL(k_request_ceiling, dim3(grid), dim3(256), 0, s, reinterpret_cast<const u32x4f *>(filter), nblocks, iters, d_sink);
    return hipGetLastError();
}

template <int NTC, bool FAST, bool STRICT, bool WIDE, bool EXACT, bool TWO, int GEO = 0, int RL = 0>
static hipError_t launch_f_t(const ClassifyArgs &a, int grid, size_t smem, hipStream_t s) {
    if (smem > (48u << 10)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_classify_f<NTC, FAST, STRICT, WIDE, EXACT, TWO, GEO, RL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_classify_f<NTC, FAST, STRICT, WIDE, EXACT, TWO, GEO, RL>), dim3(grid), dim3(kThreadsF), smem, s, a);
    return hipGetLastError();
}
// the row shape the host computed is the one a kernel with RL compiled in assumes
static bool rows_are(const ClassifyArgs &a, uint32_t rl, int k, int t) {
    return a.read_len == rl && a.w64 == (rl + 31) / 32 && a.max_pos == rl - (uint32_t)k + 1 && a.l1_stride == ((rl - (uint32_t)t + 1 + 1 + 3) & ~3u);
}
static bool geo_is(const ClassifyArgs &a, int k, int m, int t, int kp) {
    return a.k == k && a.fg.k == k && a.fg.m == m && a.fg.t == t && a.fg.kp == kp && a.fg.g == 4;
}

// EXACT implies one sub-bucket per window (TWO = false)
// variants: bit 0 = instantiations with the geometry compiled in allowed, bit 1 = with the row length too (the context's switches,
// read from the environment once when it was created)
template <bool STRICT, bool EXACT, bool TWO>
static hipError_t launch_f_s(const ClassifyArgs &a, int grid, size_t smem, int variants, hipStream_t s) {
    const bool fast = a.div_magic && a.div_l1g && a.div_hw;
    if (a.wide) return fast ? launch_f_t<0, true, STRICT, true, false, TWO>(a, grid, smem, s) : launch_f_t<0, false, STRICT, true, false, TWO>(a, grid, smem, s);
    if (!fast || a.fg.g != 4) return fast ? launch_f_t<0, true, STRICT, false, EXACT, TWO>(a, grid, smem, s) : launch_f_t<0, false, STRICT, false, EXACT, TWO>(a, grid, smem, s);
    // the BASELINE geometries with their constants folded in (kernel_geo = 0: the generic instantiations)
    const bool geo_on = (variants & 1) != 0, rl_on = (variants & 2) != 0;
    if (geo_on && !STRICT && EXACT && !TWO && geo_is(a, 21, 14, 6, 21)) {
        if (rl_on && rows_are(a, 150, 21, 6)) return launch_f_t<4, true, false, false, true, false, 1, 150>(a, grid, smem, s);
        if (rl_on && rows_are(a, 100, 21, 6)) return launch_f_t<4, true, false, false, true, false, 1, 100>(a, grid, smem, s);   // stLFR PE100
        return launch_f_t<4, true, false, false, true, false, 1>(a, grid, smem, s);
    }
    if (geo_on && STRICT && !EXACT && !TWO && geo_is(a, 31, 15, 6, 23)) {
        if (rl_on && rows_are(a, 512, 31, 6)) return launch_f_t<5, true, true, false, false, false, 2, 512>(a, grid, smem, s);
        return launch_f_t<5, true, true, false, false, false, 2>(a, grid, smem, s);
    }
    switch ((filter_nt(a.fg) + 3) / 4) {
    case 1: return launch_f_t<1, true, STRICT, false, EXACT, TWO>(a, grid, smem, s);
    case 2: return launch_f_t<2, true, STRICT, false, EXACT, TWO>(a, grid, smem, s);
    case 3: return launch_f_t<3, true, STRICT, false, EXACT, TWO>(a, grid, smem, s);
    case 4: return launch_f_t<4, true, STRICT, false, EXACT, TWO>(a, grid, smem, s);
    case 5: return launch_f_t<5, true, STRICT, false, EXACT, TWO>(a, grid, smem, s);
    case 6: return launch_f_t<6, true, STRICT, false, EXACT, TWO>(a, grid, smem, s);
    default: return launch_f_t<0, true, STRICT, false, EXACT, TWO>(a, grid, smem, s);
    }
}

hipError_t launch_classify_f(const ClassifyArgs &a, int grid, size_t smem, int variants, hipStream_t s) {
    if (a.n_reads == 0) return hipSuccess;
    if (a.fg.exact && !a.wide) return a.strict ? launch_f_s<true, true, false>(a, grid, smem, variants, s) : launch_f_s<false, true, false>(a, grid, smem, variants, s);
    if (a.fg.choices == 1) return a.strict ? launch_f_s<true, false, false>(a, grid, smem, variants, s) : launch_f_s<false, false, false>(a, grid, smem, variants, s);
    return a.strict ? launch_f_s<true, false, true>(a, grid, smem, variants, s) : launch_f_s<false, false, true>(a, grid, smem, variants, s);
}

}  // namespace hast

// kc_kernels.hip -- gfx950 (MI355X, CDNA4) device code for HAST stage 00: counting the canonical k-mers of both
// parents' reads in ONE open-addressed table in HBM, then reading the parent-unique sets and the count histograms
// straight out of it (reference: 00.build_unshare_kmers_by_jellyfish/build_unshared_kmers.sh:165-291,
// where a third-party CPU hash counter does this in seven count/dump passes over text files).
//
// Integer + HBM work only (no MFMA by design):
//   k_kc_count   : byte stream -> 2-bit codes + validity mask in LDS -> canonical k-mer per window -> find-or-insert in
//                  the window's minimizer bucket (128-B line: 8 keys + 8 paternal + 8 maternal counters) -> one atomic add
//                  (the DIRECT path: small tables, K > 27); with EMIT the same front end writes 8-byte records of minimizer runs
//   k_kc_emit4   : the emit front end of K = 17 .. 21 (m = 16): four positions, then four windows per lane (round 5)
//   k_kc_part<1|2>, k_kc_apply, k_kc_spill : the PARTITIONED path of large tables (below): records -> two levels of bucket ranges
//                  -> one slice of the table at a time in LDS, counted there
//   k_kc_stats / k_kc_histo / k_kc_select : streaming passes over the table
//   k_kc_format  : selected keys -> text lines
#include <cstdlib>
#include <cstring>

#include "hast_common.h"
#include "hast_devutil.h"
#include "kc_common.h"
#include "kc_device.h"

#include <rocprim/device/device_radix_sort.hpp>

namespace hast {

constexpr int kKcThreads = 256;

// find-or-insert `key` starting at bucket b; returns the address of its counter for `parent`, or nullptr when the table is
// full.  Slots never change once written and fill in order, so a (possibly stale) plain read can only show a PREFIX of
// the real bucket: a key seen is there for good, and "not seen" is settled by the compare-and-swap on the first slot that
// looked empty.  The ADD itself is the caller's, after the lanes of the wave have come back together: the lanes of a
// wave are consecutive windows, mostly of the same bucket, and their adds to one 32-B counter sector leave the CU as a
// single request only when they come from the same instruction (tools/atomic_merge_probe.hip; an instrumented build
// counted 48 distinct counter sectors per 150-bp read with the add inside the probe loop, against ~37 minimizer runs).
// Probe sequence of a key (kc_slot, and k_kc_apply inside its slice): the minimizer's bucket, then -- only when that one is full --
// the next three (the same DRAM page; a minimizer's ~3.5 genomic k-mers plus the error k-mers that keep it are ~8.5 keys on average
// at 30x, so half the occupied buckets overflow, while 7 buckets in 8 are empty), then a bucket chosen by the KEY's own hash and
// on from there (minimizers that thousands of keys share -- poly-A -- must not pile up in one run of full buckets).
// slice_n != 0: partitioned placement (kc_common.h): `home` lies in the slice [slice_lo, slice_lo + slice_n), and so do the next three
__device__ __forceinline__ uint32_t kc_probe(uint32_t home, uint32_t p, uint64_t key, uint32_t nb, uint32_t slice_lo, uint32_t slice_n) {
    if (p < 4) {
        if (slice_n) {
            const uint32_t x = home - slice_lo + p;
            return slice_lo + (x >= slice_n ? x - slice_n : x);
        }
        const uint32_t b = home + p;
        return b >= nb ? b - nb : b;
    }
    const uint64_t b = (uint64_t)overflow_bucket(key, nb) + (p - 4);
    return (uint32_t)(b % nb);
}
__device__ __forceinline__ uint32_t *kc_slot(unsigned long long *table, uint32_t nb, uint32_t home, uint64_t key, uint32_t parent, uint32_t slice_lo = 0,
                                              uint32_t slice_n = 0) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    uint32_t b = home;
    for (uint32_t probe = 0; probe < nb; ++probe) {
        unsigned long long *bk = table + (size_t)b * kKcBucketWords;
        uint32_t *cnt = reinterpret_cast<uint32_t *>(bk + kKcSlots) + parent * kKcSlots;   // this parent's 8 counters: one 32-B sector
        const u64x2 *v = reinterpret_cast<const u64x2 *>(bk);
        unsigned long long s[kKcSlots];
#pragma unroll
        for (int i = 0; i < kKcSlots / 2; ++i) {
            const u64x2 t = v[i];
            s[2 * i] = t.x;
            s[2 * i + 1] = t.y;
        }
        int idx = -1, first_empty = kKcSlots;
#pragma unroll
        for (int i = kKcSlots - 1; i >= 0; --i) {
            if (s[i] == key) idx = i;
            if (s[i] == kEmptySlot) first_empty = i;
        }
        for (int i = first_empty; idx < 0 && i < kKcSlots; ++i) {
            const unsigned long long old = atomicCAS(&bk[i], (unsigned long long)kEmptySlot, (unsigned long long)key);
            if (old == kEmptySlot || old == key) idx = i;
        }
        if (idx >= 0) return cnt + idx;
        b = kc_probe(home, probe + 1, key, nb, slice_lo, slice_n);              // bucket full: the next three, then the key's own overflow bucket, then on from there
    }
    return nullptr;
}

// One workgroup walks tiles of `tile_bases` window starts (+ K-1 bytes of overlap) of the byte stream.
// ---- records of the partitioned path ("super-k-mers": what KMC-style counters partition by): layout in kc_common.h --------------
// every window of a record through the atomic path (records that found no room in a buffer, spilled windows)
__device__ __forceinline__ void kc_count_record(unsigned long long *table, uint32_t nb, int k, int m, uint32_t fine_shift, unsigned long long rec, uint32_t *err) {
    const uint32_t parent = (uint32_t)(rec & 1), run = (uint32_t)((rec >> 1) & 31) + 1;
    const unsigned long long bases = rec >> 6, kmask = kmer_mask(k);           // (the offset bits lie above every window's bits)
    const uint32_t lo = kc_fine_of_hash(kc_rec_minhash(rec, k, m, kc_rec_off_bits(k, m)), (nb + (1u << fine_shift) - 1) >> fine_shift) << fine_shift;
    const uint32_t n_here = nb - lo < (1u << fine_shift) ? nb - lo : (1u << fine_shift);
    for (uint32_t j = 0; j < run; ++j) {
        const unsigned long long key = kmer_canon((bases >> (2 * (run - 1 - j))) & kmask, k);
        uint32_t *counter = kc_slot(table, nb, lo + kc_key_bucket(key, n_here), key, parent, lo, n_here);
        if (counter) atomicAdd(counter, 1u);
        else atomicOr(err, 1u);
    }
}

// a record of the emit kernels that found no room in the record buffer
__device__ __forceinline__ void kc_emit_overflow(const KcCountArgs &a, unsigned long long rec) {
    if (!a.fresh) {
        kc_count_record(a.table, a.nbuckets, a.k, a.m, a.fine_shift, rec, a.err);          // counted on the spot
        return;
    }
    const unsigned long long at = atomicAdd(a.spill_n, 1ull);                              // (nothing to count into yet: KcCountArgs)
    if (at < a.spill_cap) a.spill[at] = rec;
    else atomicOr(a.err, 1u);
}

// phase A of a tile (both front ends): 16 bytes per lane -> 32 bits of 2-bit codes + 16 validity bits, into LDS.  Bytes in front of
// the buffer and behind its end read as separators.
__device__ __forceinline__ void kc_tile_pack(uintptr_t base_addr, uintptr_t end_addr, uint64_t t0, uint32_t NW, unsigned long long *s_pack, uint32_t *s_inv, uint32_t tid) {
    const uint32_t HW = NW * 2;
    for (uint32_t j = tid; j < HW; j += kKcThreads) {
        const uintptr_t addr = base_addr + t0 + 16 * (uint64_t)j;
        const uintptr_t a4 = addr & ~(uintptr_t)3;
        const uint32_t bsh = (uint32_t)(addr & 3);
        uint32_t d[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const uintptr_t p = a4 + 4 * i;
            uint32_t x = 0x0A0A0A0Au;
            if (p + 4 <= end_addr && p >= base_addr) x = *reinterpret_cast<const uint32_t *>(p);
            else if (p < end_addr && p + 4 > base_addr) {        // dword straddles an end of the buffer: byte-wise
                x = 0;
                for (int q = 0; q < 4; ++q) {
                    const uintptr_t pb = p + q;
                    const uint32_t c = (pb >= base_addr && pb < end_addr) ? *reinterpret_cast<const uint8_t *>(pb) : 0x0Au;
                    x |= c << (8 * q);
                }
            }
            d[i] = x;
        }
        uint32_t packed = 0, invalid = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t x = __builtin_amdgcn_alignbyte(d[i + 1], d[i], bsh);
            invalid = (invalid << 4) | not_acgt4_anycase(x);
            packed = (packed << 8) | pack4(x);
        }
        // word w = bases 32w..32w+31, first base most significant: the even half-word is the high half
        reinterpret_cast<uint32_t *>(s_pack + (j >> 1))[1 - (j & 1)] = packed;
        reinterpret_cast<uint16_t *>(s_inv + (j >> 1))[1 - (j & 1)] = (uint16_t)invalid;
    }
}

// EMIT: instead of updating the table, write the windows out as records (a.rec_out; what finds no room there is counted on the spot)
template <int WT, bool EMIT>
__global__ void __launch_bounds__(kKcThreads) k_kc_count(KcCountArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t TB = a.tile_bases;
    const int K = a.k, M = a.m;
    const uint32_t W = WT ? (uint32_t)WT : (uint32_t)(K - M + 1);
    const uint32_t span = TB + (uint32_t)K - 1;                     // bytes a tile looks at
    const uint32_t NW = (span + 31) / 32 + 2;                        // 32-base words incl. pad for window_bits
    unsigned long long *s_tile = reinterpret_cast<unsigned long long *>(smem);
    unsigned long long *s_pack = s_tile + 2;                         // [NW]
    uint32_t *s_inv = reinterpret_cast<uint32_t *>(s_pack + NW);     // [NW] invalid-byte masks (bit 31 = first base)
    uint32_t *s_mh = s_inv + NW;                                     // [TB + W] (+ 64: kc_count_smem)
    // EMIT: the tile's records are gathered here and go out with ONE reservation per tile -- a reservation per wave step is 160 M
    // atomic adds on one word per 10 G windows, and adds to ONE address serialise at the memory side (measured: 10 x the kernel's time)
    unsigned long long *s_rec = reinterpret_cast<unsigned long long *>(smem + ((16 + (size_t)NW * 8 + (size_t)NW * 4 + (size_t)(TB + W + 64) * 4 + 15) & ~(size_t)15));   // [RC]
    const uint32_t RC = TB / 2;
    __shared__ uint32_t s_nrec;
    // EMIT: a workgroup reserves room for its records a.rec_chunk at a time and fills the end of a chunk it cannot use (and of its
    // last chunk) with NULL records (all ones: no record has run - 1 == 31), which the partition pass skips.  One reservation per
    // tile was 2.5 M adds on ONE word per 10 G windows -- adds to one address serialise at the memory side -- and the tile queue
    // was another 2.5 M on a second word: with tiles of half the size the kernel took 0.064 s longer (round 4).
    __shared__ unsigned long long s_rec_base, s_chunk_at, s_chunk_end, s_fill_lo, s_fill_hi;
    const uint32_t tid = threadIdx.x;
    const uint32_t kshift = 64 - 2 * K, mshift = 64 - 2 * M;
    const uint64_t n_tiles = (a.n_starts + TB - 1) / TB;
    const uintptr_t base_addr = reinterpret_cast<uintptr_t>(a.bytes);
    const uintptr_t end_addr = base_addr + a.n_bytes;                // bytes at and beyond it read as separators
    unsigned long long counted = 0;

    if (tid == 0) {
        *s_tile = EMIT ? (unsigned long long)blockIdx.x : atomicAdd(a.tile_queue, 1ull);     // EMIT: tiles cost the same, dealt in turn
        s_chunk_at = s_chunk_end = 0;
    }
    __syncthreads();
    for (;;) {
        const uint64_t tile = *s_tile;
        if (tile >= n_tiles) break;
        const uint64_t t0 = tile * TB;
        __syncthreads();
        if (tid == 0) {
            *s_tile = EMIT ? tile + gridDim.x : atomicAdd(a.tile_queue, 1ull);
            s_nrec = 0;
        }

        // ---- A: 16 bytes per lane -> 32 bits of codes + 16 validity bits -----------------------------------
        kc_tile_pack(base_addr, end_addr, t0, NW, s_pack, s_inv, tid);
        __syncthreads();

        // ---- M: hash of the canonical m-mer at every position --------------------------------------------
        if (M <= 16) {                                                // an m-mer of <= 16 bases is one dword: half the instructions
            for (uint32_t q = tid; q < TB + W - 1; q += kKcThreads) {
                const uint32_t f = (uint32_t)window_bits(s_pack, q, mshift);
                uint32_t r = __brev(f ^ 0xAAAAAAAAu);
                r = (((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u)) >> (32 - 2 * M);
                s_mh[q] = kc_mmer_hash32(f < r ? f : r);
            }
        } else {
            for (uint32_t q = tid; q < TB + W - 1; q += kKcThreads)
                s_mh[q] = kc_mmer_hash(kmer_canon(window_bits(s_pack, q, mshift), M));
        }
        __syncthreads();

        // ---- B: one lane per window --------------------------------------------------------------------
        // Consecutive windows with the SAME key (homopolymers, short tandem repeats of period 1) are counted by the first
        // of them with the run length: real reads hold poly-A k-mers by the million, and every one of them would
        // otherwise be an atomic on the same counter.
        const uint64_t left = a.n_starts - t0;
        const uint32_t nwin = left < TB ? (uint32_t)left : TB;
        const uint32_t lane = tid & 63;
        for (uint32_t base = tid - lane; base < nwin; base += kKcThreads) {       // wave-uniform trip count
            const uint32_t p = base + lane;
            const uint32_t *iw = s_inv + (p >> 5);
            const unsigned long long bits = (((unsigned long long)iw[0] << 32) | iw[1]) << (p & 31);
            bool valid = p < nwin && (bits >> (64 - K)) == 0;       // every byte of the window is a base
            const uint64_t key = kmer_canon(window_bits(s_pack, p, kshift), K);
            uint32_t mn = s_mh[p], off = 0;                              // the window's minimizer: smallest hash, leftmost on ties
            if (WT) {
#pragma unroll
                for (int j = 1; j < (WT ? WT : 1); ++j) {
                    const uint32_t h = s_mh[p + j];
                    if (EMIT) off = h < mn ? (uint32_t)j : off;
                    mn = min(mn, h);
                }
            } else {
                for (uint32_t j = 1; j < W; ++j) {
                    const uint32_t h = s_mh[p + j];
                    if (EMIT) off = h < mn ? j : off;
                    mn = min(mn, h);
                }
            }
            if (a.n_slices > 1 && kc_slice_of(mn, a.n_slices) != a.slice) valid = false;
            if (EMIT) {
                // runs of consecutive valid windows whose minimizer is the same m-mer OCCURRENCE (position p + off), cut every kc_run_max
                // windows (and by the 64 lanes of a step): every window of a run then holds the m-mer its first window names
                const uint32_t at_pos = p + off, prev_pos = (uint32_t)__shfl_up((int)at_pos, 1, 64);
                const bool prev_ok = __shfl_up((int)valid, 1, 64) != 0;
                const bool start0 = valid && (lane == 0 || !prev_ok || prev_pos != at_pos);
                const unsigned long long s0 = __ballot(start0), vm = __ballot(valid);
                if (valid) ++counted;
                const uint32_t rmax = a.rec_run_max;
                bool start = start0;
                unsigned long long sm = s0;
                if (rmax < W) {                                           // (wave-uniform; a run of one occurrence is at most W windows)
                    const unsigned long long below = s0 & (~0ull >> (63 - lane));
                    const uint32_t first = below ? 63u - (uint32_t)__builtin_clzll(below) : lane;
                    start = valid && ((lane - first) % rmax) == 0;
                    sm = __ballot(start);
                }
                const unsigned long long after = lane == 63 ? 0ull : ((sm | ~vm) >> (lane + 1));
                const uint32_t run = after ? 1u + (uint32_t)__builtin_ctzll(after) : 64u - lane;
                if (sm) {
                    const uint32_t cnt = (uint32_t)__popcll(sm);
                    uint32_t at0 = 0;
                    if (lane == 0) at0 = atomicAdd(&s_nrec, cnt);
                    at0 = (uint32_t)__shfl((int)at0, 0, 64);
                    unsigned long long rec = 0;
                    if (start) rec = (a.rec_off_bits ? (unsigned long long)off << (64 - a.rec_off_bits) : 0ull) | (window_bits(s_pack, p, 64 - 2 * ((uint32_t)K + run - 1)) << 6) |
                                     ((unsigned long long)(run - 1) << 1) | a.parent;
                    const uint32_t idx = at0 + (uint32_t)__popcll(sm & ((1ull << lane) - 1));
                    if (start && idx < RC) s_rec[idx] = rec;
                    // the staging area holds one record per two windows (a minimizer run holds ~3.3): the rare rest goes out at once
                    const unsigned long long om = __ballot(start && idx >= RC);
                    if (om) {
                        const uint32_t first = (uint32_t)__builtin_ctzll(om);
                        unsigned long long g0 = 0;
                        if (lane == first) g0 = atomicAdd(a.rec_cursor, (unsigned long long)__popcll(om));
                        g0 = __shfl(g0, (int)first, 64);
                        if (start && idx >= RC) {
                            const unsigned long long at = g0 + (unsigned long long)__popcll(om & ((1ull << lane) - 1));
                            if (at < a.rec_cap) a.rec_out[at] = rec;
                            else kc_emit_overflow(a, rec);
                        }
                    }
                }
                continue;
            }
            const uint64_t prev_key = __shfl_up(key, 1, 64);
            const bool prev_valid = __shfl_up((int)valid, 1, 64) != 0;
            const bool follower = valid && lane > 0 && prev_valid && prev_key == key;
            const unsigned long long fmask = __ballot(follower);
            if (valid) ++counted;
            const bool active = valid && !follower;
            // run length = 1 + the followers right after this lane
            const unsigned long long after = lane == 63 ? 0ull : (fmask >> (lane + 1));
            const uint32_t run = 1 + (uint32_t)__builtin_ctzll(~after);
            uint32_t *counter = nullptr;
            if (active) counter = kc_slot(a.table, a.nbuckets, bucket_of_minhash(mn, a.nbuckets), key, a.parent);
            // everyone is back together here: ONE add instruction for the whole wave
            if (active) {
                if (counter) atomicAdd(counter, run);
                else atomicOr(a.err, 1u);
            }
        }
        __syncthreads();
        if (EMIT) {
            const uint32_t n = s_nrec < RC ? s_nrec : RC;
            if (tid == 0) {
                s_fill_lo = s_fill_hi = 0;
                if (s_chunk_at + n > s_chunk_end) {                                        // (n <= RC <= a.rec_chunk)
                    s_fill_lo = s_chunk_at;
                    s_fill_hi = s_chunk_end;
                    s_chunk_at = atomicAdd(a.rec_cursor, (unsigned long long)a.rec_chunk);
                    s_chunk_end = s_chunk_at + a.rec_chunk;
                }
                s_rec_base = s_chunk_at;
                s_chunk_at += n;
            }
            __syncthreads();
            for (unsigned long long at = s_fill_lo + tid; at < s_fill_hi && at < a.rec_cap; at += kKcThreads) a.rec_out[at] = ~0ull;
            for (uint32_t i = tid; i < n; i += kKcThreads) {
                const unsigned long long at = s_rec_base + i;
                if (at < a.rec_cap) a.rec_out[at] = s_rec[i];
                else kc_emit_overflow(a, s_rec[i]);                                                        // (no room)
            }
        }
    }
    if (EMIT) {
        __syncthreads();
        for (unsigned long long at = s_chunk_at + tid; at < s_chunk_end && at < a.rec_cap; at += kKcThreads) a.rec_out[at] = ~0ull;
    }
    for (int off = 32; off > 0; off >>= 1) counted += __shfl_down(counted, off, 64);
    if ((tid & 63) == 0 && counted) atomicAdd(a.total + a.parent, counted);
}

// ---- the emit front end of the default geometries (m = 16, W = K - 15 = 2 .. 6: K = 17 .. 21) -------------------------------------
// k_kc_count<W, EMIT> above spends 152 lane-instructions per window, a lane per position and then a lane per window.  Here a lane
// takes FOUR consecutive positions, then four consecutive windows:
//   M: the 32 bases behind position 4i as one 64-bit word and its reverse complement, computed ONCE; the four 16-mers and their
//      reverse complements are 32-bit funnel shifts of the two; canonical = min; kc_mmer_hash32 (24-bit multiplies); one 16-byte store.
//   B: the 9 = W + 3 hashes behind window 4g (three LDS reads) get their index in the block ORed into the four zero bits at their
//      bottom -- ONE unsigned minimum then orders by (hash, position) -- and the four window minima share their middle; a window
//      continues the run of the one in front of it iff both are valid and name the same position (one neighbour lane for the first
//      of the four); a run's length comes from the continuation bits of this lane and the next two (a run is at most W <= 6 windows);
//      the STARTS of runs are counted over the wave step (256 windows) with four ballots and written as 4-byte descriptors
//      (window, run, offset of the m-mer) -- one LDS atomic per wave step.
//   C: a lane per DESCRIPTOR, densely: the run's bases out of the packed tile, the 8-byte record straight to its place in HBM.
// Records are what k_kc_count<W, true> writes (kc_common.h), cut at steps of 256 windows instead of 64.
constexpr uint32_t kKcE4Pad = 80;                                              // words of s_mh behind the tile's window starts
__host__ __device__ inline size_t kc_emit4_mh_at(uint32_t nw) { return (16 + (size_t)nw * 12 + 15) & ~(size_t)15; }
template <int WT>
__global__ void __launch_bounds__(kKcThreads) k_kc_emit4(KcCountArgs a) {
    static_assert(WT >= 2 && WT <= 6, "a run must end inside the next two lanes");
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t TB = a.tile_bases;                                // a multiple of 1024: four waves x 256 windows a round
    const uint32_t K = (uint32_t)a.k;
    const uint32_t span = TB + K - 1, NW = (span + 31) / 32 + 2;
    unsigned long long *s_tile = reinterpret_cast<unsigned long long *>(smem);
    unsigned long long *s_pack = s_tile + 2;                         // [NW]
    uint32_t *s_inv = reinterpret_cast<uint32_t *>(s_pack + NW);     // [NW] invalid-byte masks (bit 31 = first base)
    uint32_t *s_mh = reinterpret_cast<uint32_t *>(smem + kc_emit4_mh_at(NW));   // [TB + kKcE4Pad], 16-byte aligned
    uint32_t *s_desc = s_mh + TB + kKcE4Pad;                         // [TB]: a tile cannot hold more runs than windows
    __shared__ uint32_t s_nrec;
    __shared__ unsigned long long s_rec_base, s_chunk_at, s_chunk_end, s_fill_lo, s_fill_hi;     // (chunks: as in k_kc_count)
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t n_tiles = (a.n_starts + TB - 1) / TB;
    const uintptr_t base_addr = reinterpret_cast<uintptr_t>(a.bytes);
    const uintptr_t end_addr = base_addr + a.n_bytes;
    const uint32_t ob = a.rec_off_bits;
    unsigned long long counted = 0;

    if (tid == 0) {
        *s_tile = (unsigned long long)blockIdx.x;
        s_chunk_at = s_chunk_end = 0;
    }
    __syncthreads();
    for (;;) {
        const uint64_t tile = *s_tile;
        if (tile >= n_tiles) break;
        const uint64_t t0 = tile * TB;
        __syncthreads();
        if (tid == 0) {
            *s_tile = tile + gridDim.x;
            s_nrec = 0;
        }
        kc_tile_pack(base_addr, end_addr, t0, NW, s_pack, s_inv, tid);
        __syncthreads();

        // ---- M ---------------------------------------------------------------------------------------------------------------
        for (uint32_t q0 = 4 * tid; q0 < TB + WT - 1; q0 += 4 * kKcThreads) {
            uint32_t hh[4];
            kc_e4_mmer_hashes(window_bits(s_pack, q0, 0), hh);                                     // from bases q0 .. q0 + 31
            const uint4 h{hh[0], hh[1], hh[2], hh[3]};
            *reinterpret_cast<uint4 *>(s_mh + q0) = h;
        }
        __syncthreads();

        // ---- B ---------------------------------------------------------------------------------------------------------------
        const uint64_t left = a.n_starts - t0;
        const uint32_t nwin = left < TB ? (uint32_t)left : TB;
        for (uint32_t base = wave * 256; base < nwin; base += (kKcThreads / 64) * 256) {          // wave-uniform
            const uint32_t p0 = base + 4 * lane;
            uint32_t c[9];
            {
                const uint4 u0 = *reinterpret_cast<const uint4 *>(s_mh + p0), u1 = *reinterpret_cast<const uint4 *>(s_mh + p0 + 4);
                c[0] = u0.x; c[1] = u0.y | 1u; c[2] = u0.z | 2u; c[3] = u0.w | 3u;
                c[4] = u1.x | 4u; c[5] = u1.y | 5u; c[6] = u1.z | 6u; c[7] = u1.w | 7u;
                c[8] = s_mh[p0 + 8] | 8u;
            }
            uint32_t w[4];
            kc_e4_minima<WT>(c, w);
            uint32_t at[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) at[k] = w[k] & 15u;                                          // where in the block the window's minimizer starts
            // validity: no separator in the K bytes of the window
            const uint32_t *iw = s_inv + (p0 >> 5);
            const uint32_t ihi = (uint32_t)(((((unsigned long long)iw[0]) << 32) | iw[1]) >> (32 - (p0 & 31)));   // bases p0 .. p0 + 31
            uint32_t vm = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bool v = p0 + k < nwin && ((ihi << k) >> (32 - K)) == 0;
                if (a.n_slices > 1 && kc_slice_of(w[k] & ~15u, a.n_slices) != a.slice) v = false;
                vm |= (uint32_t)v << k;
            }
            counted += (unsigned long long)__popc(vm);
            // continuation bits: window k goes on with the run of window k - 1
            const uint32_t mine = at[3] | ((vm >> 3) << 4);
            const uint32_t prev = (uint32_t)__shfl_up((int)mine, 1, 64);
            const uint32_t cn = kc_e4_cont(vm, at, lane > 0, prev);
            const uint32_t n1 = (uint32_t)__shfl_down((int)cn, 1, 64), n2 = (uint32_t)__shfl_down((int)cn, 2, 64);
            const uint32_t C = cn | (lane < 63 ? n1 << 4 : 0u) | (lane < 62 ? n2 << 8 : 0u);
            const uint32_t sn = vm & ~cn;                                                            // windows that start a run
            const unsigned long long b0 = __ballot(sn & 1u), b1 = __ballot(sn & 2u), b2 = __ballot(sn & 4u), b3 = __ballot(sn & 8u);
            const uint32_t total = (uint32_t)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
            if (total) {                                                                             // (wave-uniform)
                uint32_t idx = 0;
                if (lane == 0) idx = atomicAdd(&s_nrec, total);
                idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
                idx = __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, idx));
                idx = __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, idx));
                idx = __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, idx));
                idx = __builtin_amdgcn_mbcnt_hi((uint32_t)(b3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b3, idx));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if ((sn >> k) & 1u) {
                        s_desc[idx] = kc_e4_desc(p0 + (uint32_t)k, kc_e4_run_minus_1(C, k), at[k] - (uint32_t)k);
                        ++idx;
                    }
                }
            }
        }
        __syncthreads();

        // ---- C ---------------------------------------------------------------------------------------------------------------
        const uint32_t n = s_nrec;
        if (tid == 0) {
            s_fill_lo = s_fill_hi = 0;
            if (s_chunk_at + n > s_chunk_end) {                                        // (n <= TB <= a.rec_chunk)
                s_fill_lo = s_chunk_at;
                s_fill_hi = s_chunk_end;
                s_chunk_at = atomicAdd(a.rec_cursor, (unsigned long long)a.rec_chunk);
                s_chunk_end = s_chunk_at + a.rec_chunk;
            }
            s_rec_base = s_chunk_at;
            s_chunk_at += n;
        }
        __syncthreads();
        for (unsigned long long o = s_fill_lo + tid; o < s_fill_hi && o < a.rec_cap; o += kKcThreads) a.rec_out[o] = ~0ull;
        for (uint32_t i = tid; i < n; i += kKcThreads) {
            const uint32_t d = s_desc[i], p = d & 0x3FFFu, runm1 = (d >> 14) & 15u, off = d >> 18;
            const unsigned long long rec = (ob ? (unsigned long long)off << (64 - ob) : 0ull) | (window_bits(s_pack, p, 64 - 2 * (K + runm1)) << 6) |
                                           ((unsigned long long)runm1 << 1) | a.parent;
            const unsigned long long o = s_rec_base + i;
            if (o < a.rec_cap) a.rec_out[o] = rec;
            else kc_emit_overflow(a, rec);                                                             // (no room)
        }
    }
    __syncthreads();
    for (unsigned long long o = s_chunk_at + tid; o < s_chunk_end && o < a.rec_cap; o += kKcThreads) a.rec_out[o] = ~0ull;
    for (int off = 32; off > 0; off >>= 1) counted += __shfl_down(counted, off, 64);
    if (lane == 0 && counted) atomicAdd(a.total + a.parent, counted);
}
// does k_kc_emit4 take this launch?  (HAST_KC_EMIT=lanes keeps k_kc_count<W, true>: A/B runs, and the tests of that kernel)
static bool kc_emit4_takes(const KcCountArgs &a) {
    const char *e = getenv("HAST_KC_EMIT");
    const bool off = e && !strcmp(e, "lanes");
    const int w = a.k - a.m + 1;
    return !off && a.rec_out && a.m == 16 && w >= 2 && w <= 6 && a.rec_run_max >= (uint32_t)w && a.k + w + 2 <= 32 && a.tile_bases % 1024 == 0 &&
           a.tile_bases <= 16384 && a.rec_chunk >= a.tile_bases;
}
template <int WT>
static hipError_t launch_kc_emit4(const KcCountArgs &a, unsigned grid, hipStream_t s) {
    const uint32_t nw = (a.tile_bases + (uint32_t)a.k - 1 + 31) / 32 + 2;
    const size_t smem = kc_emit4_mh_at(nw) + ((size_t)a.tile_bases + kKcE4Pad) * 4 + (size_t)a.tile_bases * 4;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_emit4<WT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_kc_emit4<WT>), dim3(grid), dim3(kKcThreads), smem, s, a);
    return hipGetLastError();
}

template <int WT, bool EMIT>
static hipError_t launch_kc_count_e(const KcCountArgs &a, unsigned grid, size_t smem0, hipStream_t s) {
    const size_t smem = EMIT ? ((smem0 + 15) & ~(size_t)15) + (size_t)(a.tile_bases / 2) * 8 + 16 : smem0;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_count<WT, EMIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_kc_count<WT, EMIT>), dim3(grid), dim3(kKcThreads), smem, s, a);
    return hipGetLastError();
}
template <int WT>
static hipError_t launch_kc_count_t(const KcCountArgs &a, unsigned grid, size_t smem, hipStream_t s) {
    return a.rec_out ? launch_kc_count_e<WT, true>(a, grid, smem, s) : launch_kc_count_e<WT, false>(a, grid, smem, s);
}
size_t kc_count_smem(uint32_t tile_bases, int k, int m) {
    const uint32_t span = tile_bases + (uint32_t)k - 1, nw = (span + 31) / 32 + 2, w = (uint32_t)(k - m + 1);
    // + 64 entries: the last wave-block of a tile looks up to 63 window starts past the tile (those lanes are masked off afterwards)
    return 16 + (size_t)nw * 8 + (size_t)nw * 4 + (size_t)(tile_bases + w + 64) * 4 + 16;
}
hipError_t launch_kc_count(const KcCountArgs &a, unsigned grid, hipStream_t s) {
    if (kc_emit4_takes(a)) {
        switch (a.k - a.m + 1) {
        case 2: return launch_kc_emit4<2>(a, grid, s);
        case 3: return launch_kc_emit4<3>(a, grid, s);
        case 4: return launch_kc_emit4<4>(a, grid, s);
        case 5: return launch_kc_emit4<5>(a, grid, s);
        default: return launch_kc_emit4<6>(a, grid, s);
        }
    }
    const size_t smem = kc_count_smem(a.tile_bases, a.k, a.m);
    switch (a.k - a.m + 1) {
    case 1: return launch_kc_count_t<1>(a, grid, smem, s);
    case 6: return launch_kc_count_t<6>(a, grid, smem, s);
    case 8: return launch_kc_count_t<8>(a, grid, smem, s);
    case 9: return launch_kc_count_t<9>(a, grid, smem, s);
    default: return launch_kc_count_t<0>(a, grid, smem, s);
    }
}

// ---- partitioned counting ---------------------------------------------------------------------------------------------------------
// k_kc_count (above) does one memory-side atomic per minimizer run and wave step, and this part executes 27 G of them per second
// whatever their scope or footprint (tools/l2_atomics_probe): 81 per 150-bp read = the kernel's wall.  The
// way past it is the partitioned commit of stage 01 applied to the table itself: the windows are written out as records (EMIT),
// the records are PARTITIONED by bucket range in two levels -- in LDS, so that every bin receives whole runs -- down to bins of
// 512 or 1024 buckets, and then one workgroup per bin holds its slice of the table in LDS, counts its records there with LDS
// atomics and writes the slice back: the table is read and written once per flush, sequentially, and no atomic leaves the chip
// except one reservation per (workgroup, bin) and the few windows whose buckets are full beyond the slice (k_kc_spill).
//   HBM traffic per window: 8 B x 6 / ~3.3 windows per record + the sweep of the table (11 B at 30x) ~ 25 B, against one 128-B
//   line read + one atomic per minimizer run before.
// Measured (round 4, bench.py --workload s00, 10.4 G windows, 64-GB table; DESIGN.md section 9 has the steps): emit 51 ms + two
// partition passes 22 + 17 ms + k_kc_apply 50 ms (+ 9 ms to clear the table) = 0.148 s a step = 81 Gbp/s against 0.271 s of the direct
// kernel, 3.95 KB of HBM traffic per 150-bp read against 9.59 KB; every pass is bound by VALU issue.  Round 5: emit 28 ms (k_kc_emit4) + 12 +
// 13 ms (a level-1 bin in 8 regions: the workgroups no longer all add to the same fill words) + 47 ms, nothing cleared (a table that
// holds nothing is declared empty, KcFlushArgs.fresh) = 0.101 s = 119 Gbp/s, 3.0 KB per read.  The first version (a lane per
// RECORD in the LDS pass, the minimizer recomputed in every pass, keys filed in their minimizer's bucket as in the direct layout,
// one reservation per tile on one global word) took 0.58 s.
struct KcPartGeom {
    unsigned long long *table;
    uint32_t nbuckets;
    int k, m;
    uint32_t fine_shift;         // log2(buckets per fine bin): 9 or 10
    uint32_t n_fine, n_l1, f2;   // fine bins, level-1 bins, fine bins per level-1 bin (n_l1 * f2 >= n_fine)
    uint32_t *err;
    uint32_t ob, rmax;           // kc_rec_off_bits, kc_run_max
    uint32_t f2_shift;           // f2 == 1 << f2_shift
    uint32_t l1_split;           // regions per level-1 bin (KcFlushArgs)
    uint32_t fresh;              // KcFlushArgs
};
constexpr uint32_t kKcPartRecs = 8192, kKcPartThreads = 1024, kKcMaxFan = 1024;
constexpr uint32_t kKcFillPad = kKcL1FillWords;                                // words between two level-1 fill counters
__device__ __forceinline__ uint32_t kc_rec_fine(const KcPartGeom &g, unsigned long long rec) {
    if (g.m <= 16) {                                                             // (wave-uniform) the m-mer is one dword: a third of the instructions
        const uint32_t runm1 = (uint32_t)(rec >> 1) & 31u, off = g.ob ? (uint32_t)(rec >> (64 - g.ob)) : 0u;
        const uint32_t f = (uint32_t)(rec >> (6 + 2 * ((uint32_t)(g.k - g.m) + runm1 - off))) & (uint32_t)kmer_mask(g.m);
        uint32_t r = __brev(f ^ 0xAAAAAAAAu);
        r = (((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u)) >> (32 - 2 * g.m);
        return kc_fine_of_hash(kc_mmer_hash32(f < r ? f : r), g.n_fine);
    }
    return kc_fine_of_hash(kc_rec_minhash(rec, g.k, g.m, g.ob), g.n_fine);
}
// records that found no room in a bin: to the spill list, or -- that one full too -- through the atomic path at once (no slice
// of the table is held in LDS while a partition kernel runs)
__device__ __forceinline__ void kc_spill(const KcPartGeom &g, unsigned long long rec, unsigned long long *spill, unsigned long long spill_cap,
                                         unsigned long long *spill_n) {
    const unsigned long long at = atomicAdd(spill_n, 1ull);
    if (at < spill_cap) spill[at] = rec;
    else if (g.fresh) atomicOr(g.err, 1u);                                      // (no table to count into yet: as good as full)
    else kc_count_record(g.table, g.nbuckets, g.k, g.m, g.fine_shift, rec, g.err);
}
// a workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global access in flight (its fence is for
// all memory), which is exactly what k_kc_part wants to keep in flight across its barriers
__device__ __forceinline__ void kc_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// LEVEL 1: in = the flat record buffer [0, n_flat); bin = fine / f2, region = bin.  LEVEL 2: blockIdx.y = a level-1 bin, in = its
// region (in_cap records apart, in_fill / in_valid say how many are real); bin = fine - l1 * f2, region = fine.
// A workgroup takes 8192 records: (1) their bins, counted in LDS (a record's rank in its bin is what the LDS add returns); (2) a scan
// of the counts and ONE reservation per (workgroup, bin) in the bin's region -- issued here, looked at after (3), so that the
// round trip to the memory side runs beside the scatter; (3) the records to their bin's place in LDS, with the low byte of the bin
// beside them; (4) a lane per RECORD in bin order: consecutive lanes write consecutive records of a region.
// Round 4 - 5a: the bin from the canonical m-mer by 64-bit arithmetic and two quarter-rate multiplies (60 instructions), the records out
// by 16 lanes per bin (8 records a bin: half the lanes idle, 25 instructions per bin and lane): 115 lane-instructions per record, and
// the reservation's round trip in front of a barrier -- 22 + 16 ms a step at 41 % VALU issue.
template <int LEVEL>
__global__ void __launch_bounds__(kKcPartThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) k_kc_part(KcPartGeom g, const unsigned long long *in, unsigned long long n_flat, const uint32_t *in_fill,
                                                            const uint32_t *in_valid, uint32_t in_cap, unsigned long long *out, uint32_t out_cap,
                                                            uint32_t *out_fill, uint32_t *out_valid, unsigned long long *spill, unsigned long long spill_cap,
                                                            unsigned long long *spill_n) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(smem);                       // [kKcMaxFan] records of this tile per bin        } later: the bins' low
    uint32_t *s_off = s_cnt + kKcMaxFan;                                        // where the bin starts in s_rec                   } bytes, [kKcPartRecs]
    uint8_t *s_bin8 = reinterpret_cast<uint8_t *>(smem);
    uint32_t *s_dst = s_off + kKcMaxFan;                                        // (place of the bin's run in its region) - (its start in s_rec); all ones: no room
    unsigned long long *s_rec = reinterpret_cast<unsigned long long *>(s_dst + kKcMaxFan);   // [kKcPartRecs], grouped by bin
    static_assert(kKcPartRecs <= 2 * kKcMaxFan * 4, "the bins' low bytes take the place of s_cnt and s_off");
    __shared__ uint32_t s_scan[kKcPartThreads / 64];
    __shared__ uint32_t s_first[3];                                             // first slot of bins 256, 512, 768
    __shared__ uint32_t s_lost[kKcMaxFan / 32 + 1];                             // [0]: some bin of this tile found no room; [1 + b / 32] bit b % 32: bin b did not
    const uint32_t tid = threadIdx.x;
    // LEVEL 1 writes a bin's records to ONE OF l1_split regions (by workgroup): every workgroup of the pass adds to the fill words of all
    // bins -- 385 k workgroups x 960 adds on 30 lines of memory with one region per bin; level 2 takes the regions one by one
    const uint32_t l1 = LEVEL == 2 ? blockIdx.y % g.n_l1 : 0;
    const uint32_t n_bins = LEVEL == 1 ? g.n_l1 : g.f2;
    unsigned long long n_in = n_flat;
    const unsigned long long *src = in;
    if (LEVEL == 2) {
        const uint32_t f = in_fill[blockIdx.y * kKcFillPad], v = in_valid[blockIdx.y * kKcFillPad];
        n_in = f < v ? (f < in_cap ? f : in_cap) : v;
        src = in + (size_t)blockIdx.y * in_cap;
    }
    const unsigned long long r0 = (unsigned long long)blockIdx.x * kKcPartRecs;
    if (r0 >= n_in) return;
    const uint32_t nr = (uint32_t)(n_in - r0 < kKcPartRecs ? n_in - r0 : kKcPartRecs);
    constexpr int PER = kKcPartRecs / kKcPartThreads;                           // 8 records per thread
    unsigned long long rec[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const uint32_t i = (uint32_t)q * kKcPartThreads + tid;                  // coalesced
        rec[q] = i < nr ? src[r0 + i] : ~0ull;                                  // (all ones: a chunk's unused end, k_kc_count<EMIT>)
    }
    for (uint32_t b = tid; b < kKcMaxFan; b += kKcPartThreads) s_cnt[b] = 0;
    if (tid <= kKcMaxFan / 32) s_lost[tid] = 0;
    __syncthreads();
    uint32_t bin_of[PER], rank[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        bin_of[q] = 0xFFFFFFFFu;
        if (rec[q] != ~0ull) {
            const uint32_t fine = kc_rec_fine(g, rec[q]);
            uint32_t b = LEVEL == 1 ? fine >> g.f2_shift : fine - (l1 << g.f2_shift);
            if (b >= n_bins) b = n_bins - 1;                                    // (cannot happen: a record lies where its bucket says)
            bin_of[q] = b;
            rank[q] = atomicAdd(&s_cnt[b], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of s_cnt (one bin per thread), and one reservation per (workgroup, bin)
    const uint32_t c = s_cnt[tid];                                              // (zero behind n_bins)
    uint32_t base;
    {
        uint32_t incl = c;
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false);
        if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
        base = incl - c;
    }
    const uint32_t sub = LEVEL == 1 ? (blockIdx.x % g.l1_split) * g.n_l1 : 0;
    const uint32_t region = LEVEL == 1 ? sub + tid : l1 * g.f2 + tid;
    const uint32_t fi = LEVEL == 1 ? region * kKcFillPad : region;
    uint32_t at = 0;
    // (level 1: every workgroup adds to every one of <= 1024 counters -- a line apart, or they share 32 lines)
    // (a region that has failed takes no more adds: its fill word would otherwise keep growing with every workgroup that comes
    // by and, on a flush of billions of records into one bin -- poly-A -- wrap past 2^32 and hand out room again)
    if (c) at = __hip_atomic_load(&out_valid[fi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0xFFFFFFFFu ? 0xFFFFFFFFu : atomicAdd(&out_fill[fi], c);
    kc_lds_barrier();
    {
        const uint32_t wv = tid >> 6;
        uint32_t add = 0;
#pragma unroll
        for (uint32_t w = 0; w < kKcPartThreads / 64; ++w) add += w < wv ? s_scan[w] : 0u;       // (broadcast reads)
        base += add;
    }
    uint32_t off_of[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) off_of[q] = 0;
    s_off[tid] = base;
    if ((tid & 255) == 0 && tid) s_first[(tid >> 8) - 1] = base;
    kc_lds_barrier();
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (bin_of[q] != 0xFFFFFFFFu) off_of[q] = s_off[bin_of[q]] + rank[q];
    kc_lds_barrier();                                                           // s_cnt and s_off are free: the bins' low bytes go there
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (bin_of[q] != 0xFFFFFFFFu) {
            s_rec[off_of[q]] = rec[q];
            s_bin8[off_of[q]] = (uint8_t)bin_of[q];
        }
    {   // the reservation has had the scatter's time to come back
        uint32_t dst = 0;
        if (c) {
            if (at != 0xFFFFFFFFu && (uint64_t)at + c <= out_cap) dst = at - base;
            else {
                if (at != 0xFFFFFFFFu) atomicMin(&out_valid[fi], at);           // records [0, first failed reservation) of a region are real
                s_lost[0] = 1;
                atomicOr(&s_lost[1 + (tid >> 5)], 1u << (tid & 31));
            }
        }
        s_dst[tid] = dst;
    }
    __syncthreads();
    // a lane per record, in bin order
    const uint32_t f1 = s_first[0], f2 = s_first[1], f3 = s_first[2];
    uint32_t n_valid = 0;
#pragma unroll
    for (uint32_t w = 0; w < kKcPartThreads / 64; ++w) n_valid += s_scan[w];
    const bool lost = s_lost[0] != 0;
    for (uint32_t i = tid; i < n_valid; i += kKcPartThreads) {
        const unsigned long long r = s_rec[i];
        const uint32_t b = (uint32_t)s_bin8[i] + ((i >= f1 ? 256u : 0u) + (i >= f2 ? 256u : 0u) + (i >= f3 ? 256u : 0u));
        const uint32_t d = s_dst[b];
        if (lost && ((s_lost[1 + (b >> 5)] >> (b & 31)) & 1u)) {               // (wave-uniform test first: regions rarely fill up)
            kc_spill(g, r, spill, spill_cap, spill_n);
            continue;
        }
        const uint32_t reg = LEVEL == 1 ? sub + b : l1 * g.f2 + b;
        out[(size_t)reg * out_cap + (uint32_t)(d + i)] = r;
    }
}

// One workgroup per fine bin: its slice of the table in LDS, its records counted there, the slice written back.  A window whose
// bucket and the next three are full (or lie behind the slice's end) goes to the spill list as a record of one window; a full
// spill list is a full table as far as the caller is concerned (err bit 0).
// 1024 threads: a slice takes 64 or 128 KB of LDS, so a CU holds one or two of these workgroups -- with 256 threads that was 8 waves
// per CU, every one of them waiting on its own chain of loads (measured: 0.52 s per flush of the bench's 60-GB table, 270 GB/s).
// A LANE PER WINDOW, a whole bucket per probe: a wave takes 64 records, scans their run lengths, and spreads the windows of those
// records over its lanes (a byte map window -> record in LDS, the record's words fetched from the owning lane by ds_bpermute);
// a window reads the 8 key slots of its bucket at once, compares them all, and only a key that is not there yet goes through
// compare-and-swap on the first slot that looked empty.  The first version gave a lane a RECORD and walked the slots one LDS
// round trip at a time: runs of 1..9 windows x 1..32 slots, every wave as slow as its slowest lane -- 0.38 s of LDS probes per
// flush against 0.20 s for everything else.
// tag of a key in a slice's LDS copy: 7 bits of a hash of the key under a set top bit (0 = the slot is empty)
__device__ __forceinline__ uint32_t kc_tag(unsigned long long key) { return key == kEmptySlot ? 0u : kc_tag_of_hash(kc_key_hash(key)); }
// a bucket of the slice's LDS copy: 17 words = 136 bytes from the next (shifts: a 32-bit multiply is a quarter-rate instruction)
__device__ __forceinline__ unsigned long long *kc_lds_bucket(unsigned char *smem, uint32_t b) {
    return reinterpret_cast<unsigned long long *>(smem + (b << 7) + (b << 3));
}
static_assert(kKcBucketWords + 1 == 17, "kc_lds_bucket assumes 17-word buckets");
__device__ __forceinline__ unsigned long long kc_zero_bytes64(unsigned long long x) {      // 0x80 in every byte of x that is zero (exact)
    const unsigned long long m = 0x7F7F7F7F7F7F7F7Full;
    return ~(((x & m) + m) | x | m);
}
constexpr int kKcApplyThreads = 1024;
constexpr uint32_t kKcLdsStride = kKcBucketWords + 1;                            // words per bucket in LDS (k_kc_apply)
constexpr uint32_t kKcMapMax = 1024;                                             // bytes of window -> record map per wave, at most
__host__ __device__ inline uint32_t kc_apply_map_bytes(uint32_t rmax) { return 64u * rmax < kKcMapMax ? 64u * rmax : kKcMapMax; }
__global__ void __launch_bounds__(kKcApplyThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) k_kc_apply(KcPartGeom g, const unsigned long long *recs, uint32_t cap, const uint32_t *fill, const uint32_t *valid,
                                                              unsigned long long *spill, unsigned long long spill_cap, unsigned long long *spill_n) {
    extern __shared__ __align__(16) unsigned char smem[];
    // [buckets of the slice][kKcLdsStride]: a bucket is 16 words = 128 B = exactly the 32 LDS banks, so slot i of EVERY bucket would sit
    // in the same bank and the 64 lanes of a probe (64 different buckets, the same slot) would take 64 turns at it; one word of
    // padding per bucket spreads them
    const uint32_t fine = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t f = fill[fine], v = valid[fine];
    const uint32_t n = f < v ? (f < cap ? f : cap) : v;
    if (n == 0 && !g.fresh) return;
    const uint32_t b0 = fine << g.fine_shift;
    const uint32_t nb_here = g.nbuckets - b0 < (1u << g.fine_shift) ? g.nbuckets - b0 : (1u << g.fine_shift);
    const uint32_t map_bytes = kc_apply_map_bytes(g.rmax);
    uint8_t *s_map = smem + ((size_t)1 << g.fine_shift) * kKcLdsStride * 8 + (size_t)wave * map_bytes;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    {
        const u64x2 *gsrc = reinterpret_cast<const u64x2 *>(g.table + (size_t)b0 * kKcBucketWords);
        const uint32_t nvec = nb_here * (kKcBucketWords / 2);
        for (uint32_t i0 = tid; i0 < nvec; i0 += 4 * kKcApplyThreads) {            // four loads in flight per lane
            u64x2 t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t i = i0 + (uint32_t)q * kKcApplyThreads;
                if (g.fresh) t[q] = (i & 7) < kKcSlots / 2 ? u64x2{kEmptySlot, kEmptySlot} : u64x2{0ull, 0ull};      // (wave-uniform) a table nobody has written: k_kc_clear's pattern
                else if (i < nvec) t[q] = gsrc[i];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t i = i0 + (uint32_t)q * kKcApplyThreads;
                if (i < nvec) {
                    unsigned long long *d = kc_lds_bucket(smem, i >> 3) + 2 * (i & 7);
                    d[0] = t[q].x;
                    d[1] = t[q].y;
                    // the bucket's 17th word: a TAG byte per key slot (0 = empty), so that a probe reads 8 bytes, not 8 keys
                    if ((i & 7) < kKcSlots / 2)
                        reinterpret_cast<uint16_t *>(kc_lds_bucket(smem, i >> 3) + kKcBucketWords)[i & 7] = (uint16_t)(kc_tag(t[q].x) | (kc_tag(t[q].y) << 8));
                }
            }
        }
    }
    const unsigned long long *mine = recs + (size_t)fine * cap;
    const unsigned long long kmask = kmer_mask(g.k);
    const uint32_t rmax = g.rmax;
    const uint32_t RS = map_bytes / rmax < 64u ? map_bytes / rmax : 64u;       // records of a wave step (64 unless K is tiny)
    // (the records of a step are fetched one step ahead, the first ones beside the slice itself)
    unsigned long long rec_next = lane < RS && wave * RS + lane < n ? mine[wave * RS + lane] : 0ull;
    __syncthreads();
    for (uint32_t base = wave * RS; base < n; base += (kKcApplyThreads / 64) * RS) {               // wave-uniform
        const bool has = lane < RS && base + lane < n;
        const unsigned long long rec = rec_next;
        {
            const uint32_t nx = base + (kKcApplyThreads / 64) * RS + lane;
            rec_next = lane < RS && nx < n ? mine[nx] : 0ull;
        }
        const uint32_t run = has ? (uint32_t)((rec >> 1) & 31) + 1 : 0u;
        // (prefix sum in the VALU's own data path: six ds_bpermute rounds queued behind the probes' LDS traffic)
        uint32_t incl = run;
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false);
        const uint32_t P = incl - run, T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        for (uint32_t j = 0; j < run; ++j) s_map[P + j] = (uint8_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t w0 = 0; w0 < T; w0 += 64) {
            const uint32_t w = w0 + lane;
            const bool act = w < T;
            const uint32_t src = act ? s_map[w] : lane;
            const unsigned long long rw = __shfl(rec, (int)src, 64);
            const uint32_t pw = (uint32_t)__shfl((int)P, (int)src, 64);
            const uint32_t parent = (uint32_t)(rw & 1), run_w = (uint32_t)((rw >> 1) & 31) + 1;
            const uint32_t jw = act ? w - pw : 0u;                                 // window jw of its record
            const unsigned long long raw = ((rw >> 6) >> (2 * (run_w - 1 - jw))) & kmask;
            const unsigned long long key = kmer_canon(raw, g.k);
            const uint32_t kh = kc_key_hash(key);
            const uint32_t hw = kc_bucket_of_hash(kh, nb_here);
            const uint32_t tag = kc_tag_of_hash(kh);
            const uint32_t t32 = __builtin_amdgcn_perm(tag, tag, 0u);              // the tag in all four bytes (shifts + ors become a quarter-rate multiply)
            const unsigned long long tagx = ((unsigned long long)t32 << 32) | t32;
            bool done = !act;
#pragma unroll 1
            for (uint32_t p = 0; p < 4; ++p) {                                     // (kc_probe: the key's bucket in the slice, then the next three)
                if (!__ballot(!done)) break;
                uint32_t b = hw + p;
                b = b >= nb_here ? b - nb_here : b;
                if (!done) {
                    unsigned long long *bk = kc_lds_bucket(smem, b);
                    // ONE 8-byte read of the bucket's tags instead of its eight keys (round 4: 8 LDS reads and 16 compare + select per
                    // window and probe -- 158 lane-instructions per window, 4.3 bank-conflict cycles per LDS instruction): a slot whose tag
                    // matches is read and compared (one in 128 occupied slots matches by chance), a key that is not there goes in by
                    // compare-and-swap from the first slot whose tag says empty, and writes its tag behind itself.  A tag that is not
                    // written yet only sends another lane with the same key into the same compare-and-swap, which then finds it there.
                    const unsigned long long tg = bk[kKcBucketWords];
                    const unsigned long long mt = kc_zero_bytes64(tg ^ tagx);
                    // the first slot whose tag matches, in straight-line code (a key that is there is found here; a second matching tag
                    // -- one occupied slot in 128 matches by chance -- is the loop's business)
                    const int i0 = __builtin_ctzll(mt | (1ull << 63)) >> 3;
                    int idx = (mt != 0 && bk[i0] == key) ? i0 : -1;
                    unsigned long long rest = idx < 0 ? (mt & (mt - 1)) : 0ull;
                    while (rest) {
                        const int i = __builtin_ctzll(rest) >> 3;
                        rest &= rest - 1;
                        if (bk[i] == key) { idx = i; break; }
                    }
                    if (idx < 0) {
                        const unsigned long long em = kc_zero_bytes64(tg);
                        // slots fill in order and never change: a key not seen is settled by compare-and-swap from the first slot that looked empty
                        for (int i = em ? __builtin_ctzll(em) >> 3 : kKcSlots; idx < 0 && i < kKcSlots; ++i) {
                            const unsigned long long old = atomicCAS(&bk[i], (unsigned long long)kEmptySlot, key);
                            if (old == kEmptySlot) reinterpret_cast<uint8_t *>(bk + kKcBucketWords)[i] = (uint8_t)tag;
                            if (old == kEmptySlot || old == key) idx = i;
                        }
                    }
                    if (idx >= 0) {
                        atomicAdd(reinterpret_cast<uint32_t *>(bk + kKcSlots) + parent * kKcSlots + idx, 1u);
                        done = true;
                    }
                }
            }
            const bool lost = !done;                                                // four full buckets: the atomic path's business
            // (one reservation per wave and step, not per window: adds to one address serialise at the memory side)
            const unsigned long long lm = __ballot(lost);
            if (lm) {
                const uint32_t first = (uint32_t)__builtin_ctzll(lm);
                unsigned long long at0 = 0;
                if (lane == first) at0 = atomicAdd(spill_n, (unsigned long long)__popcll(lm));
                at0 = __shfl(at0, (int)first, 64);
                if (lost) {
                    const unsigned long long at = at0 + (unsigned long long)__popcll(lm & ((1ull << lane) - 1));
                    // a record of one window: the minimizer lies jw bases nearer to its start
                    if (at < spill_cap) spill[at] = (g.ob ? (unsigned long long)((uint32_t)(rw >> (64 - g.ob)) - jw) << (64 - g.ob) : 0ull) | (raw << 6) | parent;
                    else atomicOr(g.err, 1u);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                     // (the map is rewritten by the next step)
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {
        u64x2 *gdst = reinterpret_cast<u64x2 *>(g.table + (size_t)b0 * kKcBucketWords);
        for (uint32_t i = tid; i < nb_here * (kKcBucketWords / 2); i += kKcApplyThreads) {
            const unsigned long long *sp = kc_lds_bucket(smem, i >> 3) + 2 * (i & 7);
            gdst[i] = u64x2{sp[0], sp[1]};
        }
    }
}
// the spill list through the atomic path (after k_kc_apply: nobody holds a slice any more)
__global__ void __launch_bounds__(256) k_kc_spill(KcPartGeom g, const unsigned long long *spill, unsigned long long spill_cap, const unsigned long long *spill_n) {
    const unsigned long long n = *spill_n < spill_cap ? *spill_n : spill_cap;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        if (spill[i] != ~0ull) kc_count_record(g.table, g.nbuckets, g.k, g.m, g.fine_shift, spill[i], g.err);
}

bool kc_flush_is_small(const KcFlushArgs &a) {
    return a.n_records && (a.small_flush == 2 || (a.small_flush == 0 && a.n_records < (unsigned long long)a.nbuckets * (kKcBucketWords * 8) / 256));
}
hipError_t launch_kc_flush(const KcFlushArgs &a, hipStream_t s) {
    KcPartGeom g{a.table, a.nbuckets, a.k, a.m, a.fine_shift, a.n_fine, a.n_l1, a.f2, a.err, kc_rec_off_bits(a.k, a.m), kc_run_max(a.k, a.m), (uint32_t)__builtin_ctz(a.f2), a.l1_split, a.fresh};
    const size_t lds_part = (size_t)3 * kKcMaxFan * 4 + (size_t)kKcPartRecs * 8;
    const size_t lds_apply = ((size_t)1 << a.fine_shift) * kKcLdsStride * 8 + (size_t)(kKcApplyThreads / 64) * kc_apply_map_bytes(kc_run_max(a.k, a.m));
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_part<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_part<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_part);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_apply), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_apply);
    if (e != hipSuccess) return e;
    // fills to zero, "valid" marks to all ones: [l1 fill | l1 valid | fine fill | fine valid]
    e = hipMemsetAsync(a.l1_fill, 0, (size_t)a.n_l1 * a.l1_split * kKcFillPad * 4, s);
    if (e == hipSuccess) e = hipMemsetAsync(a.l1_valid, 0xFF, (size_t)a.n_l1 * a.l1_split * kKcFillPad * 4, s);
    if (e == hipSuccess) e = hipMemsetAsync(a.fine_fill, 0, (size_t)a.n_fine * 4, s);
    if (e == hipSuccess) e = hipMemsetAsync(a.fine_valid, 0xFF, (size_t)a.n_fine * 4, s);
    if (e != hipSuccess) return e;
    if (kc_flush_is_small(a)) {
        // few records for this table (the sweep moves 2 x 128 B per bucket whatever there is to count, the atomic path ~40 G windows/s):
        // count them where they lie
        hipLaunchKernelGGL(k_kc_spill, dim3(256 * 8), dim3(256), 0, s, g, a.records, a.n_records, a.rec_cursor);
    } else if (a.n_records) {
        hipLaunchKernelGGL(k_kc_part<1>, dim3((unsigned)((a.n_records + kKcPartRecs - 1) / kKcPartRecs)), dim3(kKcPartThreads), lds_part, s, g, a.records, a.n_records,
                           (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, a.l1_recs, a.l1_cap, a.l1_fill, a.l1_valid, a.spill, a.spill_cap, a.spill_n);
        // level 2 overwrites the flat buffer (its records are all in the level-1 regions by then)
        hipLaunchKernelGGL(k_kc_part<2>, dim3((a.l1_cap + kKcPartRecs - 1) / kKcPartRecs, a.n_l1 * a.l1_split), dim3(kKcPartThreads), lds_part, s, g, a.l1_recs, 0ull, a.l1_fill,
                           a.l1_valid, a.l1_cap, a.records, a.fine_cap, a.fine_fill, a.fine_valid, a.spill, a.spill_cap, a.spill_n);
        hipLaunchKernelGGL(k_kc_apply, dim3(a.n_fine), dim3(kKcApplyThreads), lds_apply, s, g, a.records, a.fine_cap, a.fine_fill, a.fine_valid, a.spill, a.spill_cap, a.spill_n);
    }
    hipLaunchKernelGGL(k_kc_spill, dim3(256 * 8), dim3(256), 0, s, g, a.spill, a.spill_cap, a.spill_n);
    return hipGetLastError();
}

// empty table: keys all ones, counters zero
__global__ void __launch_bounds__(256) k_kc_clear(unsigned long long *table, size_t nbuckets) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2 *v = reinterpret_cast<u64x2 *>(table);
    const size_t n = nbuckets * (kKcBucketWords / 2);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        v[i] = (i % (kKcBucketWords / 2)) < kKcSlots / 2 ? u64x2{kEmptySlot, kEmptySlot} : u64x2{0ull, 0ull};
}

// ---- passes over the table -------------------------------------------------------------------------------------
// out[0], out[1] = distinct keys per parent, out[2] = keys in the table
__global__ void __launch_bounds__(256) k_kc_stats(const unsigned long long *table, size_t nbuckets, unsigned long long *out) {
    unsigned long long c0 = 0, c1 = 0, cu = 0;
    const size_t nslots = nbuckets * kKcSlots;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nslots; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long *bk = table + (i / kKcSlots) * kKcBucketWords;
        if (bk[i % kKcSlots] == kEmptySlot) continue;
        const uint32_t *c = reinterpret_cast<const uint32_t *>(bk + kKcSlots) + i % kKcSlots;
        cu++;
        c0 += c[0] != 0;
        c1 += c[kKcSlots] != 0;
    }
    for (int off = 32; off > 0; off >>= 1) {
        c0 += __shfl_down(c0, off, 64);
        c1 += __shfl_down(c1, off, 64);
        cu += __shfl_down(cu, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (c0) atomicAdd(&out[0], c0);
        if (c1) atomicAdd(&out[1], c1);
        if (cu) atomicAdd(&out[2], cu);
    }
}

// histogram of one parent's counts: out[min(c, high + 1)]++ for every key with c > 0.  Counts 1 and 2 (sequencing
// errors: the bulk of all keys) are tallied in registers, the rest in an LDS histogram; one global add per bin and block.
__global__ void __launch_bounds__(256) k_kc_histo(const unsigned long long *table, size_t nbuckets, uint32_t parent,
                                                  unsigned long long *out) {
    __shared__ uint32_t h[kKcHistoHigh + 2];
    for (uint32_t i = threadIdx.x; i < kKcHistoHigh + 2; i += blockDim.x) h[i] = 0;
    __syncthreads();
    uint32_t n1 = 0, n2 = 0;
    const size_t nslots = nbuckets * kKcSlots;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nslots; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long *bk = table + (i / kKcSlots) * kKcBucketWords;
        if (bk[i % kKcSlots] == kEmptySlot) continue;
        const uint32_t c = reinterpret_cast<const uint32_t *>(bk + kKcSlots)[parent * kKcSlots + i % kKcSlots];
        if (c == 1) n1++;
        else if (c == 2) n2++;
        else if (c) atomicAdd(&h[c > kKcHistoHigh ? kKcHistoHigh + 1 : c], 1u);
    }
    if (n1) atomicAdd(&h[1], n1);
    if (n2) atomicAdd(&h[2], n2);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kKcHistoHigh + 2; i += blockDim.x)
        if (h[i]) atomicAdd(&out[i], (unsigned long long)h[i]);
}

// keys of `parent` with lower <= count <= upper that the other parent does not have at all
// (build_unshared_kmers.sh:246-291), as print keys.  out == nullptr: count only.
__global__ void __launch_bounds__(256) k_kc_select(const unsigned long long *table, size_t nbuckets, uint32_t parent, uint32_t lower,
                                                   uint32_t upper, int k, unsigned long long *out, size_t cap,
                                                   unsigned long long *cursor) {
    // the selected keys of kSelRounds rounds are gathered in LDS and leave with ONE reservation per workgroup: with keys spread
    // evenly over the buckets (kc_common.h PLACEMENT) every wave of every round holds one, and one add per wave on the one cursor
    // word -- 58 M of them over a 60-GB table -- took 86 ms a pass where the scan itself takes 15
    constexpr int kSelRounds = 8;
    __shared__ unsigned long long s_buf[kSelRounds * 256];
    __shared__ uint32_t s_n;
    __shared__ unsigned long long s_base;
    const size_t nslots = nbuckets * kKcSlots;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t rounds = (nslots + stride - 1) / stride;                       // (the same for every thread)
    const uint32_t lane = threadIdx.x & 63;
    unsigned long long n_mine = 0;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (size_t r0 = 0; r0 < rounds; r0 += kSelRounds) {
        for (size_t r = r0; r < r0 + kSelRounds && r < rounds; ++r) {
            const size_t i = r * stride + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
            bool take = false;
            unsigned long long key = 0;
            if (i < nslots) {
                const unsigned long long *bk = table + (i / kKcSlots) * kKcBucketWords;
                key = bk[i % kKcSlots];
                if (key != kEmptySlot) {
                    const uint32_t *c = reinterpret_cast<const uint32_t *>(bk + kKcSlots) + i % kKcSlots;
                    const uint32_t mine = c[parent * kKcSlots], other = c[(1 - parent) * kKcSlots];
                    take = mine >= lower && mine <= upper && other == 0;
                }
            }
            if (!out) {
                n_mine += take;
                continue;
            }
            const unsigned long long m = __ballot(take);
            if (m == 0) continue;
            const uint32_t first = (uint32_t)__ffsll((long long)m) - 1;
            uint32_t at0 = 0;
            if (lane == first) at0 = atomicAdd(&s_n, (uint32_t)__popcll(m));
            at0 = (uint32_t)__shfl((int)at0, (int)first, 64);
            if (take) s_buf[at0 + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = kc_to_print_key(key, k);
        }
        if (!out) continue;
        __syncthreads();
        const uint32_t n = s_n;
        if (threadIdx.x == 0 && n) s_base = atomicAdd(cursor, (unsigned long long)n);
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < n; j += 256)
            if (s_base + j < cap) out[s_base + j] = s_buf[j];
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
    }
    if (!out) {                                                                   // count only: one add per wave and launch
        for (int off = 32; off > 0; off >>= 1) n_mine += __shfl_down(n_mine, off, 64);
        if (lane == 0 && n_mine) atomicAdd(cursor, n_mine);
    }
}

// print keys -> text, one K-letter line each
__global__ void __launch_bounds__(256) k_kc_format(const unsigned long long *keys, size_t n, int k, char *text) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long key = keys[i];
        char *o = text + i * (size_t)(k + 1);
        for (int j = 0; j < k; ++j) o[j] = "ACGT"[(key >> (2 * (k - 1 - j))) & 3];
        o[k] = '\n';
    }
}
// print keys -> stage-01 table keys (for handing the sets to the classifier without a text round trip)
__global__ void __launch_bounds__(256) k_kc_to_table_keys(const unsigned long long *keys, size_t n, int k, unsigned long long *out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = kc_from_print_key(keys[i], k);
}

__global__ void __launch_bounds__(256) k_kc_synth(KcSynth g, int parent, uint64_t first_read, size_t n_bytes, uint8_t *out) {
    const uint32_t rec = g.read_len + 1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_bytes; i += (size_t)gridDim.x * blockDim.x)
        out[i] = kc_synth_byte(g, parent, first_read + i / rec, (uint32_t)(i % rec));
}

static unsigned stream_grid(size_t n) {
    const size_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 256 * 16 ? 256 * 16 : b));
}
hipError_t launch_kc_clear(unsigned long long *table, size_t nbuckets, hipStream_t s) {
    hipLaunchKernelGGL(k_kc_clear, dim3(stream_grid(nbuckets * (kKcBucketWords / 2))), dim3(256), 0, s, table, nbuckets);
    return hipGetLastError();
}
hipError_t launch_kc_stats(const unsigned long long *table, size_t nbuckets, unsigned long long *d_out3, hipStream_t s) {
    hipLaunchKernelGGL(k_kc_stats, dim3(stream_grid(nbuckets * kKcSlots)), dim3(256), 0, s, table, nbuckets, d_out3);
    return hipGetLastError();
}
hipError_t launch_kc_histo(const unsigned long long *table, size_t nbuckets, uint32_t parent, unsigned long long *d_out, hipStream_t s) {
    hipLaunchKernelGGL(k_kc_histo, dim3(stream_grid(nbuckets * kKcSlots) > 1024 ? 1024 : stream_grid(nbuckets * kKcSlots)), dim3(256), 0, s,
                       table, nbuckets, parent, d_out);
    return hipGetLastError();
}
hipError_t launch_kc_select(const unsigned long long *table, size_t nbuckets, uint32_t parent, uint32_t lower, uint32_t upper, int k,
                            unsigned long long *d_out, size_t cap, unsigned long long *d_cursor, hipStream_t s) {
    hipLaunchKernelGGL(k_kc_select, dim3(stream_grid(nbuckets * kKcSlots)), dim3(256), 0, s, table, nbuckets, parent, lower, upper, k,
                       d_out, cap, d_cursor);
    return hipGetLastError();
}
hipError_t launch_kc_format(const unsigned long long *d_keys, size_t n, int k, char *d_text, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_kc_format, dim3(stream_grid(n)), dim3(256), 0, s, d_keys, n, k, d_text);
    return hipGetLastError();
}
hipError_t launch_kc_to_table_keys(const unsigned long long *d_keys, size_t n, int k, unsigned long long *d_out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_kc_to_table_keys, dim3(stream_grid(n)), dim3(256), 0, s, d_keys, n, k, d_out);
    return hipGetLastError();
}
hipError_t launch_kc_synth(const KcSynth &g, int parent, uint64_t first_read, size_t n_bytes, uint8_t *d_out, hipStream_t s) {
    if (n_bytes == 0) return hipSuccess;
    hipLaunchKernelGGL(k_kc_synth, dim3(stream_grid(n_bytes)), dim3(256), 0, s, g, parent, first_read, n_bytes, d_out);
    return hipGetLastError();
}
// output order only: ascending print keys = lexicographic order of the lines (plain library sort)
hipError_t kc_sort_keys(void *d_tmp, size_t *tmp_bytes, unsigned long long *d_in, unsigned long long *d_out, size_t n, int k, hipStream_t s) {
    return rocprim::radix_sort_keys(d_tmp, *tmp_bytes, d_in, d_out, n, 0, (unsigned)(2 * k), s);
}

}  // namespace hast

// kc_kernels.hip -- gfx950 (MI355X, CDNA4) device code for HAST stage 00: counting the canonical k-mers of both
// parents' reads in ONE open-addressed table in HBM, then reading the parent-unique sets and the count histograms
// straight out of it (reference: 00.build_unshare_kmers_by_jellyfish/build_unshared_kmers.sh:165-291,
// where a third-party CPU hash counter does this in seven count/dump passes over text files).
//
// Integer + HBM work only (no MFMA by design):
//   k_kc_count   : byte stream -> 2-bit codes + validity mask in LDS -> canonical k-mer per window -> find-or-insert in
//                  the window's minimizer bucket (128-B line: 8 keys + 8 paternal + 8 maternal counters) -> one atomic add
//   k_kc_stats / k_kc_histo / k_kc_select : streaming passes over the table
//   k_kc_format  : selected keys -> text lines
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
__device__ __forceinline__ uint32_t *kc_slot(unsigned long long *table, uint32_t nb, uint32_t b, uint64_t key, uint32_t parent) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
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
        b = next_bucket(b, probe + 1, key, nb);              // bucket full: jump to the key's own overflow bucket, then walk on
    }
    return nullptr;
}

// One workgroup walks tiles of `tile_bases` window starts (+ K-1 bytes of overlap) of the byte stream.
template <int WT>
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
    uint32_t *s_mh = s_inv + NW;                                     // [TB + W]
    const uint32_t tid = threadIdx.x;
    const uint32_t kshift = 64 - 2 * K, mshift = 64 - 2 * M;
    const uint64_t n_tiles = (a.n_starts + TB - 1) / TB;
    const uintptr_t base_addr = reinterpret_cast<uintptr_t>(a.bytes);
    const uintptr_t end_addr = base_addr + a.n_bytes;                // bytes at and beyond it read as separators
    unsigned long long counted = 0;

    if (tid == 0) *s_tile = atomicAdd(a.tile_queue, 1ull);
    __syncthreads();
    for (;;) {
        const uint64_t tile = *s_tile;
        if (tile >= n_tiles) break;
        const uint64_t t0 = tile * TB;
        __syncthreads();
        if (tid == 0) *s_tile = atomicAdd(a.tile_queue, 1ull);

        // ---- A: 16 bytes per lane -> 32 bits of codes + 16 validity bits -----------------------------------
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
        __syncthreads();

        // ---- M: hash of the canonical m-mer at every position --------------------------------------------
        for (uint32_t q = tid; q < TB + W - 1; q += kKcThreads)
            s_mh[q] = mmer_hash32(kmer_canon(window_bits(s_pack, q, mshift), M));
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
            uint32_t mn = s_mh[p];
            if (WT) {
#pragma unroll
                for (int j = 1; j < (WT ? WT : 1); ++j) mn = min(mn, s_mh[p + j]);
            } else {
                for (uint32_t j = 1; j < W; ++j) mn = min(mn, s_mh[p + j]);
            }
            if (a.n_slices > 1 && kc_slice_of(mn, a.n_slices) != a.slice) valid = false;
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
    }
    for (int off = 32; off > 0; off >>= 1) counted += __shfl_down(counted, off, 64);
    if ((tid & 63) == 0 && counted) atomicAdd(a.total + a.parent, counted);
}

template <int WT>
static hipError_t launch_kc_count_t(const KcCountArgs &a, unsigned grid, size_t smem, hipStream_t s) {
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_kc_count<WT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_kc_count<WT>), dim3(grid), dim3(kKcThreads), smem, s, a);
    return hipGetLastError();
}
size_t kc_count_smem(uint32_t tile_bases, int k, int m) {
    const uint32_t span = tile_bases + (uint32_t)k - 1, nw = (span + 31) / 32 + 2, w = (uint32_t)(k - m + 1);
    // + 64 entries: the last wave-block of a tile looks up to 63 window starts past the tile (those lanes are masked off afterwards)
    return 16 + (size_t)nw * 8 + (size_t)nw * 4 + (size_t)(tile_bases + w + 64) * 4 + 16;
}
hipError_t launch_kc_count(const KcCountArgs &a, unsigned grid, hipStream_t s) {
    const size_t smem = kc_count_smem(a.tile_bases, a.k, a.m);
    switch (a.k - a.m + 1) {
    case 1: return launch_kc_count_t<1>(a, grid, smem, s);
    case 6: return launch_kc_count_t<6>(a, grid, smem, s);
    case 8: return launch_kc_count_t<8>(a, grid, smem, s);
    case 9: return launch_kc_count_t<9>(a, grid, smem, s);
    default: return launch_kc_count_t<0>(a, grid, smem, s);
    }
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
    const size_t nslots = nbuckets * kKcSlots;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t rounds = (nslots + stride - 1) / stride;
    for (size_t r = 0; r < rounds; ++r) {
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
        // one atomic per wave
        const unsigned long long m = __ballot(take);
        if (m == 0) continue;
        const uint32_t lane = threadIdx.x & 63;
        unsigned long long base = 0;
        if (lane == (uint32_t)__ffsll((long long)m) - 1) base = atomicAdd(cursor, (unsigned long long)__popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1, 64);
        if (take && out) {
            const size_t at = base + __popcll(m & ((1ull << lane) - 1));
            if (at < cap) out[at] = kc_to_print_key(key, k);
        }
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

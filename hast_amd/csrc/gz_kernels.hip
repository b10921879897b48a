// gz_kernels.hip -- gfx950 device code of the gzip inflate that feeds the FASTQ framer (reference: gzstream.h:47,
// classify.cpp:245-254 read each .gz input through ONE zlib stream on the host).  gz_core.h holds what a lane does with a
// chunk of the compressed bytes; here are the launches around it:
//   k_gz_search     a wave per chunk: 64 bit positions per step through the cheap header test (gz_core.h candidate), survivors
//                   parsed in full by their own lane, lowest position first
//   k_gz_decode     a LANE per chunk: blocks -> 16-bit symbols (literal, or marker "byte i of the 32 KB in front of this chunk").
//                   Serial by nature -- Huffman codes have no boundaries anyone wrote down -- so the parallelism is chunks:
//                   a 614-MB .fq.gz is 19 000 chunks of 32 KB, all in flight at once
//   k_gz_window_a   per accepted chunk: the 32 KB behind it when they hold no marker (FASTQ: nearly always) -- else flagged
//   k_gz_window_b   ONE workgroup walks the flagged chunks in stream order (each needs the window in front of it)
//   k_gz_crc        per chunk: CRC-32 of its bytes by 256 slices, combined with GF(2) operators (gz_core.h crc_*)
//   k_gz_translate  symbols -> bytes (markers through the window in front of the chunk) straight into the caller's buffer,
//                   e.g. a block buffer of the FASTQ framer
// Bound: k_gz_decode by memory LATENCY per lane (table look-ups and copies are dependent loads), hence by lanes in flight; the
// other kernels stream (2 B read + 1 B written per byte of output).
#include <hip/hip_runtime.h>

#include "gz_core.h"
#include "gz_device.h"

namespace hast {
namespace gz {

// ---- search: the first position in [from_bit, from_bit + search_to_lo) that parses as a non-final dynamic block header -----
__global__ void __launch_bounds__(64) k_gz_search(ChunkJob *jobs, uint32_t n_jobs, const uint32_t *w, uint64_t nbits, uint32_t *tabs) {
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
    uint32_t *my_tabs = tabs + (size_t)j * kTabWords;
    uint64_t found = ~0ull;
    for (uint64_t base = from; base < to && found == ~0ull; base += 64) {
        const uint64_t bit = base + lane;
        bool c = false;
        if (bit < to) c = candidate(bits_at(w, bit), bits_at(w, bit + 56));
        unsigned long long mask = __ballot(c);
        while (mask) {                                                          // survivors, lowest position first (about 1 in 1500)
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1;
            int ok = 0;
            if ((int)lane == l) ok = header_parses(w, nbits, bit, my_tabs) ? 1 : 0;
            ok = __shfl(ok, l);
            if (ok) {
                found = base + (uint64_t)l;
                break;
            }
        }
    }
    if (lane == 0) {
        job.start_bit = found == ~0ull ? from : found;
        job.end_bit = job.start_bit;
        job.n_out = 0;
        job.err_code = 0;
        job.status = found == ~0ull ? 0u : kStFound;
    }
}

// ---- decode: a lane per chunk ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_gz_decode(ChunkJob *jobs, uint32_t n_jobs, const uint32_t *w, uint64_t nbits, uint32_t *tabs, uint16_t *syms) {
    const uint32_t j = blockIdx.x * 64 + threadIdx.x;
    if (j >= n_jobs) return;
    ChunkJob job = jobs[j];
    if (!(job.status & kStFound)) {
        job.n_out = 0;
        job.end_bit = job.start_bit;
        jobs[j] = job;
        return;
    }
    decode_chunk(job, w, nbits, tabs + (size_t)j * kTabWords, syms + job.sym_off);
    jobs[j] = job;
}

// ---- windows -------------------------------------------------------------------------------------------------------------------
// windows[c] = the kWindow bytes of the stream behind accepted chunk c (window -1 = `carry`, what the batch in front left)
__global__ void __launch_bounds__(256) k_gz_window_a(const AccDev *acc, uint32_t n_acc, uint8_t *windows, uint32_t *need) {
    const uint32_t c = blockIdx.x, tid = threadIdx.x;
    if (c >= n_acc) return;
    const AccDev a = acc[c];
    uint8_t *wdw = windows + (size_t)c * kWindow;
    const uint16_t *s = a.sym;
    if (a.n_out < kWindow && !a.no_history) {                                   // part of the window is the window in front
        if (tid == 0) need[c] = 1;
        return;
    }
    const uint32_t zeros = a.n_out < kWindow ? kWindow - a.n_out : 0u;          // (a member shorter than the window: nothing valid copies from there)
    const uint32_t first = a.n_out - (kWindow - zeros);
    int marker = 0;
    for (uint32_t k = tid; k < kWindow; k += 256) {
        uint32_t v = 0;
        if (k >= zeros) {
            v = s[first + (k - zeros)];
            marker |= v >= kMarker;
        }
        wdw[k] = (uint8_t)v;
    }
    marker = __syncthreads_or(marker);
    if (tid == 0) need[c] = marker ? 1u : 0u;
}
__global__ void __launch_bounds__(1024) k_gz_window_b(const AccDev *acc, uint32_t n_acc, uint8_t *windows, const uint8_t *carry, const uint32_t *need) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t c = 0; c < n_acc; ++c) {
        if (!need[c]) continue;                                                 // (uniform)
        const AccDev a = acc[c];
        const uint8_t *prev = c ? windows + (size_t)(c - 1) * kWindow : carry;
        uint8_t *wdw = windows + (size_t)c * kWindow;
        const uint16_t *s = a.sym;
        const uint32_t own = a.n_out < kWindow ? a.n_out : kWindow;             // symbols of this chunk in its window
        for (uint32_t k = tid; k < kWindow; k += 1024) {
            uint8_t b;
            if (k < kWindow - own) b = prev[k + own];                           // the window in front, shifted
            else {
                const uint32_t v = s[a.n_out - own + (k - (kWindow - own))];
                b = v < kMarker ? (uint8_t)v : prev[v - kMarker];
            }
            wdw[k] = b;
        }
        __threadfence();                                                        // the next flagged chunk reads this window
        __syncthreads();
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
    for (long long i = lo < 0 ? 0 : lo; i < hi; ++i) {
        const uint32_t x = s[i];
        const uint32_t b = x < kMarker ? x : prev[x - kMarker];
        v = s_tab[(v ^ b) & 0xFF] ^ (v >> 8);
    }
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
constexpr uint32_t kTile = 4096;                                                // symbols per workgroup
__global__ void __launch_bounds__(256) k_gz_translate(const AccDev *acc, uint32_t c_first, const uint8_t *windows, const uint8_t *carry,
                                                      uint64_t o_lo, uint64_t o_hi, uint8_t *dst) {
    const uint32_t c = c_first + blockIdx.y;
    const AccDev a = acc[c];
    const uint32_t t0 = blockIdx.x * kTile;
    if (t0 >= a.n_out) return;
    const uint8_t *prev = c ? windows + (size_t)(c - 1) * kWindow : carry;
    const uint16_t *s = a.sym;
    const uint32_t t1 = t0 + kTile < a.n_out ? t0 + kTile : a.n_out;
    for (uint32_t i = t0 + threadIdx.x; i < t1; i += 256) {
        const uint64_t o = a.out_off + i;
        if (o < o_lo || o >= o_hi) continue;
        const uint32_t x = s[i];
        dst[o - o_lo] = x < kMarker ? (uint8_t)x : prev[x - kMarker];
    }
}

// ---- launchers -------------------------------------------------------------------------------------------------------------------
hipError_t launch_search(ChunkJob *d_jobs, uint32_t n, const uint32_t *d_w, uint64_t nbits, uint32_t *d_tabs, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_gz_search, dim3(n), dim3(64), 0, s, d_jobs, n, d_w, nbits, d_tabs);
    return hipGetLastError();
}
hipError_t launch_decode(ChunkJob *d_jobs, uint32_t n, const uint32_t *d_w, uint64_t nbits, uint32_t *d_tabs, uint16_t *d_syms, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_gz_decode, dim3((n + 63) / 64), dim3(64), 0, s, d_jobs, n, d_w, nbits, d_tabs, d_syms);
    return hipGetLastError();
}
hipError_t launch_windows(const AccDev *d_acc, uint32_t n, uint8_t *d_windows, const uint8_t *d_carry, uint32_t *d_need, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_gz_window_a, dim3(n), dim3(256), 0, s, d_acc, n, d_windows, d_need);
    hipLaunchKernelGGL(k_gz_window_b, dim3(1), dim3(1024), 0, s, d_acc, n, d_windows, d_carry, d_need);
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

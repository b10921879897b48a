// fq_device.h -- what fq_kernels.hip and fq_api.cpp share: the device-side state of one block of a FASTQ stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hast {

struct FqState {               // one per buffer slot, in device memory; copied to the host after the framing kernels
    uint64_t parse_lo, parse_hi;   // bytes of the buffer this block's framing covers: previous tail + new bytes
    uint64_t tail_lo;              // first byte after the last complete record: [tail_lo, parse_hi) goes to the next block
    uint64_t tail_in;              // bytes taken over from the previous block
    uint64_t bases;                // sum of the record's base-line lengths
    uint32_t n_nl, n_rec;          // newlines in the parse range, records framed
    uint32_t max_len;              // longest base line
    uint32_t flags;                // 1: a read shorter than K without 'N' (the reference aborts, kmer.h:171); 2: a record larger than the pad
};

hipError_t launch_fq_block(uint8_t *d_buf, FqState *d_st, const uint8_t *d_prev_buf, const FqState *d_prev_st, uint64_t pad, uint64_t n_bytes,
                           uint32_t *d_tile_cnt, uint32_t *d_nl, uint64_t *d_off, uint32_t *d_len, uint32_t *d_bc_pos, uint32_t *d_bc_len,
                           uint32_t *h_bc, uint32_t h_cap, uint32_t k, int last, hipStream_t s);

}  // namespace hast

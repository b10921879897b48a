// gz_device.h -- what gz_api.cpp (host) and gz_kernels.hip (device) share: the device-side view of an accepted chunk and the
// kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gz_core.h"

namespace hast {
namespace gz {

struct AccDev {                  // an accepted chunk, in stream order
    const uint16_t *sym;         // its symbols (device)
    uint64_t out_off;            // offset of its first byte in the inflated stream
    uint32_t n_out;
    uint32_t no_history;         // a member starts with it: no markers, the window in front of it is not part of its own
};

hipError_t launch_search(ChunkJob *d_jobs, uint32_t n, const uint32_t *d_w, uint64_t nbits, hipStream_t s);
// a wave per job; job.sym_off = the device address of the job's symbol buffer / 2; d_syms: the allocation those buffers lie in
// jobs with kJobPoolSlot take their symbol buffer out of a pool of pool_slots slots of job.sym_cap symbols from job.sym_off on (*d_pool_cursor
// = slots taken, zeroed by the caller in front of the launch)
hipError_t launch_decode(ChunkJob *d_jobs, uint32_t n, const uint32_t *d_w, uint64_t nbits, uint16_t *d_syms, uint32_t *d_pool_cursor, uint32_t pool_slots, hipStream_t s);
// d_windows[c] = the 32 KB behind chunk c; d_carry = the 32 KB in front of chunk 0; d_scratch: windows_scratch_bytes(n)
size_t windows_scratch_bytes(uint32_t n);
hipError_t launch_windows(const AccDev *d_acc, uint32_t n, uint8_t *d_windows, const uint8_t *d_carry, void *d_scratch, hipStream_t s);
hipError_t launch_crc(const AccDev *d_acc, uint32_t n, const uint8_t *d_windows, const uint8_t *d_carry, uint32_t *d_crc, hipStream_t s);
// bytes [o_lo, o_hi) of the stream out of chunks c_first .. c_first + n_chunks - 1 (max_syms = the largest n_out among them) -> d_dst[0 ..)
hipError_t launch_translate(const AccDev *d_acc, uint32_t c_first, uint32_t n_chunks, uint32_t max_syms, const uint8_t *d_windows, const uint8_t *d_carry,
                            uint64_t o_lo, uint64_t o_hi, uint8_t *d_dst, hipStream_t s);

}  // namespace gz
}  // namespace hast

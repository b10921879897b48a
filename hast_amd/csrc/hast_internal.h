// hast_internal.h -- what the translation units of libhast.so share besides the public ABI.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/hast.h"

namespace hast {
hast_status set_error(hast_status st, const char *fmt, ...) __attribute__((format(printf, 2, 3)));   // text for hast_last_error()
int default_minimizer_for(int k);                                                                    // honours HAST_MINIMIZER
// reads given as starts + lengths inside one device buffer (FASTQ framed on the GPU, fq_api.cpp): per-read votes only ...
hast_status classify_framed(hast_ctx *c, const uint8_t *d_buf, size_t buf_bytes, const uint64_t *d_off, const uint32_t *d_len, uint32_t max_len,
                            uint32_t *d_votes, size_t n_reads, hipStream_t hs);
// ... and the per-barcode bookkeeping once the host has named the barcodes
hast_status commit_framed(hast_ctx *c, const uint32_t *d_votes, const uint32_t *d_ids, size_t n_reads, hipStream_t hs);
hipStream_t ctx_stream_of(hast_ctx *c);
size_t ctx_n_barcodes(const hast_ctx *c);
// FREES THAT DO NOT STOP THE DEVICE.  hipFree and hipHostFree wait for every stream of the device and hold the runtime's lock while
// they do: a stream that is closed while another one still decodes (the reader threads of `classify`, the upload thread of a .gz stream
// that has sent its last byte) made every HIP call of the process wait 20-50 ms, 0.3 s on some boxes of the pool (tools/hipstall,
// profiles/round5_hipstall_slow_box.txt).  The streams' buffers are PARKED instead and freed for real by hast_release_parked(), by
// hast_ctx_destroy, when more than HAST_PARK_GB (default 32: two .gz streams of `classify` park 21; 0 = free at once, as before)
// are waiting, or when an allocation of a stream fails.
// (site: 0 a .gz stream's upload staging, 1 hast_gz_close, 2 a buffer of a .gz stream that grows, 3 a FASTQ stream's slots;
// HAST_PARK_SITES=<bit mask> parks at those sites only -- how the race that hast_table_clone's device-to-device copy had was found:
// with the device no longer stopped by the frees, `classify --devices 0,0,0` lost hits until the copy was put on the clone's stream)
void park_device(void *p, size_t bytes, int site = 0);
void park_pinned(void *p, size_t bytes, int site = 0);
void release_parked();
// ALLOCATIONS THAT KNOW ABOUT THE PARKED MEMORY.  Up to HAST_PARK_GB of closed streams' buffers stay allocated; whoever then asks for
// more than what is left must not be told "out of memory" (or quietly take a slower path) while tens of GB wait to be freed: every
// device and pinned allocation of the library goes through these two, which on hipErrorOutOfMemory free what is parked and ask once more.
hipError_t dev_malloc(void **p, size_t bytes);
hipError_t pinned_malloc(void **p, size_t bytes, unsigned flags = hipHostMallocDefault);
template <class T> inline hipError_t dev_malloc(T **p, size_t bytes) { return dev_malloc(reinterpret_cast<void **>(p), bytes); }
template <class T> inline hipError_t pinned_malloc(T **p, size_t bytes, unsigned flags = hipHostMallocDefault) { return pinned_malloc(reinterpret_cast<void **>(p), bytes, flags); }
size_t parked_bytes();                                                                               // what is waiting right now
}  // namespace hast

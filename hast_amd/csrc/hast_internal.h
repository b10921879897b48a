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
}  // namespace hast

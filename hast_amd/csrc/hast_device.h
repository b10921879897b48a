// hast_device.h -- host-visible declarations of the device launchers in hast_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "hast_common.h"

namespace hast {

struct TableGeom {
    uint32_t nbuckets;
    int k;          // k-mer length
    int m;          // minimizer length (m == k: plain hashing of the key)
    int wide;       // k == 32: two tag-less tables back to back (hap h at slots + h * nbuckets * 8)
};

struct ClassifyArgs {
    const uint8_t *bases;        // ASCII bases, reads back to back
    uint64_t bases_bytes;        // readable bytes at `bases`
    const uint64_t *offsets;     // n_reads+1 offsets or nullptr (fixed length); with `lens`: n_reads start offsets
    const uint32_t *lens;        // per-row length (rows = segments of long reads) or nullptr
    const uint32_t *seg_read;    // row -> output read index (votes are atomically added) or nullptr
    uint32_t *votes;             // [n_reads][2] or nullptr
    unsigned long long *tile_queue; // zeroed before the launch: next tile index (dynamic load balance)
    const uint64_t *slots;       // table
    uint64_t n_reads;
    const unsigned long long *n_rows_ptr;   // not nullptr: the number of rows is read from here (segment tables built on the device:
                                            // no host round trip for their size); n_reads is then an upper bound for the launch
    uint32_t nbuckets;
    uint32_t read_len;           // fixed length, or upper bound when offsets != nullptr
    uint32_t max_pos;            // read_len-K+1 (0 when read_len<K): position stride per read
    uint32_t w64;                // 64-bit LDS words per read (32 bases each), excluding the pad word
    uint32_t tile_reads;         // reads per workgroup tile
    uint32_t div_magic;          // floor(2^32/max_pos)+1 when exact over the tile's range, else 0
    uint32_t mh_stride;          // m-mer positions per read: read_len-m+1 (0 when read_len<m)
    uint32_t div_mh;             // same trick for mh_stride
    uint32_t div_hw;             // same trick for 2*w64 (16-base half-words per read)
    int k;
    int m;                       // minimizer length
    int wide;                    // K == 32: probe the two per-haplotype tables (slots, slots + nbuckets*8)
    int strict;                  // per-window validity (stage-03 string semantics) instead of the whole-read N skip
    // ---- filter front end (k_classify_f); unused by the exact-table kernel
    const void *filter;          // 4^fg.m blocks of 128 B
    FilterGeom fg;
    uint32_t l1_stride;          // first-level sliding-minimum entries per read (row stride), a multiple of 4
    uint32_t div_l1g;            // exact multiply-high division by l1_stride / 4, or 0
};

hipError_t launch_insert_keys(uint64_t *slots, TableGeom g, const uint64_t *d_keys, size_t n, uint32_t hap,
                              uint32_t *d_err, hipStream_t s);
hipError_t launch_insert_text(uint64_t *slots, TableGeom g, const char *d_text, size_t n_lines,
                              uint32_t hap, int acgt_only, uint32_t *d_err, hipStream_t s);
hipError_t launch_erase_keys(uint64_t *slots, TableGeom g, const uint64_t *d_keys, size_t n, uint8_t *d_hit, hipStream_t s);
hipError_t launch_lookup_keys(uint64_t *slots, TableGeom g, const uint64_t *d_keys, size_t n, uint8_t *d_tags, hipStream_t s);
hipError_t launch_export_slots(const uint64_t *slots, size_t nslots, uint64_t *d_out, size_t cap, unsigned long long *d_counter, hipStream_t s);
hipError_t launch_import_slots(uint64_t *slots, TableGeom g, const uint64_t *d_in, size_t n, uint32_t *d_err, hipStream_t s);
hipError_t launch_count_tags(const uint64_t *slots, TableGeom g, unsigned long long *d_out, hipStream_t s);
hipError_t launch_classify(const ClassifyArgs &a, int grid, size_t smem, hipStream_t s);
// filter front end: builds the filter from the live slots of the exact table; classify through it
hipError_t launch_filter_build(const uint64_t *slots, TableGeom g, void *filter, FilterGeom fg, hipStream_t s);
hipError_t launch_classify_f(const ClassifyArgs &a, int grid, size_t smem, int variants, hipStream_t s);
// measurement: grid x 256 lanes, each group of 8 lanes reads 4 x iters random 128-B blocks of the filter
hipError_t launch_request_ceiling(const void *filter, uint64_t nblocks, uint32_t iters, int grid, uint32_t *d_sink, hipStream_t s);
size_t classify_f_queue_bytes();      // LDS the kernel needs besides the per-read arrays
// d_offsets == nullptr: reads of fixed_len bytes back to back
hipError_t launch_build_segments(const uint64_t *d_offsets, const uint32_t *d_lens, uint64_t fixed_len, size_t n_reads, int k, uint32_t seg_windows, uint64_t *seg_off,
                                 uint32_t *seg_len, uint32_t *seg_read, unsigned long long *d_counter, const uint8_t *d_skip,
                                 uint64_t cap, uint64_t bases_bytes, uint32_t *d_err, hipStream_t s);
hipError_t launch_scan_n(const uint8_t *d_bases, const uint64_t *d_offsets, const uint32_t *d_lens, uint64_t fixed_len, size_t n_reads, uint8_t *d_has_n, hipStream_t s);
hipError_t launch_commit_votes(const uint32_t *d_votes, const uint32_t *d_barcode_ids, unsigned long long *d_counts, uint32_t *d_votes_out,
                               size_t n_reads, hipStream_t s);
// the same bookkeeping without one atomic per read (large batches over many barcodes): pairs partitioned by barcode range in
// LDS, bins summed in LDS, plain read-modify-writes of the counters; max_votes = the most votes a read can have (<= 255)
bool commit_partition_usable(size_t n_reads, size_t n_barcodes, uint32_t max_votes, bool forced);
size_t commit_partition_scratch_bytes(size_t n_reads, size_t n_barcodes, uint32_t *n_bins_out, uint32_t *cap_out);
hipError_t launch_commit_partitioned(const uint32_t *d_votes, const uint32_t *d_barcode_ids, unsigned long long *d_counts, size_t n_barcodes, size_t n_reads,
                                     void *d_scratch, hipStream_t s);
hipError_t launch_add_u64(unsigned long long *d_dst, const unsigned long long *d_src, size_t n, hipStream_t s);          // dst[i] += src[i]
// counts[n][4] -> packed = c0[n] | c1[n] | neg[n] and back (the reserved word stays where it is)
hipError_t launch_counts_pack(const unsigned long long *d_counts, unsigned long long *d_packed, size_t n, hipStream_t s);
hipError_t launch_counts_unpack(unsigned long long *d_counts, const unsigned long long *d_packed, size_t n, hipStream_t s);
hipError_t launch_synth_keys(const SynthParams &p, int hap, uint64_t first, size_t n, uint64_t *d_out, hipStream_t s);
hipError_t launch_synth_reads(const SynthParams &p, uint64_t first, size_t n, uint8_t *d_bases, uint32_t *d_bc, hipStream_t s);

}  // namespace hast

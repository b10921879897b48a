// kc_device.h -- launch interface of kc_kernels.hip (stage 00 k-mer counting) for kc_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "kc_common.h"

namespace hast {

struct KcCountArgs {
    const uint8_t *bytes;            // sequence bytes; any byte outside ACGTacgt separates runs of bases
    size_t n_bytes;                  // readable bytes
    size_t n_starts;                 // windows start at positions [0, n_starts)
    unsigned long long *table;       // nbuckets x 128 B
    uint32_t nbuckets;
    int k, m;
    uint32_t parent;                 // 0 paternal, 1 maternal
    uint32_t slice, n_slices;        // only windows whose minimizer falls into this slice of the key space are counted
    uint32_t tile_bases;             // window starts per tile (multiple of 32)
    unsigned long long *tile_queue;  // zeroed before the launch
    unsigned long long *total;       // [2] windows counted per parent
    uint32_t *err;                   // bit 0: table full
    // partitioned path (rec_out != nullptr): the windows are written out as records instead (kc_kernels.hip, "records")
    unsigned long long *rec_out;
    unsigned long long rec_cap;
    unsigned long long *rec_cursor;  // records written so far (may run past rec_cap: those were counted on the spot)
    uint32_t fine_shift;             // log2(buckets of a slice): where a record that finds no room in the buffer is counted (kc_common.h PLACEMENT)
    uint32_t rec_run_max, rec_off_bits;   // kc_run_max / kc_rec_off_bits of (k, m)
    uint32_t rec_chunk;              // records a workgroup reserves at a time (>= tile_bases / 2); unused ends are filled with null records
    // a table that has not been written yet (`fresh`: its first flush writes every slice without reading it, KcFlushArgs) cannot count a
    // record on the spot: what finds no room in the record buffer goes to the spill list, which the flush applies last
    uint32_t fresh;
    unsigned long long *spill, spill_cap, *spill_n;
};

// one flush of the partitioned path: records -> level-1 regions -> fine regions (in the flat buffer's place) -> slices in LDS -> spill
struct KcFlushArgs {
    unsigned long long *table;
    uint32_t nbuckets;
    int k, m;
    uint32_t fine_shift, n_fine, n_l1, f2;
    unsigned long long *records;     // flat buffer: n_records records in; then n_fine regions of fine_cap records
    unsigned long long n_records;
    const unsigned long long *rec_cursor;   // device word behind n_records (>= n_records)
    int small_flush;                 // 0: few records for the table's size go through the atomic path where they lie; 1: never; 2: always
    uint32_t fresh;                  // the table holds nothing (and has not been cleared): k_kc_apply starts every slice empty instead of reading it,
                                     // and writes slices without records too.  Never with a flush that kc_flush_is_small() (it skips the sweep)
    unsigned long long *l1_recs;     // n_l1 x l1_split regions of l1_cap records: region s * n_l1 + b = level-1 bin b as the workgroups with blockIdx % l1_split == s wrote it
    uint32_t l1_split;
    uint32_t l1_cap, fine_cap;
    uint32_t *l1_fill, *l1_valid, *fine_fill, *fine_valid;
    unsigned long long *spill;       // spill_cap records
    unsigned long long spill_cap;
    unsigned long long *spill_n;     // zeroed by the caller before the first record of a flush period is emitted
    uint32_t *err;
};
constexpr uint32_t kKcL1FillWords = 1;    // words between two level-1 fill / valid counters (a line apart, 32, was slower: the adds of a wave
                                          // to neighbouring counters leave the CU merged per 32-B sector)
hipError_t launch_kc_flush(const KcFlushArgs &a, hipStream_t s);
size_t kc_count_smem(uint32_t tile_bases, int k, int m);
hipError_t launch_kc_count(const KcCountArgs &a, unsigned grid, hipStream_t s);
bool kc_flush_is_small(const KcFlushArgs &a);
hipError_t launch_kc_clear(unsigned long long *table, size_t nbuckets, hipStream_t s);
hipError_t launch_kc_stats(const unsigned long long *table, size_t nbuckets, unsigned long long *d_out3, hipStream_t s);
hipError_t launch_kc_histo(const unsigned long long *table, size_t nbuckets, uint32_t parent, unsigned long long *d_out, hipStream_t s);
hipError_t launch_kc_select(const unsigned long long *table, size_t nbuckets, uint32_t parent, uint32_t lower, uint32_t upper, int k,
                            unsigned long long *d_out, size_t cap, unsigned long long *d_cursor, hipStream_t s);
hipError_t launch_kc_format(const unsigned long long *d_keys, size_t n, int k, char *d_text, hipStream_t s);
hipError_t launch_kc_to_table_keys(const unsigned long long *d_keys, size_t n, int k, unsigned long long *d_out, hipStream_t s);
hipError_t launch_kc_synth(const KcSynth &g, int parent, uint64_t first_read, size_t n_bytes, uint8_t *d_out, hipStream_t s);
hipError_t kc_sort_keys(void *d_tmp, size_t *tmp_bytes, unsigned long long *d_in, unsigned long long *d_out, size_t n, int k, hipStream_t s);

}  // namespace hast

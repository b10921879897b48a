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
    // striped streams (the blocks of one file go to several GPUs): the view [parse_lo, parse_hi) = this block's own bytes
    // followed by the first bytes of the next block; records that START before own_hi are this block's
    uint64_t own_hi;
    uint32_t phase;                // newlines of the file in front of parse_lo, mod 4 (the line of a record parse_lo lies in)
    uint32_t bol;                  // a line starts at parse_lo (the byte in front of it is a newline, or the file starts here)
    uint32_t eof;                  // the file ends at parse_hi
    uint32_t reserved;
};

// Device-side cache barcode text -> dense id (the authority stays the host's dictionary, which is one per job): 32-byte
// entries, open addressing.  key = the 16-byte text record the framer makes (length byte + up to 15 bytes).
struct NameEntry {
    uint32_t key[4];
    uint32_t id;
    uint32_t state;            // 0 empty, 1 being written, 2 ready
    uint32_t pad[2];
};
struct NamePub {               // what the host publishes after naming a record the cache did not know
    uint32_t key[4];
    uint32_t id;
    uint32_t pad[3];
};
constexpr uint32_t kNameUnknown = 0xFFFFFFFFu;

// ids of the first n records from the cache: d_ids / h_ids get the id or kNameUnknown, h_unknown = [count, index, index, ...]
hipError_t launch_fq_name(const uint32_t *d_text, uint32_t n, const NameEntry *tab, uint32_t mask, uint32_t *h_ids, uint32_t *h_unknown, hipStream_t s);
hipError_t launch_names_insert(const NamePub *pubs, uint32_t n, NameEntry *tab, uint32_t mask, hipStream_t s);
// the dictionary variant: unknown texts get the next id from *d_n_ids (while < limit) and are filed under it in d_text_of_id[id] (16 B each)
// (d_ids: the same ids in device memory as well, or NULL)
hipError_t launch_fq_name_claim(const uint32_t *d_text, uint32_t n, NameEntry *tab, uint32_t mask, uint32_t *d_n_ids, uint32_t limit, void *d_text_of_id,
                                uint32_t *h_ids, uint32_t *h_unknown, uint32_t *d_ids, hipStream_t s);
// ... launched behind a block's framing kernels on their stream: n = min(d_st->n_rec, cap), read on the device
hipError_t launch_fq_name_claim_framed(const uint32_t *d_text, const FqState *d_st, uint32_t cap, NameEntry *tab, uint32_t mask, uint32_t *d_n_ids, uint32_t limit,
                                       void *d_text_of_id, uint32_t *h_ids, uint32_t *h_unknown, uint32_t *d_ids, hipStream_t s);

hipError_t launch_fq_block(uint8_t *d_buf, FqState *d_st, const uint8_t *d_prev_buf, const FqState *d_prev_st, uint64_t pad, uint64_t n_bytes,
                           uint32_t *d_tile_cnt, uint32_t *d_nl, uint64_t *d_off, uint32_t *d_len, uint32_t *d_bc_pos, uint32_t *d_bc_len,
                           uint32_t *h_bc, uint32_t *d_text, uint32_t h_cap, uint32_t k, int last, hipStream_t s);

// counters renumbered (dictionaries of several GPUs merged by text): record i < n_perm of d_src added to record d_perm[i] of d_dst, the
// records from n_perm to n_old keep their places
hipError_t launch_counts_permute(unsigned long long *d_dst, const unsigned long long *d_src, const uint32_t *d_perm, size_t n_perm, size_t n_old, size_t n_new, hipStream_t s);

// ---- routing (the wrapper's steps 10-11: every record to the file of its barcode's class, quartering_fastq.awk) ----------------
struct RouteState {            // one per buffer slot, device + a pinned host copy behind the routing kernels
    uint64_t bytes[4];         // bytes of this block's records per class: 0 nobarcode, 1 paternal, 2 maternal, 3 homozygous
    uint64_t tail_lo, tail_hi; // the partial record at the end of the FILE (fewer than four newlines): the host routes it, awk's rules
    uint32_t count[4];         // records per class
    uint32_t n_cand;           // record slots the kernels walked (records, and candidates of a striped block that are not its own)
    uint32_t flags;            // 1: a record only the host can route (field longer than 15 bytes, or a barcode no list holds: the ERROR
                               //    line needs its text); 2: a record that does not end inside the view
    uint32_t reserved[2];
};
constexpr uint8_t kRouteUnclassified = 0xFE, kRouteHost = 0xFF, kRouteSkip = 0xFD;
constexpr int kRouteTile = 256;                  // records per tile of the per-class prefix sums
// class + extent of every record of a framed block (d_nl, FqState of the framing kernels), the per-class exclusive prefix sums of the
// record lengths, and the records copied whole, in input order, into d_out = [class 0 | class 1 | class 2 | class 3]
hipError_t launch_fq_route(const uint8_t *d_buf, const FqState *d_st, const uint32_t *d_nl, int striped, int last, const NameEntry *tab, uint32_t mask,
                           uint32_t *d_rstart, uint32_t *d_rlen, uint8_t *d_rcls, uint32_t *d_rtile, uint32_t max_rec, RouteState *d_rs, uint8_t *d_out,
                           hipStream_t s);

// ---- striped streams: a block is framed on its own, from the number of newlines in front of it ----------------------------
// newlines of the block's own bytes [pad, pad + n_bytes) -> d_st->n_nl (the host adds them up over the blocks of the file)
hipError_t launch_fq_count_own(const uint8_t *d_buf, FqState *d_st, uint64_t pad, uint64_t n_bytes, uint32_t *d_tile_cnt, hipStream_t s);
// frames the records that start in [pad, pad + n_bytes); their header and base lines may reach into the n_over bytes behind
hipError_t launch_fq_block_striped(const uint8_t *d_buf, FqState *d_st, uint64_t pad, uint64_t n_bytes, uint64_t n_over, uint32_t phase, int bol, int eof,
                                   uint32_t *d_tile_cnt, uint32_t *d_nl, uint64_t *d_off, uint32_t *d_len, uint32_t *d_bc_pos, uint32_t *d_bc_len,
                                   uint32_t *h_bc, uint32_t *d_text, uint32_t h_cap, uint32_t k, hipStream_t s);

}  // namespace hast

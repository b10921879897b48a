/*
 * hast.h -- C ABI of libhast.so, the MI355X-native (gfx950) implementation of HAST's stage-01
 * read-classification hot path.
 *
 * What it replaces.  The reference has no FFI for this path: it is a process
 * (`01.classify_stlfr_reads/classify`, argv in / TSV on stdout, classify.cpp:373-450).  The
 * drop-in boundary is therefore the `classify` executable built from hast_amd/csrc/classify_main.cpp,
 * and THIS header is the seam between that host program (and bench.py / pytest via ctypes) and
 * the HIP device code.  Each entry point cites the reference code whose work it takes over
 * (paths relative to /root/reference/01.classify_stlfr_reads/).
 *
 * Conventions
 *   - plain C types only; every call returns a hast_status and never throws across the ABI;
 *     hast_last_error() gives a thread-local message for the last failing call.
 *   - the caller owns host buffers; pointers named d_* are DEVICE pointers (from hast_dev_alloc
 *     or from any other allocator on the same device, e.g. a torch tensor's data_ptr()).
 *   - one context per device; calls on one context are serialised by the caller.
 *   - hast_stream is a hipStream_t passed as void* (NULL = the context's own stream).  A context has ONE set of launch
 *     scratch (tile queue, segment table, vote scratch): at most one classification of a context may be in flight, i.e. all
 *     classify calls of a context must go to the same stream (any stream, but one).
 *   - there is NO CPU fallback: with no usable GPU hast_ctx_create fails with HAST_ERR_NO_DEVICE.
 */
#ifndef HAST_H
#define HAST_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    HAST_OK = 0,
    HAST_ERR_INVALID = 1,    /* bad argument / K out of [1,32] / wrong call order            */
    HAST_ERR_NO_DEVICE = 2,  /* no HIP device, or the device is not usable                  */
    HAST_ERR_HIP = 3,        /* a HIP runtime call failed (message in hast_last_error)      */
    HAST_ERR_OOM = 4,        /* host or device allocation failed                            */
    HAST_ERR_TABLE_FULL = 5, /* more distinct keys than hast_table_reserve planned for      */
    HAST_ERR_FORMAT = 6,     /* k-mer text is not fixed-width K-byte lines (kmer.h:154)     */
    HAST_ERR_RCCL = 7,       /* the RCCL all-reduce failed                                  */
    HAST_ERR_IO = 8,         /* file could not be opened / read / is damaged                */
    HAST_ERR_UNSUPPORTED = 9 /* this entry cannot take this input; another one can (e.g. hast_gz_open: inflate on the host) */
} hast_status;

typedef struct hast_ctx hast_ctx;
typedef void *hast_stream;

const char *hast_version(void);
const char *hast_last_error(void);

/* ---- context ------------------------------------------------------------------------------
 * Holds what the reference keeps in process globals: g_K (classify.cpp:29), g_kmers[2]
 * (classify.cpp:27) as ONE merged open-addressed table in HBM, and the per-barcode counters
 * (BarcodeCache, classify.cpp:50-64).  k in [1,32] (the reference's correct range; at k = 32 a key
 * fills the 64-bit slot, so the table is two tag-less per-haplotype tables and every window is probed twice). */
hast_status hast_ctx_create(int device_ordinal, int k, hast_ctx **out);
void        hast_ctx_destroy(hast_ctx *);
/* Streams that are closed (hast_fq_destroy, hast_gz_close) PARK their device and pinned buffers instead of freeing them: a free waits
 * for every stream of the device, i.e. stalls the streams that are still at work.  The parked memory is freed for real here, by
 * hast_ctx_destroy, when more than HAST_PARK_GB (environment, default 32; 0 = free at once) are waiting, or when a stream's allocation fails.  No reference
 * counterpart (the reference's streams are host objects). */
void        hast_release_parked(void);
int         hast_ctx_k(const hast_ctx *);
/* Tuning: length m of the minimizer that places a key's bucket (see hast_common.h).  Defaults to
 * K for K<=16, else max(16, K-8).  May only be changed before hast_table_reserve. */
int         hast_ctx_minimizer(const hast_ctx *);
hast_status hast_ctx_set_minimizer(hast_ctx *, int m);
int         hast_ctx_device(const hast_ctx *);
/* Measurement switches.  The environment is read ONCE, when a context is created (HAST_CLASSIFY, HAST_FILTER_*, HAST_COMMIT,
 * HAST_F_GEO, HAST_F_RL, HAST_TILE_LDS, HAST_MINIMIZER); results never depend on them.  hast_ctx_set_option changes one on a live
 * context: "commit" 0 = by batch size / 1 = one atomic per read / 2 = partitioned; "kernel_geo", "kernel_rl" 0 = the generic
 * k_classify_f instantiations instead of the ones with the BASELINE geometry / row length compiled in; "tile_lds" bytes (0 =
 * default).  hast_ctx_options writes the switches that differ from their defaults as "name=value ..." ("" = none). */
hast_status hast_ctx_set_option(hast_ctx *, const char *name, long value);
hast_status hast_ctx_options(const hast_ctx *, char *out, size_t cap);
hast_stream hast_ctx_stream(const hast_ctx *);
hast_status hast_stream_sync(hast_ctx *, hast_stream);

/* ---- raw device memory (for callers without another allocator, e.g. ctypes tests) ---------- */
hast_status hast_dev_alloc(hast_ctx *, size_t bytes, void **d_out);
hast_status hast_dev_free(hast_ctx *, void *d_ptr);
/* free and total memory of the context's device, and the bytes closed streams have parked (hast_release_parked): what `classify --stats`
 * samples for its HBM head-room line.  Any of the three may be NULL. */
hast_status hast_dev_mem_info(hast_ctx *, size_t *free_bytes, size_t *total_bytes, size_t *parked_bytes);
hast_status hast_memcpy_h2d(hast_ctx *, void *d_dst, const void *src, size_t bytes);
hast_status hast_memcpy_d2h(hast_ctx *, void *dst, const void *d_src, size_t bytes);
hast_status hast_memset_d(hast_ctx *, void *d_dst, int byte, size_t bytes, hast_stream);

/* ---- the k-mer table: g_kmers[0], g_kmers[1] (classify.cpp:27) -----------------------------
 * One table, slot = (canonical_key << 2) | tags, tag bit h set <=> key in haplotype h's set.
 * Buckets of 8 slots = 64 B; home bucket from the hash of the key's minimizer (consecutive windows of
 * a read then mostly share a bucket line); a key whose home bucket is full goes to a bucket chosen by the key's own
 * hash and walks on from there (hast_common.h: overflow_bucket / next_bucket). */

/* Size the table for up to `max_keys` distinct keys (both haplotypes together) at the given
 * load factor (0 => 0.2: 288 GB of HBM make a sparse table free, and full buckets rare).  Discards
 * any previous table. */
hast_status hast_table_reserve(hast_ctx *, uint64_t max_keys, double load_factor);

/* load_kmers (classify.cpp:30-46) on an in-memory image of the file: '\n'-separated K-byte lines,
 * parsed, canonicalised (Kmer::str2Kmer, kmer.h:153-166) and inserted ON THE DEVICE.  A final piece
 * without '\n' is dropped (classify.cpp:41).  *lines_out = number of lines used ("Recorded N").
 * HAST_ERR_FORMAT if any line is not exactly K bytes (the reference asserts, kmer.h:154). */
hast_status hast_table_insert_text(hast_ctx *, int hap, const char *text, size_t nbytes,
                                   uint64_t *lines_out);
/* The same straight from a regular FILE, streamed (several threads pread the next piece into pinned memory while the
 * previous one is inserted): 2 x 4.4 GB of 21-mer text load in well under a second instead of at one thread's read rate.
 * HAST_ERR_IO if the path cannot be opened or is not a regular file (a pipe: read it and use hast_table_insert_text). */
hast_status hast_table_insert_text_file(hast_ctx *, int hap, const char *path, uint64_t *lines_out);
/* acgt_only != 0: both calls above also fail with HAST_ERR_FORMAT when a line holds a byte other than upper-case A/C/G/T
 * (checked on the device).  Stage 01 accepts any byte (kmer.h:11 codes them all); the stage-03 classifier compares k-mers
 * as case-sensitive strings (S03/src_main/classify.cpp:59-65), which integer keys reproduce only for such lines. */
hast_status hast_ctx_set_text_check(hast_ctx *, int acgt_only);
/* Same, keys already canonical 2K-bit values (host / device resident). */
hast_status hast_table_insert_keys(hast_ctx *, int hap, const uint64_t *canon_keys, size_t n);
hast_status hast_table_insert_keys_device(hast_ctx *, int hap, const uint64_t *d_canon_keys, size_t n,
                                          hast_stream);

/* InitAdaptor (classify.cpp:314-339): remove each key from BOTH sets.  out_hit[i] (optional) gets
 * bit h set when key i was present in haplotype h's set (the reference logs one INFO line each). */
hast_status hast_table_erase(hast_ctx *, const uint64_t *canon_keys, size_t n, uint8_t *out_hit);

/* g_kmers[h].size() as used by getHap (classify.cpp:70-71): distinct canonical keys per haplotype
 * after any erase. */
hast_status hast_table_sizes(hast_ctx *, uint64_t *n_hap0, uint64_t *n_hap1);

/* Membership of host keys (test/diagnostic entry): out_tags[i] = tag bits of key i (0 = absent). */
hast_status hast_table_lookup(hast_ctx *, const uint64_t *canon_keys, size_t n, uint8_t *out_tags);

/* Binary key-set cache (so the multi-GB k-mer text files are parsed once): hast_table_save writes the live keys
 * with their tag bits (8 B per distinct key, sorted; both sets, after any erase) to `path`; hast_table_load sizes a
 * new table for them (load factor as in hast_table_reserve) and re-inserts them.  K must match the context. */
hast_status hast_table_save(hast_ctx *, const char *path);
hast_status hast_table_load(hast_ctx *, const char *path, double load_factor);
hast_status hast_table_file_info(const char *path, int *k_out, uint64_t *n_keys_out);   /* no GPU needed */

/* Replicate src's finished table (and its filter) onto dst's device with one peer copy (xGMI), instead of building it
 * once per GPU: what the multi-GPU classify CLI does after building the table on its first device.  Same K required;
 * dst's previous table is discarded. */
hast_status hast_table_clone(hast_ctx *dst, hast_ctx *src);

/* geometry, for roofline accounting */
hast_status hast_table_info(const hast_ctx *, uint64_t *n_buckets, uint64_t *bytes);

/* ---- the fingerprint filter in front of the table ----------------------------------------------
 * What hast_classify_* actually probes (hast_amd/csrc/hast_common.h): 16-bit prints of every key,
 * in 128-B blocks named by a forward mod-minimizer of the key's string and of its reverse complement, so that the
 * consecutive windows of a read ask for few blocks.  A window the filter cannot rule out is looked up in the table
 * above, which alone decides hits, so results never depend on the filter.  It is (re)built from the table's live keys
 * by the first classification after keys were added, or explicitly by hast_filter_build (e.g. outside a timed region).
 * The geometry follows the key count: 4^m blocks with m up to 15 (137 GB, chosen above 537M keys; 4^14 = 34 GB below that and
 * when HBM has no room), a print filed in the less loaded of two sub-buckets where blocks are crowded, in one where they hold
 * <= 2.2 strings on average (the probe then loads and compares one sub-bucket per window).
 * Where a filed string fits a 16-bit entry EXACTLY (its block is a bijection of the sampled m-mer, the entry holds the rest:
 * 2(K-m) + log2(W) <= 17 bits, e.g. K = 21 with m = 14) the entries are exact codes + the key's tag bits instead of prints,
 * and a match in the filter IS the hit: only windows that land in a full sub-bucket still ask the table.  Same results.
 * hast_ctx_set_filter: enable = 0 probes the table directly (the round-1 kernel; also HAST_CLASSIFY=exact in the
 * environment), 1 = filter, exact entries where they fit, 2 = filter with prints always (also HAST_FILTER_EXACT=0); m (sampled
 * m-mer, 4^m blocks), t (ordering t-mer), kp (bases of a window the sampling looks at) = 0 picks the geometry from K and
 * the key count.  hast_filter_info: *enabled = 0 off, 1 prints, 2 exact entries (known once the filter is built). */
hast_status hast_ctx_set_filter(hast_ctx *, int enable, int m, int t, int kp);
hast_status hast_filter_build(hast_ctx *);
hast_status hast_filter_info(const hast_ctx *, int *enabled, int *m, int *t, int *kp, uint64_t *bytes);
/* Measurement entry (bench.py's roofline): requests per second at which this GPU serves uniformly random 128-B blocks of the
 * context's own filter, read in the probe kernel's access shape with no arithmetic -- the request-rate ceiling of this box
 * over this footprint, against which k_classify_f's own request rate is priced in the same run. */
hast_status hast_filter_request_ceiling(hast_ctx *, double *requests_per_s);

/* ---- per-barcode counters: BarcodeCache (classify.cpp:50-64) -------------------------------
 * Device layout: uint64 counts[n_barcodes][4] = { c0, c1, neg, reserved } (32-byte records: a record is one sector for the commit
 * kernels' updates; what leaves the GPU -- the all-reduce, the copy to the host -- is the three live words, see hast_counts_pack):
 *   c0/c1 = sum of per-read votes for key 0/1 (classify.cpp:203-206), neg = key -1 (:191,:207-208).
 * A barcode was "seen" (gets an output row, classify.cpp:94) iff c0|c1|neg != 0.
 * 64-bit accumulation on the device, narrowed by whoever prints ("Hot barcode"): the reference counts in `int`
 * (classify.cpp:51), which the no-barcode bucket "0_0_0" -- 10-20 % of real stLFR reads -- overflows on large runs (undefined
 * there).  Here no counter wraps and none carries into its neighbour, whichever commit path a batch takes: one atomic per
 * counter and read (k_commit_votes), or, for large device-resident batches over many barcodes, pairs grouped by barcode range and
 * summed in LDS, then plain adds by the one workgroup that owns the range (hast_ctx_set_option "commit" / HAST_COMMIT=
 * atomic|partition at context creation force either; same sums).  The classify program prints the exact count and warns when it
 * is above INT_MAX.
 * Exclusivity: while a classification of a context is in flight nothing else may update the counters it is bound to -- the
 * partitioned path adds with plain read-modify-writes.  Two contexts (or two streams) may share a caller-owned buffer only if
 * their classify calls do not overlap in time. */
hast_status hast_counts_resize(hast_ctx *, size_t n_barcodes);                 /* library-owned, zeroed */
hast_status hast_counts_bind(hast_ctx *, uint64_t *d_counts, size_t n_barcodes); /* caller-owned buffer, 32-byte aligned */
hast_status hast_counts_zero(hast_ctx *, hast_stream);
hast_status hast_counts_read(hast_ctx *, uint64_t *c0, uint64_t *c1, uint64_t *neg, size_t n_barcodes);
/* the same for records [first, first + n): a job whose ids come from two numberings (a device dictionary from 0, the host's above
 * hast_names_limit) reads the two ranges that exist */
hast_status hast_counts_read_range(hast_ctx *, size_t first, size_t n, uint64_t *c0, uint64_t *c1, uint64_t *neg);
/* The three live words of the first n_barcodes records as three arrays, d_packed = c0[n] | c1[n] | neg[n] (3 x n_barcodes u64, device),
 * and back (the records' three words are overwritten, the reserved one is left alone): what a caller that runs its own collective --
 * bench.py: one torch.distributed all_reduce over RCCL -- moves is 24 bytes per barcode, not the 32 of the padded records;
 * hast_counts_read and hast_counts_allreduce do the same inside.  Asynchronous on `stream`. */
hast_status hast_counts_pack(hast_ctx *, uint64_t *d_packed, size_t n_barcodes, hast_stream);
hast_status hast_counts_unpack(hast_ctx *, const uint64_t *d_packed, size_t n_barcodes, hast_stream);
/* The bookkeeping alone (classify.cpp:203-208) for per-read votes the caller holds: d_votes[n_reads][2] = (vote0, vote1), 8-byte
 * aligned; d_barcode_ids[n_reads] < n_barcodes; max_votes = an upper bound on any single vote (reads of up to 255 windows may take
 * the partitioned path).  Asynchronous on `stream`. */
hast_status hast_counts_add_votes(hast_ctx *, const uint32_t *d_votes, const uint32_t *d_barcode_ids, size_t n_reads,
                                  uint32_t max_votes, hast_stream);
/* Thread-merge of the reference (collectBarcodes/BarcodeCache::Add, classify.cpp:57-63,226-229)
 * across the GPUs of ONE process: a single in-place RCCL all-reduce(sum,u64) over the counters of
 * n_ctx contexts (same n_barcodes).  Contexts that share a device (a logical split: --devices 0,0 or 0,0,1,1) are summed by a kernel
 * on that device into its first context first; the all-reduce then runs over one context per distinct device (none when there is
 * only one), and the totals are handed back to every context.  The communicators of a device list
 * are created by the first call that names it and kept until the process ends (later calls only enqueue the all-reduce). */
hast_status hast_counts_allreduce(hast_ctx *const *ctxs, int n_ctx);
/* New counters of n_new records (zeroed): record i < n_perm of the old ones is ADDED to record perm[i] (perm: host array; several old
 * records may name one new record; 0xFFFFFFFF: dropped), the old records from n_perm on keep their places.  How contexts whose
 * dictionaries numbered the barcodes differently -- one device dictionary per GPU -- are brought into one id space (hast_names_merge
 * gives perm) before hast_counts_allreduce sums them. */
hast_status hast_counts_permute(hast_ctx *, const uint32_t *perm, size_t n_perm, size_t n_new);

/* ---- classification: MultiThread::process_reads (classify.cpp:186-209) ---------------------
 * For each read: whole-read skip when it holds an upper-case 'N' (containN, :182-185,190-193);
 * otherwise every window of K bases is 2-bit coded ((c&6)>>1, kmer.h:11), canonicalised
 * (kmer.h:169-194) and looked up in both sets (:195-202); votes go to the read's barcode (:203-208).
 *
 *   d_bases        ASCII bases of all reads, back to back (device)
 *   d_offsets      n_reads+1 byte offsets into d_bases, or NULL => fixed length: read i at i*read_len
 *   read_len       fixed read length (d_offsets==NULL) or an upper bound on every read's length
 *   d_barcode_ids  per-read dense barcode id (< n_barcodes), or NULL => counters untouched
 *   d_votes        optional [n_reads][2] per-read (vote0, vote1) output (per-read mode), 8-byte aligned, or NULL
 *   bases_bytes    total bytes readable at d_bases.  The kernel loads whole aligned dwords: up to 3 bytes in front of
 *                  d_bases (when it is not 4-byte aligned) and up to 3 bytes behind d_bases + bases_bytes are loaded and
 *                  ignored; they must be mapped device memory, which holds for any pointer into an allocation made by
 *                  hipMalloc or a sub-allocator with >= 4-byte granules (torch's: 512 B).
 * A read shorter than K has no windows (the reference aborts, kmer.h:171): it votes 0/0.
 * Reads of any length: with read_len > 4096 (with or without d_offsets) the reads are cut into segments on the device (same
 * result; the segment count stays on the device, nothing waits for the host).
 * Asynchronous on `stream`.  Per call: the probes (k_classify_f) write per-read votes -- to d_votes, or to library scratch that
 * grows on demand (a larger batch than ever before synchronises once) -- and a second step turns them into the per-barcode
 * counters: one atomic per read (k_commit_votes), or, for batches of >= 2M reads over many barcodes, the pairs grouped by barcode
 * range and summed in LDS (k_commit_partition + k_commit_bins) -- DESIGN.md section 3: updates interleaved with the probes' reads
 * cost four times as much, hence kernels of their own. */
hast_status hast_classify_device(hast_ctx *, const uint8_t *d_bases, size_t bases_bytes,
                                 const uint64_t *d_offsets, uint32_t read_len,
                                 const uint32_t *d_barcode_ids, uint32_t *d_votes,
                                 size_t n_reads, hast_stream);
/* Kernel timing for measurements (bench.py's roofline): with n_slots > 0 every hast_classify_device call records HIP
 * events around its two kernels (k_classify, k_commit_votes) on the launch stream, for the last n_slots calls;
 * hast_classify_times waits for them and returns the durations in call order (ms), oldest first, and forgets them. */
hast_status hast_classify_timing(hast_ctx *, int n_slots);
hast_status hast_classify_times(hast_ctx *, float *classify_ms, float *commit_ms, int max, int *n_out);
/* Same with HOST buffers: staged to the device through the context's pinned double buffer
 * (what the classify CLI uses).  Returns once the batch is enqueued; buffers may be reused on return. */
hast_status hast_classify_batch(hast_ctx *, const uint8_t *bases, const uint64_t *offsets,
                                const uint32_t *barcode_ids, size_t n_reads, uint32_t max_read_len);

/* Per-read mode (BASELINE config 5; reference: the per-read classifier of stage 03,
 * 03.mkoutput_by_fabulous2.0/src_main/classify.cpp:203-218): no barcodes, no whole-read N skip; a window
 * counts for haplotype h when its K bytes are upper-case A/C/G/T and its canonical k-mer is in set h (== the
 * reference's string lookup of the window against every k-mer line and its reverse complement, given
 * upper-case ACGT k-mer lines).  Reads may have any length (20 kb PacBio-style reads are cut into segments on
 * the device).  d_votes[n_reads][2] is overwritten with (hits0, hits1). */
hast_status hast_classify_perread_device(hast_ctx *, const uint8_t *d_bases, size_t bases_bytes,
                                         const uint64_t *d_offsets, size_t n_reads, uint32_t *d_votes, hast_stream);
/* Same with host buffers, synchronous (H2D, classify, D2H of the votes). */
hast_status hast_classify_perread(hast_ctx *, const uint8_t *bases, const uint64_t *offsets, size_t n_reads,
                                  uint32_t *votes_out);

/* Zero-copy variant for host-fed callers (the CLI's parser threads write straight into pinned memory):
 * hast_batch_begin hands out the pinned staging arrays of the next batch (waiting for that buffer's previous
 * batch to finish on the GPU); the caller fills bases back to back, offsets[0..n_reads] and barcode ids, then
 * hast_batch_submit enqueues H2D + classification.  Capacities: bases_capacity bytes, reads_capacity reads. */
hast_status hast_batch_begin(hast_ctx *, size_t bases_capacity, size_t reads_capacity,
                             uint8_t **bases, uint64_t **offsets, uint32_t **barcode_ids);
hast_status hast_batch_submit(hast_ctx *, size_t n_reads, uint32_t max_read_len);

/* ---- FASTQ framing on the GPU: processFastq (classify.cpp:238-278) --------------------------------
 * The reference's producer thread reads a record as four getlines (no '@'/'+' validation, no CR stripping), takes the
 * barcode out of line 1 (parseName, :112-119) and hands line 2 to a worker.  hast_fq does that framing for whole blocks of
 * raw file bytes on the device: the caller only moves bytes (pread / inflate straight into the pinned buffer it is given)
 * and names barcodes; the reads are classified where they lie in the raw block.  Per block, in this order:
 *     hast_fq_acquire   pinned host memory for the next block_bytes of the file (waits until that buffer is free again)
 *     hast_fq_submit    n_bytes are in it; last != 0: the file ends here (the EOF rules of classify.cpp:257-268 apply:
 *                       a final record counts when its header line is terminated).  Enqueues the copy and the framing.
 *     hast_fq_next      oldest submitted block: waits for its record table (hast_fq_poll tells whether it is there), starts the classification of its reads, and
 *                       gives the barcode text extents of its records: record i's barcode = bytes[bc_pos[i] .. +bc_len[i])
 *     (caller)          ids[i] = dense id of that barcode (one dictionary per job, shared by all GPUs; < n_barcodes of
 *                       the context's counters)
 *     hast_fq_commit    copies the ids over and runs the per-barcode bookkeeping (classify.cpp:203-208); frees the buffer
 * Blocks may be submitted ahead of hast_fq_next (n_buffers of them); records that straddle two blocks are handled on the
 * device (the unfinished tail of a block is put in front of the next one), up to 4 MB per record.  After a block submitted
 * with last != 0 the next one starts a new file.  One hast_fq per input stream; several may share a context. */
typedef struct hast_fq hast_fq;
typedef struct {
    uint64_t n_records;        /* records framed in this block */
    uint64_t n_bases;          /* sum of their base-line lengths */
    uint32_t max_read_len;
    uint32_t short_read;       /* != 0: some read is shorter than K and holds no 'N' (the reference aborts, kmer.h:171);
                                  nothing of this block has been classified */
    const uint8_t *bytes;      /* host view of the block (with the previous block's tail in front of it); NULL for a block submitted
                                  with hast_fq_submit_device (hast_fq_block_host_bytes fetches it) */
    const uint32_t *bc_pos;    /* [n_records] */
    const uint32_t *bc_len;    /* [n_records] */
    const uint8_t *bc_text;    /* [n_records][16] or NULL: the barcode text itself, byte 0 = its length, bytes 1.. = the text;
                                  length byte 0xFF = longer than 15 bytes, take it from bytes[bc_pos ..) */
    uint32_t *ids;             /* [n_records]: ids the device-side name cache knew are filled in; the caller fills the others */
    const uint32_t *unknown;   /* [n_unknown] indices of the records whose ids the caller has to fill in, or NULL: all of them */
    uint64_t n_unknown;
    uint64_t dict_ids;         /* a stream over a device dictionary (hast_names_create_dict): ids it had handed out when this block was
                                  named -- every id of the block is below; the caller's counters must hold them before hast_fq_commit */
} hast_fq_block;
/* Device-side cache barcode text -> id of one GPU, shared by the FASTQ streams of all contexts on that GPU: barcodes repeat (hundreds of
 * reads each), so after the first blocks the framer names almost every record itself and the host only sees new barcodes
 * (and the ones longer than 15 bytes).  The ids still come from the caller's dictionary -- one per job, the same on every
 * GPU -- and hast_fq_commit teaches the cache what the caller named.  max_barcodes sizes it (2 x 32 B per barcode); more
 * barcodes than that are simply not cached. */
typedef struct hast_names hast_names;
hast_status hast_names_create(hast_ctx *, size_t max_barcodes, hast_names **out);
/* The table as the job's DICTIONARY on its GPU (round 6; classify.cpp:52-56: a barcode gets its map entry at its first sighting): the
 * framer's naming kernel hands out the dense ids itself -- 0 .. hast_names_limit() - 1 in the order in which new texts are met, one atomic
 * counter -- so no first sighting goes through the host.  Left to the caller (hast_fq_block.unknown, as before): texts longer than 15
 * bytes and what arrives once every id is out; the caller names those in an id range of its own at or above hast_names_limit().
 * hast_names_count: ids handed out so far; hast_names_texts: the text records (16 bytes: length byte + text) of ids [first, first + n),
 * read once at the end for printing.  Dictionaries of different GPUs number independently: their counters are merged by text
 * (hast_names_merge into the first dictionary's numbering, hast_counts_permute, then hast_counts_allreduce). */
hast_status hast_names_create_dict(hast_ctx *, size_t max_barcodes, hast_names **out);
size_t      hast_names_limit(const hast_names *);
hast_status hast_names_count(hast_names *, size_t *n_ids);
hast_status hast_names_texts(hast_names *, size_t first, size_t n, uint8_t *out16);
/* Two dictionaries, one numbering: ids_out[i] = the id that dst has for src's text of id first + i (i < n) -- a text dst does not know
 * yet gets dst's next id there and then.  On the GPU: src's text records are copied to dst's device (peer to peer when that is another
 * GPU) and go through dst's naming kernel; no host map.  HAST_ERR_TABLE_FULL when dst has no id left for a new text. */
hast_status hast_names_merge(hast_names *dst, hast_names *src, size_t first, size_t n, uint32_t *ids_out);
void        hast_names_destroy(hast_names *);
/* Entries the caller knows: n text records (16 bytes each: length byte + up to 15 bytes of text, the format of hast_fq_block.bc_text)
 * with their ids -- e.g. a table barcode -> class for a routing stream (hast_fq_set_route).  Returns when they are on the device. */
hast_status hast_names_insert(hast_names *, const uint8_t *text16, const uint32_t *ids, size_t n);
hast_status hast_fq_create(hast_ctx *, size_t block_bytes, int n_buffers, hast_names *names_or_null, hast_fq **out);
/* device_blocks != 0: a stream of device-side blocks only (hast_fq_device_block / hast_fq_submit_device, below): no pinned host
 * copy of the block buffers is set up (hast_fq_acquire hands out NULL), which saves ~0.1 s of page pinning per stream. */
hast_status hast_fq_create_ex(hast_ctx *, size_t block_bytes, int n_buffers, hast_names *names_or_null, int device_blocks, hast_fq **out);
/* The blocks of ONE input stream on several contexts in turn (block i -> ctxs[i % n_ctx], one context per GPU; several
 * contexts on one GPU also work): the reference spreads the reads of one file over all its workers (classify.cpp:211-219),
 * one process per file would leave all GPUs but two idle on HAST's two input files (HAST.sh:162-166).  A block is framed
 * on its own GPU from the number of newlines in front of it in the file -- each block's own count is a kernel queued behind
 * its upload, the host adds them up -- so no GPU waits for another one's framing.  A block owns the records that START in
 * it; their header and base lines may reach into the first bytes of the next block (at most min(1 MB, block_bytes): a
 * larger record is HAST_ERR_FORMAT), which are uploaded to both GPUs.  Differences to hast_fq_create streams: every block
 * but the last of a file must be full (n_bytes == hast_fq_block_bytes), and a block can be opened (hast_fq_poll /
 * hast_fq_next) once the block behind it has been submitted.  n_buffers_per_ctx >= 2 and at least 3 buffers in all.  names: one
 * cache per context, or NULL. */
hast_status hast_fq_create_striped(hast_ctx *const *ctxs, int n_ctx, size_t block_bytes, int n_buffers_per_ctx,
                                   hast_names *const *names_or_null, hast_fq **out);
/* device_blocks != 0: a striped stream of device-side blocks (hast_fq_device_block / hast_fq_submit_device) -- what a .gz file
 * inflated on the GPUs feeds (hast_gz_open_multi): block i's bytes are written on GPU i % n_ctx by the caller's kernels, the first
 * bytes of block i + 1 reach block i's view by a copy from that GPU to this one, no pinned host copy of the blocks is set up. */
hast_status hast_fq_create_striped_ex(hast_ctx *const *ctxs, int n_ctx, size_t block_bytes, int n_buffers_per_ctx,
                                      hast_names *const *names_or_null, int device_blocks, hast_fq **out);
int         hast_fq_lanes(const hast_fq *);                      /* contexts the stream's blocks rotate over (1 for hast_fq_create) */
uint64_t    hast_fq_lane_records(const hast_fq *, int lane);     /* records opened so far on that context (striped streams) */
void        hast_fq_destroy(hast_fq *);
size_t      hast_fq_block_bytes(const hast_fq *);
hast_status hast_fq_acquire(hast_fq *, uint8_t **host_buf);
hast_status hast_fq_submit(hast_fq *, size_t n_bytes, int last);
/* Blocks whose bytes are written ON THE DEVICE (e.g. by hast_gz_read_device: a .gz input inflated on the GPU): after
 * hast_fq_acquire (its host buffer stays unused), hast_fq_device_block gives the device address the block's bytes belong at and
 * the stream the writes must be enqueued on; hast_fq_submit_device then frames them where they lie -- no upload.  At most
 * n_buffers - 1 blocks may be in hand (hast_fq_device_block called, not yet submitted) at a time.  A stream takes
 * host blocks or device blocks, not both.  On a striped stream block i's address lies on GPU i % n_ctx and the stream is that
 * GPU's.  hast_fq_block.bytes is NULL for such a block:
 * hast_fq_block_host_bytes (valid between hast_fq_next and hast_fq_commit, for any block) fetches the host copy when the caller
 * needs the text behind bc_pos / bc_len -- a barcode longer than the 15 bytes bc_text holds. */
hast_status hast_fq_device_block(hast_fq *, uint8_t **d_block, hast_stream *fill_stream);
hast_status hast_fq_submit_device(hast_fq *, size_t n_bytes, int last);
hast_status hast_fq_block_host_bytes(hast_fq *, const uint8_t **bytes);
int         hast_fq_poll(hast_fq *);      /* 1: hast_fq_next would not have to wait for the framing of the oldest submitted block */
hast_status hast_fq_next(hast_fq *, hast_fq_block *out);
hast_status hast_fq_commit(hast_fq *);

/* ---- routing: steps 10 and 11 of the wrapper (classify_stlfr_reads.sh:155-190, quartering_fastq.awk:12-61) --------------------
 * After `classify` the wrapper runs a single-threaded awk program over every input once more: each record (four lines) goes to
 * <name>.{nobarcode,paternal,maternal,homozygous}.fastq by the list its barcode is in.  awk splits the header at every '#' or '/'
 * (-F '#|/'): NF > 1 and $2 != "0_0_0" -> $2 is looked up in the paternal, maternal, homozygous list (:22-35; in none: an ERROR
 * line, the record is dropped), otherwise the read has no barcode (:36-39).  A stream put into routing mode frames its blocks as
 * before (plain or striped, host or device blocks), does not classify them, and sorts the records of a block on the GPU into four
 * runs of whole records in input order, which arrive in pinned host memory:
 *     hast_fq_set_route     tables: text record of $2 -> class (1 paternal, 2 maternal, 3 homozygous; hast_names_create +
 *                           hast_names_insert), one per lane of a striped stream.  Between files only.  NULL: classify again.
 *     hast_fq_acquire / _submit / _device_block / _submit_device     as before
 *     hast_fq_next_routed   oldest submitted block: waits for its runs
 *     hast_fq_commit        the caller is through with the runs: the buffer is free again
 * What the device cannot decide stays exact because the caller decides it: a block holding a record whose $2 is longer than the 15
 * bytes a text record holds, or is in no table entry, comes back with host_block != 0 -- the view's bytes and every record's extent
 * and class (0..3; 0xFE: in no list; 0xFF: longer than 15 bytes; 0xFD: not a record of this block) -- and the caller routes that
 * block itself (the ERROR line needs the text); the partial record at the end of a file (fewer than four newlines; awk still
 * prints its lines) is handed over as `tail`. */
typedef struct {
    uint64_t n_records;            /* records in the four runs */
    uint64_t count[4];             /* per class: 0 nobarcode, 1 paternal, 2 maternal, 3 homozygous */
    const uint8_t *run[4];         /* pinned host memory, valid until hast_fq_commit */
    uint64_t run_bytes[4];
    int host_block;                /* != 0: route this block from bytes / rec_* instead (the runs hold the decided records only) */
    const uint8_t *bytes;
    const uint32_t *rec_start, *rec_len;
    const uint8_t *rec_class;
    uint64_t n_slots;
    const uint8_t *tail;           /* the partial record at the end of the file, or NULL */
    uint64_t tail_bytes;
} hast_fq_routed;
hast_status hast_fq_set_route(hast_fq *, hast_names *const *tables, int n_tables);
hast_status hast_fq_next_routed(hast_fq *, hast_fq_routed *out);

/* ---- gzip input decoded on the device (gzstream.h:47, classify.cpp:245-254: one zlib stream per .gz file) -----------------
 * HAST's real inputs are ordinary .fq.gz files: ONE deflate stream per file, which the reference inflates on the thread that
 * also frames the records.  hast_gz inflates such a file on the GPU: the COMPRESSED bytes cross PCIe (5-6 x fewer than the
 * FASTQ text) and the inflated bytes are written where the caller wants them in HBM -- e.g. straight into a block of the FASTQ
 * framer (hast_fq_submit_device), which reads them where they lie.  Method (hast_amd/csrc/gz_core.h, gz_chain.h, gz_kernels.hip):
 * every 16-KB chunk of the compressed bytes (round 6; 32 KB until then) searches its first dynamic-block header and is decoded by ONE WAVE into 16-bit
 * symbols with the 32 KB in front of it unknown ("marker" symbols); a chunk counts iff the chunk in front of it ended exactly
 * at its start (holes are decoded by follow-up jobs); windows are resolved in stream order, markers translated, and every
 * member's CRC-32 and ISIZE are checked (CRC by slices on the device, combined with GF(2) operators), so a decoding bug or a
 * damaged file cannot pass as data.  Any gzip file is taken: all block types, several members, header fields, trailing garbage
 * (ignored, as gzread does).
 * hast_gz_open: HAST_ERR_UNSUPPORTED when the path is not a regular file, does not start with a gzip member (zlib passes such a
 * file through as it is) or the device has no room -- the caller then inflates on the host; HAST_ERR_IO when it cannot be read.
 * hast_gz_read_device: the next up to `cap` bytes of the inflated stream, written to d_dst by a kernel on `stream` (NULL: the
 * context's); *n_out < cap at the end of the stream (0 = nothing left) -- or when damage follows what was delivered: a damaged or
 * truncated file is HAST_ERR_IO only AFTER what could be decoded in front of the damage has been handed over, as gzread does, i.e.
 * with the call behind a short one.  A caller must therefore read on until a call returns 0 bytes (or an error) before it takes the
 * stream for complete.  Blocks until the bytes it returns are decoded.  HAST_GZ_CHUNK_BYTES / HAST_GZ_PASS_CHUNKS in the environment
 * set the geometry of hast_gz_open (tests: many passes over a small file).
 * A file of more than 2 GB is not kept whole on the device: its compressed bytes go round a ring there (hast_gz_stats.ring_bytes).
 * One reader per object; several of them -- one per input file -- run side by side. */
typedef struct hast_gz hast_gz;
typedef struct {
    uint64_t compressed_bytes, out_bytes;
    uint64_t chunks, accepted, followup_jobs, followup_rounds, followup_accepted, members;
    double decode_s;           /* search + decode passes (host wall time of the producer thread) */
    double windows_crc_s;      /* windows + CRC passes */
    double wait_upload_s;      /* the producer waited for compressed bytes to reach the device */
    double wait_consumer_s;    /* ... for the reader to be through with a symbol arena */
    double wait_decode_s;      /* the reader waited for decoded bytes */
    double open_s;             /* hast_gz_open itself: device buffers (the first symbol arena), streams, threads */
    uint64_t ring_bytes;       /* 0: the whole compressed file lies on the device; else the size of the ring it goes round in
                                * (files beyond 2 GB; HAST_GZ_RING_BYTES / HAST_GZ_PIECE_BYTES set the geometry in tests) */
    uint64_t upload_waited_for_ring;   /* pieces whose upload had to wait for the chain to move on */
    double chain_walk_s;       /* host time between the kernels: accepting chunks, planning follow-up jobs, combining CRC-32s -- the part
                                * of ONE deflate stream that stays serial however many GPUs decode its passes */
    uint64_t ring_laps;        /* times the upload position wrapped round the ring */
} hast_gz_stats;
hast_status hast_gz_open(hast_ctx *, const char *path, hast_gz **out);
/* test / tuning entry: compressed bytes per chunk (0 = 16384), chunks per pass (0 = 6144; 4096 with a chunk size given), symbols of room per
 * compressed byte (0 = 20 with the default geometry, else 12).  A pass's arena holds symbol slots for 70 % of its chunks (those WITH a block
 * start take one on the device; HAST_GZ_SLOT_FRACTION fixes the share, by default it follows what the passes find).  Two passes of a
 * stream are on the GPU at a time, on two streams, with three arenas to take turns (HAST_GZ_AHEAD=0: one pass, two arenas). */
hast_status hast_gz_open_ex(hast_ctx *, const char *path, size_t chunk_bytes, size_t chunks_per_pass, double room, hast_gz **out);
/* ONE .gz file inflated by several GPUs (the reference deals the reads of one file to all its workers whatever the file's encoding,
 * classify.cpp:211-219,245-254; HAST's inputs are two .fq.gz files, HAST.sh:162-166): the passes of the one deflate stream (4096
 * chunks each) go to the GPUs of ctxs[] in turn -- a chunk's decode into marker symbols needs nothing from its neighbours -- one
 * "unit" per distinct device (contexts that share a GPU share it), each with the whole compressed file, its own symbol arenas and
 * streams; the chain of accepted chunks is one, the 32-KB window a pass leaves travels to the next pass's GPU through pinned host
 * memory, CRC-32 / ISIZE are checked as for one GPU.  hast_gz_read_device then takes a destination on ANY of those GPUs (it asks
 * the runtime where the pointer lives): the bytes are translated on the unit that decoded them and, when that is another GPU,
 * copied over peer to peer -- e.g. into the device-side blocks of a striped FASTQ stream (hast_fq_create_striped).
 * HAST_GZ_SPLIT=contexts in the environment: one unit per CONTEXT even where contexts share a GPU, and every read through the
 * hand-over buffer + peer copy (the several-GPU paths on one GPU, for tests).  hast_gz_units: how many units the stream has. */
hast_status hast_gz_open_multi(hast_ctx *const *ctxs, int n_ctx, const char *path, hast_gz **out);
hast_status hast_gz_open_multi_ex(hast_ctx *const *ctxs, int n_ctx, const char *path, size_t chunk_bytes, size_t chunks_per_pass, double room, hast_gz **out);
int         hast_gz_units(const hast_gz *);
hast_status hast_gz_read_device(hast_gz *, uint8_t *d_dst, size_t cap, size_t *n_out, hast_stream);
hast_status hast_gz_get_stats(hast_gz *, hast_gz_stats *out);
void        hast_gz_close(hast_gz *);

/* ---- host-side pieces of the path (no device work) ---------------------------------------- */
/* parseName (classify.cpp:112-119): barcode = head[last '#' + 1 .. last '/'). */
void     hast_parse_barcode(const char *head, size_t len, size_t *start, size_t *n);
/* getHap (classify.cpp:66-86). */
int      hast_get_hap(const char *barcode, size_t blen, uint64_t c0, uint64_t c1,
                      uint64_t n_hap0, uint64_t n_hap1, double w0, double w1);
/* Kmer::str2Kmer (kmer.h:153-166) / chopRead2Kmer (kmer.h:169-194) on the host, used for the
 * <=2x(45-K+1) adaptor keys of InitAdaptor. */
uint64_t hast_canon_kmer(const char *s, int k);
size_t   hast_chop_read(const char *seq, size_t len, int k, uint64_t *out);
void     hast_kmer_to_str(uint64_t kmer, int k, char *out /* k+1 */);

/* ---- synthetic workload; identical generator on host and device -------------- */
typedef struct {
    uint64_t seed_k, seed_r, seed_b; /* 0 => defaults 0x4841535401/02/03 */
    uint64_t n_keys_per_hap;
    uint32_t n_barcodes;
    uint32_t read_len;
    uint32_t k;
    uint32_t reserved;               /* 0 = independent random keys; 1 = clustered keys: runs of K
                                        overlapping windows around variant sites, like real parent-specific k-mers */
} hast_synth_params;

hast_status hast_synth_keys_host(const hast_synth_params *, int hap, uint64_t first, size_t n, uint64_t *out);
hast_status hast_synth_reads_host(const hast_synth_params *, uint64_t first_read, size_t n_reads,
                                  uint8_t *bases /* n*read_len */, uint32_t *barcode_ids);
hast_status hast_synth_keys_device(hast_ctx *, const hast_synth_params *, int hap, uint64_t first, size_t n,
                                   uint64_t *d_out, hast_stream);
hast_status hast_synth_reads_device(hast_ctx *, const hast_synth_params *, uint64_t first_read, size_t n_reads,
                                    uint8_t *d_bases, uint32_t *d_barcode_ids, hast_stream);
/* generate + insert both haplotypes' keys on the device (no host copy of the keys) */
hast_status hast_synth_table_build(hast_ctx *, const hast_synth_params *);

/* ==== stage 00: parent-unique k-mer sets ======================================
 * Replaces the compute of 00.build_unshare_kmers_by_jellyfish/build_unshared_kmers.sh:165-291, which drives a
 * third-party CPU hash counter (jellyfish 2.3.0, vendored there as a binary) through seven count/dump passes over text
 * files.  Here ONE table in HBM holds, for every canonical k-mer of either parent, its paternal and its maternal count;
 * the histograms (analysis_kmercount.sh:7-9), and the two final sets (script lines 246-291)
 *     paternal.unique.filter = { x : p_lower <= count_pat(x) <= p_upper and count_mat(x) == 0 }   (and vice versa)
 * are read out of it directly.  Counting rules = `jellyfish count -m K -C`: every window of K consecutive bases,
 * both cases of ACGT are bases, any other byte ends the run; a k-mer and its reverse complement are one key.
 * parent: 0 = paternal, 1 = maternal.
 *
 * Key space slices: when the table cannot hold every distinct k-mer, hast_kc_sync reports HAST_ERR_TABLE_FULL;
 * the caller then processes the key space in n_slices passes over the input (hast_kc_set_slice), each pass
 * counting only the k-mers of its slice; histograms add up and selections concatenate over the slices.  The
 * same call shards the key space over several GPUs (one context per GPU, disjoint slices, no collective). */
typedef struct hast_kc hast_kc;
#define HAST_KC_HISTO_HIGH 10000u          /* counts above are lumped into bin HIGH+1 (jellyfish histo default) */

/* table_bytes == 0 (or more than that): 85 % of the free device memory.  k in [1,32]. */
hast_status hast_kc_create(int device_ordinal, int k, size_t table_bytes, hast_kc **out);
/* expected_windows: an upper bound on the k-mer occurrences the caller is going to count into this table (0 = unknown), e.g. from
 * the input files' sizes: the record buffers of the partitioned path then take room for that many windows' records instead of what
 * happens to be free on the device (tens of GB that a 20-Mbp job never fills, and that are slow to get right after another process
 * gave them back). */
hast_status hast_kc_create_ex(int device_ordinal, int k, size_t table_bytes, uint64_t expected_windows, hast_kc **out);
void        hast_kc_destroy(hast_kc *);
hast_stream hast_kc_stream(hast_kc *);
/* empties the table and restricts counting to slice `slice` of `n_slices` (1 slice = everything) */
hast_status hast_kc_set_slice(hast_kc *, uint32_t slice, uint32_t n_slices);
/* Count the k-mers of a byte stream for one parent.  The stream is a sequence of records' bases with at least one
 * non-base byte (e.g. '\n') between records; windows start at [0, n_bytes).  _device: bytes already in HBM,
 * asynchronous on the context's stream.  Host variant: copies through pinned staging, double-buffered; it may
 * return HAST_ERR_TABLE_FULL early (the device's word is polled without waiting), hast_kc_sync reports it for sure. */
hast_status hast_kc_count_device(hast_kc *, int parent, const uint8_t *d_bytes, size_t n_bytes);
hast_status hast_kc_count(hast_kc *, int parent, const uint8_t *bytes, size_t n_bytes);
/* wait for all counting submitted so far; HAST_ERR_TABLE_FULL when some k-mer found no slot */
hast_status hast_kc_sync(hast_kc *);
/* How the windows reach the table.  Tables of >= 2^20 buckets (128 MB) with a K a record has room for (K <= 27 at the default
 * minimizer length) count by PARTITIONING (hast_amd/csrc/kc_kernels.hip; kc_common.h "records", "PLACEMENT"): the windows are written
 * out as 8-byte records of a minimizer run each, the records are partitioned by bucket range in two levels down to slices of 512 or
 * 1024 buckets, and one workgroup per slice counts its records in LDS -- the table is read and written once per flush, sequentially,
 * instead of one memory-side atomic per minimizer run (k_kc_count, the DIRECT path of smaller tables and larger K, which stands at
 * the rate at which this part executes atomics).  Same counts, histograms and selections (tests); the two paths file a key in
 * different buckets, so a table is filled by one of them only.  HAST_KC_COUNT=atomic|partition in the environment at hast_kc_create
 * forces either.  Flushes happen at hast_kc_sync, before the table is read, and when the record buffer is nearly full (it takes what
 * is left of the device memory next to the table, at most one record per slot of the table; HAST_KC_RECORD_MB caps it); a flush of
 * few records for the table's size goes through the direct path where the records lie (HAST_KC_FLUSH=sweep|atomic pins that).
 * out[0] = 1: partitioned; out[1] flushes; out[2] records applied; out[3] windows that took the direct path after all (their four
 * buckets in the slice full); out[4] record capacity. */
hast_status hast_kc_partition_info(hast_kc *, uint64_t out[5]);
/* out[0..1] distinct k-mers per parent, out[2] keys in the table, out[3] table capacity (slots),
 * out[4..5] k-mer occurrences counted per parent */
hast_status hast_kc_stats(hast_kc *, uint64_t out[6]);
/* ADDS the count histogram of one parent to histo[0 .. HAST_KC_HISTO_HIGH+1] (`jellyfish histo`) */
hast_status hast_kc_histo(hast_kc *, int parent, uint64_t *histo);
/* find_bounds.awk:1-33 over the non-empty rows of a histogram: out = MIN_INDEX, MAX_INDEX, LOWER_INDEX, UPPER_INDEX.
 * Host-only arithmetic (no GPU needed). */
void        hast_kc_find_bounds(const uint64_t *histo, long out[4]);
/* Append to the context's selection of `parent` the keys of the current table with lower <= count <= upper that the
 * other parent does not have.  n_added may be NULL. */
hast_status hast_kc_select(hast_kc *, int parent, uint32_t lower, uint32_t upper, size_t *n_added);
/* forget what has been selected so far (e.g. before starting over with more slices) */
hast_status hast_kc_selection_clear(hast_kc *);
/* move what `src` has selected so far to the end of `dst`'s selections (key space split over several contexts) */
hast_status hast_kc_selection_adopt(hast_kc *dst, hast_kc *src);
/* free the table (the selections stay) */
hast_status hast_kc_release_table(hast_kc *);
/* sort the selection of `parent` (ascending = lexicographic order of the printed k-mers) and keep it on the device */
hast_status hast_kc_selection_sort(hast_kc *, int parent, size_t *n);
/* rows [first, first+count) of the sorted selection as text: count lines of K upper-case letters + '\n', the member of
 * {k-mer, reverse complement} that comes first in the order A<C<G<T (what the reference's .mer files hold) */
hast_status hast_kc_selection_text(hast_kc *, int parent, size_t first, size_t count, char *out);
/* the same rows as stage-01 table keys (hast_table_insert_keys), no text round trip */
hast_status hast_kc_selection_keys(hast_kc *, int parent, size_t first, size_t count, uint64_t *out);

/* synthetic trio for benches and tests: two parental genomes = one random genome + parent-specific SNPs; reads are
 * sampled with substitution errors on a random strand; the stream is read_len bases + '\n' per read */
typedef struct hast_kc_synth {
    uint64_t seed;            /* 0 => default */
    uint64_t genome_len;
    uint32_t read_len;
    uint32_t snp_per_1024;    /* parent-specific SNPs per 1024 bases */
    uint32_t err_per_4096;    /* substitution errors per 4096 bases */
    uint32_t n_per_4096;      /* reads with one 'N' per 4096 reads */
} hast_kc_synth;
hast_status hast_kc_synth_host(const hast_kc_synth *, int parent, uint64_t first_read, size_t n_reads, uint8_t *out);
hast_status hast_kc_synth_device(hast_kc *, const hast_kc_synth *, int parent, uint64_t first_read, size_t n_reads, uint8_t *d_out);

#ifdef __cplusplus
}
#endif
#endif

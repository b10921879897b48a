/*
 * hast_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C11) of the reference's stage-01 read-classification path:
 *   /root/reference/01.classify_stlfr_reads/classify.cpp   (cited below as classify.cpp:N)
 *   /root/reference/01.classify_stlfr_reads/kmer/kmer.h    (cited below as kmer.h:N)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (include/hast.h, hast_amd/csrc) never links, loads or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks it against
 *   (i)  the reference's own known-answer vectors, TestAll() classify.cpp:341-367, and
 *   (ii) golden outputs of the real reference binary (oracle/_ref/classify, compiled from the
 *        reference sources in this container by oracle/Makefile) committed under tests/golden/.
 */
#ifndef HAST_ORACLE_H
#define HAST_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- L0 primitives (kmer.h) ------------------------------------------------------------ */
uint8_t  ho_base2int(uint8_t c);                         /* kmer.h:11  (c&6)>>1  A0 C1 T2 G3   */
char     ho_int2base(int code);                          /* kmer.h:12  "ACTG"[code]            */
uint64_t ho_mask(int k);                                 /* kmer.h:129-148 createFilter (K<=32)*/
uint64_t ho_pack(const char *s, int k);                  /* kmer.h:156-160 first base MSB      */
uint64_t ho_revcomp(uint64_t fwd, int k);                /* kmer.h:196-210 fastReverseComp     */
uint64_t ho_canon_str(const char *s, int k);             /* kmer.h:153-166 str2Kmer            */
/* kmer.h:169-194 chopRead2Kmer, rolling fwd/rc exactly as the reference does.
 * Writes len-k+1 canonical k-mers, returns that count; returns 0 when len<k (reference asserts). */
size_t   ho_chop_read(const char *seq, size_t len, int k, uint64_t *out);
void     ho_kmer_to_str(uint64_t kmer, int k, char *out /* k+1 bytes */);   /* kmer.h:244-254 */

/* classify.cpp:112-119 parseName: barcode = head[last '#'+1 .. last '/')                     */
void     ho_parse_name(const char *head, size_t len, size_t *start, size_t *n);
/* classify.cpp:66-86 getHap.  c0/c1 = counts for key 0/1 (0 == key absent), n0/n1 = set sizes */
int      ho_get_hap(const char *barcode, size_t blen, int64_t c0, int64_t c1,
                    uint64_t n0, uint64_t n1, double w0, double w1);

/* ---- classifier state (g_kmers[2], g_K, weights, adaptors, BarcodeCache) ---------------- */
typedef struct ho_classifier ho_classifier;

ho_classifier *ho_new(void);
void     ho_free(ho_classifier *);
void     ho_set_weights(ho_classifier *, double w0, double w1);          /* classify.cpp:22-23 */
int      ho_k(const ho_classifier *);
uint64_t ho_set_size(const ho_classifier *, int hap);                    /* g_kmers[h].size()  */
uint64_t ho_lines_loaded(const ho_classifier *, int hap);                /* "Recorded N" count */

/* classify.cpp:30-46 load_kmers.  hap 0 must be loaded first (defines K).  Returns 0 on
 * success, <0 on error (unopenable file / wrong-length line; the reference hangs / asserts).  */
int      ho_load_kmers_file(ho_classifier *, const char *path, int hap);
int      ho_load_kmers_text(ho_classifier *, const char *text, size_t nbytes, int hap);
/* Bulk entry for large synthetic sets: keys are already canonical 2K-bit values.            */
int      ho_load_keys(ho_classifier *, const uint64_t *canon_keys, size_t n, int hap, int k);
int      ho_load_keys_mt(ho_classifier *, const uint64_t *canon_keys, size_t n, int hap, int k, int threads);
int      ho_contains(const ho_classifier *, int hap, uint64_t canon_key);

/* classify.cpp:314-339 InitAdaptor: erase every canonical k-mer of both adaptors from both
 * sets.  log may be NULL.  Returns number of erased (hap,key) pairs.                         */
int      ho_init_adaptor(ho_classifier *, const char *adaptor_f, const char *adaptor_r, FILE *log);

/* classify.cpp:186-209 process_reads on one record (string-keyed barcode cache).            */
int      ho_process_read(ho_classifier *, const char *head, size_t hlen,
                         const char *seq, size_t slen);
/* classify.cpp:238-278 processFastq (record framing, .gz by suffix).                        */
int      ho_process_fastq(ho_classifier *, const char *path);
/* classify.cpp:93-102 printBarcodeInfos: byte-sorted rows "barcode\thap\tc0\tc1\n".          */
int      ho_print(ho_classifier *, FILE *out);
size_t   ho_n_barcodes(const ho_classifier *);

/* Per-read votes only (classify.cpp:188-202): returns 0 and sets v0/v1; has_n=1 => skipped. */
void     ho_read_votes(const ho_classifier *, const char *seq, size_t slen,
                       uint32_t *v0, uint32_t *v1, int *has_n);

/* Id-keyed bulk path used against the GPU and as bench.py's cpu_baseline ("port"):
 * same per-read semantics (classify.cpp:186-209) with barcodes given as dense ids:
 *   c0[id]+=v0 (if >0), c1[id]+=v1 (if >0), neg[id]+=1 (if N or both zero), seen[id]+=1.
 * threads>=1 pthread workers over contiguous read ranges, shared counters with atomic adds
 * (the reference: t workers + private maps merged at the end, classify.cpp:226-229).          */
int      ho_classify_ids(const ho_classifier *, const uint8_t *bases, const uint64_t *offsets,
                         const uint32_t *barcode_ids, size_t n_reads,
                         uint32_t *c0, uint32_t *c1, uint32_t *neg, uint32_t *seen, int threads);
/* same, and votes[2r], votes[2r+1] (optional) get read r's own (vote0, vote1) */
int      ho_classify_ids_votes(const ho_classifier *, const uint8_t *bases, const uint64_t *offsets,
                               const uint32_t *barcode_ids, size_t n_reads, uint32_t *c0, uint32_t *c1,
                               uint32_t *neg, uint32_t *seen, uint32_t *votes, int threads);

/* ---- secondary oracle: the per-read classifier of stage 03 (BASELINE config 5) ------------------
 * Restatement of /root/reference/03.mkoutput_by_fabulous2.0/src_main/classify.cpp (cited s03:N):
 * k-mers are raw case-sensitive STRINGS; each set holds every line and its reverse complement
 * (s03:59-60,64-65, complement map s03:25-36, unknown bytes map to '\0'); every window of a read is looked
 * up as a substring (s03:209-214), no N handling at all; density = hits / line count (s03:68,216). */
typedef struct ho_s03 ho_s03;
ho_s03  *ho_s03_new(void);
void     ho_s03_free(ho_s03 *);
int      ho_s03_load_text(ho_s03 *, const char *text, size_t nbytes, int hap);   /* s03:51-70 load_kmers    */
int      ho_s03_load_file(ho_s03 *, const char *path, int hap);
int      ho_s03_k(const ho_s03 *);
uint64_t ho_s03_lines(const ho_s03 *, int hap);                                  /* total_kmers[hap]        */
/* s03:203-214: integer hit counts of one read (the parity quantity; densities are host-side doubles) */
void     ho_s03_read_hits(const ho_s03 *, const char *seq, size_t slen, uint32_t *h0, uint32_t *h1);
/* s03:104-135 PrintOutput row for one read: writes "name\tlabel\tvalue\n" into out (>= nlen+64 bytes) */
int      ho_s03_format_row(const ho_s03 *, const char *name, size_t nlen, uint32_t h0, uint32_t h1, char *out);
/* s03:248-302 processFastq / processFasta into `out` (rows ordered by read id). format: 0 fasta, 1 fastq */
int      ho_s03_process_file(const ho_s03 *, const char *path, int format, FILE *out);

#ifdef __cplusplus
}
#endif
#endif

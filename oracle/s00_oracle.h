/*
 * s00_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C11) of the reference's stage 00, "build parent-unique k-mer sets"
 * (SURVEY 8(f) #4):
 *   /root/reference/00.build_unshare_kmers_by_jellyfish/build_unshared_kmers.sh   (cited as s00.sh:N)
 *   /root/reference/00.build_unshare_kmers_by_jellyfish/analysis_kmercount.sh     (cited as ana.sh:N)
 *   /root/reference/00.build_unshare_kmers_by_jellyfish/find_bounds.awk           (cited as bounds.awk:N)
 * The counting itself is done in the reference by a THIRD-PARTY program that is vendored as a binary only:
 * jellyfish 2.3.0 (00.build_unshare_kmers_by_jellyfish/jellyfish-linux, static ELF; no sources in the reference).
 * Its published behaviour for the sub-commands the script uses is restated here:
 *   count -m K -C   : every window of K consecutive bases of a FASTA/FASTQ record, both cases of ACGT accepted,
 *                     any other byte breaks the run; key = canonical k-mer = min(k-mer, reverse complement) in
 *                     the order A<C<G<T; exact counts.
 *   dump [-L l][-U u]: the keys whose count is in [l, u] (text = the canonical representative, upper case).
 *   histo           : rows "count number" for count 1..10000, counts above 10000 are lumped into row 10001, empty
 *                     rows are not printed.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Parity status: PINNED against the real thing run in this container: tests/golden/gen_golden.py runs the
 * reference's own script (which runs the vendored jellyfish) on small inputs and commits inputs and outputs under
 * tests/golden/s00_*; tests/test_oracle_golden.py checks this restatement against them.  The row ORDER of the
 * reference's .mer files is jellyfish's hash order (not reproducible without jellyfish); nothing downstream depends
 * on it (stage 01 loads the lines into a set, classify.cpp:30-46), so parity is on the sorted lines.
 */
#ifndef HAST_S00_ORACLE_H
#define HAST_S00_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HO_S00_HISTO_HIGH 10000          /* jellyfish histo default --high */

typedef struct ho_s00 ho_s00;

ho_s00  *ho_s00_new(int k);              /* k in [1,32] (the script wants >= 11, s00.sh:141-151) */
void     ho_s00_free(ho_s00 *);
int      ho_s00_k(const ho_s00 *);

/* jellyfish coding: A0 C1 G2 T3, first base most significant; canonical = min(fwd, revcomp) */
uint64_t ho_s00_canon_str(const char *s, int k);
void     ho_s00_key_to_str(uint64_t key, int k, char *out /* k+1 bytes */);

/* count the k-mers of one record's sequence (lines already joined) for parent p (0 = paternal, 1 = maternal) */
void     ho_s00_add_seq(ho_s00 *, int parent, const char *seq, size_t len);
/* a byte stream in which ANY byte outside ACGTacgt separates runs (what the GPU entry point takes) */
void     ho_s00_add_stream(ho_s00 *, int parent, const uint8_t *bytes, size_t n);
/* one `jellyfish count` input: FASTA or FASTQ by its first byte, multi-line records; `paths` are read as ONE
 * concatenated stream when gz != 0 (s00.sh:187-188: zcat files | jellyfish ... /dev/fd/0) and one by one otherwise
 * (s00.sh:190).  Returns 0, or <0: -1 unopenable, -2 unsupported format,
 * -3 malformed FASTQ (see s00_oracle.c: jellyfish silently loses data there). */
int      ho_s00_add_files(ho_s00 *, int parent, const char *const *paths, int n_paths, int gz);

uint64_t ho_s00_distinct(const ho_s00 *, int parent);
uint64_t ho_s00_total(const ho_s00 *, int parent);                 /* k-mer occurrences counted */
uint32_t ho_s00_count(const ho_s00 *, int parent, uint64_t canon_key);

/* jellyfish histo (ana.sh:7-9): out[c] for c in [0, HIGH+1], out[HIGH+1] lumps everything above HIGH */
void     ho_s00_histo(const ho_s00 *, int parent, uint64_t *out /* HO_S00_HISTO_HIGH + 2 */);
int      ho_s00_write_histo(const uint64_t *histo, FILE *f);       /* "count number\n", empty rows skipped */
/* bounds.awk:1-33 over the printed rows; returns MIN_INDEX, MAX_INDEX, LOWER_INDEX, UPPER_INDEX */
void     ho_s00_find_bounds(const uint64_t *histo, long *min_index, long *max_index, long *lower, long *upper);

/* s00.sh:246-291: keys of `parent` with lower <= count <= upper that do not occur in the other parent at all.
 * Sorted ascending (= lexicographic order of the printed strings).  out may be NULL to get the number only. */
size_t   ho_s00_select(const ho_s00 *, int parent, long lower, long upper, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif

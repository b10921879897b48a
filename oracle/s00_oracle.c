/*
 * s00_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See s00_oracle.h for what is restated and how it is pinned.
 */
#include "s00_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ---- canonical k-mers in jellyfish's coding (A0 C1 G2 T3) ------------------------------------- */
static int jf_code(uint8_t c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;                                   /* breaks the run of bases */
    }
}
static uint64_t mask_of(int k) { return k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1); }

uint64_t ho_s00_canon_str(const char *s, int k) {
    uint64_t f = 0, r = 0;
    for (int i = 0; i < k; i++) {
        uint64_t c = (uint64_t)jf_code((uint8_t)s[i]);
        f = (f << 2) | c;
        r = (r >> 2) | ((3 - c) << (2 * (k - 1)));
    }
    return f < r ? f : r;
}
void ho_s00_key_to_str(uint64_t key, int k, char *out) {
    for (int i = 0; i < k; i++) out[i] = "ACGT"[(key >> (2 * (k - 1 - i))) & 3];
    out[k] = 0;
}

/* ---- count table: open addressing, key -> two counters ------------------------------------------ */
typedef struct { uint64_t key; uint32_t c[2]; } cell;      /* empty: c[0] == c[1] == 0 */
struct ho_s00 {
    int k;
    cell *cells;
    size_t cap, used;                                      /* cap is a power of two */
    uint64_t total[2];
};
static size_t cell_home(uint64_t key, size_t cap) {
    uint64_t h = key * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    return (size_t)(h * 0xBF58476D1CE4E5B9ull >> 17) & (cap - 1);
}
ho_s00 *ho_s00_new(int k) {
    if (k < 1 || k > 32) return NULL;
    ho_s00 *o = calloc(1, sizeof *o);
    o->k = k;
    o->cap = 1 << 16;
    o->cells = calloc(o->cap, sizeof(cell));
    return o;
}
void ho_s00_free(ho_s00 *o) {
    if (!o) return;
    free(o->cells);
    free(o);
}
int ho_s00_k(const ho_s00 *o) { return o->k; }

static cell *cell_find(const ho_s00 *o, uint64_t key) {
    size_t i = cell_home(key, o->cap);
    for (;;) {
        cell *c = &o->cells[i];
        if ((c->c[0] | c->c[1]) == 0 || c->key == key) return c;
        i = (i + 1) & (o->cap - 1);
    }
}
static void grow(ho_s00 *o) {
    cell *old = o->cells;
    size_t oc = o->cap;
    o->cap *= 2;
    o->cells = calloc(o->cap, sizeof(cell));
    for (size_t i = 0; i < oc; i++)
        if (old[i].c[0] | old[i].c[1]) *cell_find(o, old[i].key) = old[i];
    free(old);
}
static void bump(ho_s00 *o, int parent, uint64_t key) {
    cell *c = cell_find(o, key);
    if ((c->c[0] | c->c[1]) == 0) {
        c->key = key;
        o->used++;
    }
    c->c[parent]++;
    o->total[parent]++;
    if (o->used * 10 > o->cap * 6) grow(o);
}

void ho_s00_add_seq(ho_s00 *o, int parent, const char *seq, size_t len) {
    const int k = o->k;
    const uint64_t m = mask_of(k);
    uint64_t f = 0, r = 0;
    int run = 0;
    for (size_t i = 0; i < len; i++) {
        int c = jf_code((uint8_t)seq[i]);
        if (c < 0) { run = 0; continue; }
        f = ((f << 2) | (uint64_t)c) & m;
        r = (r >> 2) | ((uint64_t)(3 - c) << (2 * (k - 1)));
        if (++run >= k) bump(o, parent, f < r ? f : r);
    }
}
void ho_s00_add_stream(ho_s00 *o, int parent, const uint8_t *bytes, size_t n) {
    ho_s00_add_seq(o, parent, (const char *)bytes, n);     /* a separator is just another non-base byte */
}

/* ---- FASTA / FASTQ records as jellyfish reads them ------------------------------------------------ */
typedef struct { char *p; size_t n, cap; } buf;
static void buf_add(buf *b, const char *s, size_t n) {
    if (b->n + n + 1 > b->cap) {
        b->cap = (b->n + n + 1) * 2;
        b->p = realloc(b->p, b->cap);
    }
    memcpy(b->p + b->n, s, n);
    b->n += n;
}
static int slurp(const char *path, int gz, buf *b) {
    char tmp[1 << 16];
    if (gz) {
        gzFile f = gzopen(path, "rb");
        if (!f) return -1;
        int r;
        while ((r = gzread(f, tmp, sizeof tmp)) > 0) buf_add(b, tmp, (size_t)r);
        gzclose(f);
    } else {
        FILE *f = fopen(path, "rb");
        if (!f) return -1;
        size_t r;
        while ((r = fread(tmp, 1, sizeof tmp, f)) > 0) buf_add(b, tmp, r);
        fclose(f);
    }
    return 0;
}
/* next line of [p, end): *len excludes the '\n' and any '\r' in front of it (probed: jellyfish joins "ACG\r\nTTT" to
 * ACGTTT, while a '\r' inside a line breaks the run like any other byte); returns the start of the following line */
static const char *next_line(const char *p, const char *end, size_t *len) {
    const char *nl = memchr(p, '\n', (size_t)(end - p));
    size_t n = nl ? (size_t)(nl - p) : (size_t)(end - p);
    while (n && p[n - 1] == '\r') n--;
    *len = n;
    return nl ? nl + 1 : end;
}
static const char *skip_blank(const char *p, const char *end) {
    while (p < end && (*p == '\n' || *p == '\r')) p++;
    return p;
}
static int parse_stream(ho_s00 *o, int parent, const char *p, size_t n) {
    if (n == 0) return 0;
    const char *end = p + n;
    buf seq = {0, 0, 0};
    size_t len;
    if (*p == '>') {
        while (p < end) {
            const char *line = p;
            p = next_line(p, end, &len);
            if (len && line[0] == '>') {                     /* header: the previous record is complete */
                ho_s00_add_seq(o, parent, seq.p, seq.n);
                seq.n = 0;
            } else buf_add(&seq, line, len);                 /* sequence lines are joined */
        }
        ho_s00_add_seq(o, parent, seq.p, seq.n);
    } else if (*p == '@') {
        /* Probed on the vendored jellyfish 2.3.0: blank lines are skipped wherever a line may start; the quality
         * string must have exactly as many bytes as the sequence (over any number of lines) and be followed by a
         * '@' line or the end of input, otherwise jellyfish silently loses the buffer it was filling (we report
         * -3 instead); a record without '+' at the end of input still counts.  One more loss we do NOT restate:
         * when the LAST quality line of an input has no '\n', jellyfish drops its last buffer of reads as well. */
        while (p < end) {
            if (*p != '@') { free(seq.p); return -3; }
            p = next_line(p, end, &len);                     /* header */
            seq.n = 0;
            for (;;) {                                       /* sequence lines up to the '+' line */
                p = skip_blank(p, end);
                if (p >= end || *p == '+') break;
                const char *line = p;
                p = next_line(p, end, &len);
                buf_add(&seq, line, len);
            }
            if (p < end) {
                p = next_line(p, end, &len);                 /* the '+' line */
                size_t q = 0;
                while (p < end && q < seq.n) {               /* as many quality bytes as bases */
                    p = next_line(p, end, &len);
                    q += len;
                }
                if (q != seq.n) { free(seq.p); return -3; }
                p = skip_blank(p, end);
            }
            ho_s00_add_seq(o, parent, seq.p, seq.n);
        }
    } else {
        free(seq.p);
        return -2;
    }
    free(seq.p);
    return 0;
}
int ho_s00_add_files(ho_s00 *o, int parent, const char *const *paths, int n_paths, int gz) {
    buf b = {0, 0, 0};
    int rc = 0;
    for (int i = 0; i < n_paths && rc == 0; i++) {
        if (slurp(paths[i], gz, &b)) rc = -1;
        else if (!gz) {                                      /* plain files are separate inputs */
            rc = parse_stream(o, parent, b.p, b.n);
            b.n = 0;
        }
    }
    if (rc == 0 && gz) rc = parse_stream(o, parent, b.p, b.n);
    free(b.p);
    return rc;
}

/* ---- results ---------------------------------------------------------------------------------------- */
uint64_t ho_s00_distinct(const ho_s00 *o, int parent) {
    uint64_t n = 0;
    for (size_t i = 0; i < o->cap; i++) n += o->cells[i].c[parent] != 0;
    return n;
}
uint64_t ho_s00_total(const ho_s00 *o, int parent) { return o->total[parent]; }
uint32_t ho_s00_count(const ho_s00 *o, int parent, uint64_t key) {
    const cell *c = cell_find(o, key);
    return ((c->c[0] | c->c[1]) && c->key == key) ? c->c[parent] : 0;
}
void ho_s00_histo(const ho_s00 *o, int parent, uint64_t *out) {
    memset(out, 0, (HO_S00_HISTO_HIGH + 2) * sizeof *out);
    for (size_t i = 0; i < o->cap; i++) {
        uint32_t c = o->cells[i].c[parent];
        if (c) out[c > HO_S00_HISTO_HIGH ? HO_S00_HISTO_HIGH + 1 : c]++;
    }
}
int ho_s00_write_histo(const uint64_t *h, FILE *f) {
    for (int c = 1; c <= HO_S00_HISTO_HIGH + 1; c++)
        if (h[c] && fprintf(f, "%d %llu\n", c, (unsigned long long)h[c]) < 0) return -1;
    return 0;
}
void ho_s00_find_bounds(const uint64_t *h, long *min_index, long *max_index, long *lower, long *upper) {
    /* bounds.awk:8-25.  The script tests a variable `S` that is never assigned before (awk: 0); rows come in
     * ascending count order; the row that ends the descent is not considered for the maximum. */
    double MIN = 0, MAX = 0;
    long MIN_INDEX = 0, MAX_INDEX = 0;
    int S = 0;
    for (long i = 1; i <= HO_S00_HISTO_HIGH + 1; i++) {
        if (!h[i]) continue;                                  /* not printed by histo */
        double c = (double)h[i];
        if (S == 0) {
            if (MIN == 0 || c < MIN) { MIN = c; MIN_INDEX = i; }
            else S = 1;
        } else if (MAX == 0 || c > MAX) { MAX = c; MAX_INDEX = i; }
    }
    const long up_bounds = 3 * MAX_INDEX - 2 * MIN_INDEX;     /* bounds.awk:28 */
    *min_index = MIN_INDEX;
    *max_index = MAX_INDEX;
    *lower = MIN_INDEX + 1;
    *upper = up_bounds - 1;
}
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}
size_t ho_s00_select(const ho_s00 *o, int parent, long lower, long upper, uint64_t *out) {
    /* s00.sh:246-252 dump -L l -U u (filter); :262-268 keys seen in exactly one parent (unique);
     * :276-283 the intersection of both. */
    size_t n = 0;
    for (size_t i = 0; i < o->cap; i++) {
        const cell *c = &o->cells[i];
        if (c->c[parent] == 0 || c->c[1 - parent] != 0) continue;
        if ((long)c->c[parent] < lower || (long)c->c[parent] > upper) continue;
        if (out) out[n] = c->key;
        n++;
    }
    if (out) qsort(out, n, sizeof *out, cmp_u64);
    return n;
}

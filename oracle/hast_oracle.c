/*
 * hast_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See hast_oracle.h.
 *
 * CPU restatement of HAST stage-01 `classify` (classify.cpp + kmer/kmer.h).  Every function
 * cites the reference lines it follows.  The reference keeps a 128-bit Kmer{high,low}; for
 * K<=32 `high` is always 0 (kmer.h:94-100 only spills into `high` past 64 bits), so a single
 * uint64_t carries the same value.  K>=33 is silently broken in the reference and out of
 * contract here (ho_load_* reject it).
 *
 * Parity status: PINNED (reference KATs + golden outputs of the real reference binary,
 * tests/test_oracle_golden.py).
 */
#define _GNU_SOURCE
#include "hast_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------------------------------
 * L0: base coding and k-mer arithmetic (kmer.h)
 * ---------------------------------------------------------------------------------------- */

uint8_t ho_base2int(uint8_t c) { return (uint8_t)((c & 0x06) >> 1); }      /* kmer.h:11 */
char ho_int2base(int code) { return "ACTG"[code & 3]; }                      /* kmer.h:12 */

/* kmer.h:129-148 createFilter: low 2K bits set (all 64 when 2K==64). */
uint64_t ho_mask(int k) { return (2 * k < 64) ? ((1ULL << (2 * k)) - 1) : ~0ULL; }

/* kmer.h:156-160: shift left by 2, OR the next code => first base most significant. */
uint64_t ho_pack(const char *s, int k) {
    uint64_t w = 0;
    for (int i = 0; i < k; i++) w = (w << 2) | ho_base2int((uint8_t)s[i]);
    return w;
}

/* kmer.h:196-210 fastReverseComp for seq_size<32, and the seq_size==32 path (:212-222,
 * RightBitMove(64) :225-238 leaves low = bit-reversed-by-pairs complement, high = 0). */
uint64_t ho_revcomp(uint64_t x, int k) {
    x ^= 0xAAAAAAAAAAAAAAAAULL;                                               /* code ^ 2 */
    x = ((x & 0x3333333333333333ULL) << 2) | ((x & 0xCCCCCCCCCCCCCCCCULL) >> 2);
    x = ((x & 0x0F0F0F0F0F0F0F0FULL) << 4) | ((x & 0xF0F0F0F0F0F0F0F0ULL) >> 4);
    x = ((x & 0x00FF00FF00FF00FFULL) << 8) | ((x & 0xFF00FF00FF00FF00ULL) >> 8);
    x = ((x & 0x0000FFFF0000FFFFULL) << 16) | ((x & 0xFFFF0000FFFF0000ULL) >> 16);
    x = ((x & 0x00000000FFFFFFFFULL) << 32) | ((x & 0xFFFFFFFF00000000ULL) >> 32);
    if (k < 32) x >>= (64 - 2 * k);
    return x;
}

/* kmer.h:153-166 str2Kmer: word < bal_word ? word : bal_word. */
uint64_t ho_canon_str(const char *s, int k) {
    uint64_t w = ho_pack(s, k), b = ho_revcomp(w, k);
    return (w < b) ? w : b;
}

/* kmer.h:169-194 chopRead2Kmer.  fwd rolls with nextKmer (kmer.h:109-114), rc rolls with
 * prevKmer (kmer.h:116-127) fed from the reverse-complemented read (kmer.h:38-52,174,187). */
size_t ho_chop_read(const char *seq, size_t len, int k, uint64_t *out) {
    if ((size_t)k > len) return 0;                          /* reference: assert kmer.h:171 */
    const uint64_t mask = ho_mask(k);
    uint64_t w = ho_pack(seq, k);
    uint64_t b = ho_revcomp(w, k);
    size_t n = 0;
    out[n++] = (w < b) ? w : b;
    for (size_t index = 1; index + (size_t)k <= len; index++) {
        uint8_t ch = ho_base2int((uint8_t)seq[index - 1 + k]);
        w = ((w << 2) & mask) | ch;                                        /* nextKmer */
        /* bal_read[rlen-index-overlap] == int_comp(read[index+overlap-1]) (kmer.h:47-50) */
        uint64_t cc = (uint64_t)(ch ^ 0x02);
        b = (b >> 2) | (cc << (2 * (k - 1)));                              /* prevKmer */
        out[n++] = (w < b) ? w : b;
    }
    return n;
}

/* kmer.h:244-254 ToBaseStr + kmer.h:14-25 BaseStr2Str. */
void ho_kmer_to_str(uint64_t kmer, int k, char *out) {
    for (int i = 0; i < k; i++) {
        out[k - 1 - i] = ho_int2base((int)(kmer & 3));
        kmer >>= 2;
    }
    out[k] = 0;
}

/* classify.cpp:112-119 parseName: s = last '#', e = last '/', substr(s+1, e-s-1).
 * std::string::substr(pos, n) with n = (size_t)(e-s-1): a negative value wraps to npos-ish and
 * means "to the end"; pos > size() would throw (cannot happen: s < size). */
void ho_parse_name(const char *head, size_t len, size_t *start, size_t *n) {
    long s = -1, e = -1;
    for (size_t i = 0; i < len; i++) {
        if (head[i] == '#') s = (long)i;
        if (head[i] == '/') e = (long)i;
    }
    size_t pos = (size_t)(s + 1);
    long cnt = e - s - 1;
    size_t avail = len - pos;
    *start = pos;
    *n = (cnt < 0 || (size_t)cnt > avail) ? avail : (size_t)cnt;
}

/* classify.cpp:66-86 getHap.  A map key 0/1 exists iff its count > 0 (keys are only created
 * by IncrBarcodeHaps with a positive vote, classify.cpp:203-206). */
int ho_get_hap(const char *bc, size_t blen, int64_t c0, int64_t c1,
               uint64_t n0, uint64_t n1, double w0, double w1) {
    if ((blen == 5 && memcmp(bc, "0_0_0", 5) == 0) || (blen == 3 && memcmp(bc, "0_0", 3) == 0) ||
        (blen == 1 && bc[0] == '0'))
        return -1;
    if (c0 > 0 && c1 > 0) {
        double df0 = (double)c0 / (double)n0;
        double df1 = (double)c1 / (double)n1;
        df0 *= w0;
        df1 *= w1;
        if (df0 > df1) return 0;
        if (df1 > df0) return 1;
        return -1;
    } else if (c0 > 0) {
        return 0;
    } else if (c1 > 0) {
        return 1;
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------
 * std::unordered_set<Kmer> stand-in: open addressing, linear probing, backward-shift erase.
 * Only membership and size are observable in the reference (classify.cpp:27,38,42,70-71,
 * 196-201,319-336), so any exact set gives identical results.
 * ---------------------------------------------------------------------------------------- */
#define HO_EMPTY (~0ULL) /* never canonical: all-G's reverse complement all-C is smaller */

typedef struct {
    uint64_t *slot;
    uint64_t cap; /* power of two */
    uint64_t size;
} ho_set;

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

static int set_alloc(ho_set *s, uint64_t cap) {
    s->slot = (uint64_t *)malloc(cap * sizeof(uint64_t));
    if (!s->slot) return -1;
    memset(s->slot, 0xFF, cap * sizeof(uint64_t));
    s->cap = cap;
    s->size = 0;
    return 0;
}

static int set_insert_nogrow(ho_set *s, uint64_t key) {
    uint64_t m = s->cap - 1, i = mix64(key) & m;
    for (;;) {
        uint64_t v = s->slot[i];
        if (v == key) return 0;
        if (v == HO_EMPTY) {
            s->slot[i] = key;
            s->size++;
            return 1;
        }
        i = (i + 1) & m;
    }
}

static int set_reserve(ho_set *s, uint64_t n) {
    uint64_t need = 1024;
    while (need < n * 2) need <<= 1;
    if (s->slot && s->cap >= need) return 0;
    ho_set t;
    if (set_alloc(&t, need)) return -1;
    if (s->slot) {
        for (uint64_t i = 0; i < s->cap; i++)
            if (s->slot[i] != HO_EMPTY) set_insert_nogrow(&t, s->slot[i]);
        free(s->slot);
    }
    *s = t;
    return 0;
}

static int set_insert(ho_set *s, uint64_t key) {
    if (!s->slot || (s->size + 1) * 2 > s->cap)
        if (set_reserve(s, (s->size + 1) * 2)) return -1;
    return set_insert_nogrow(s, key);
}

static inline int set_contains(const ho_set *s, uint64_t key) {
    if (!s->slot) return 0;
    uint64_t m = s->cap - 1, i = mix64(key) & m;
    for (;;) {
        uint64_t v = s->slot[i];
        if (v == key) return 1;
        if (v == HO_EMPTY) return 0;
        i = (i + 1) & m;
    }
}

static int set_erase(ho_set *s, uint64_t key) {
    if (!s->slot) return 0;
    uint64_t m = s->cap - 1, i = mix64(key) & m;
    for (;;) {
        uint64_t v = s->slot[i];
        if (v == HO_EMPTY) return 0;
        if (v == key) break;
        i = (i + 1) & m;
    }
    /* backward-shift deletion keeps probe chains intact */
    uint64_t j = i;
    for (;;) {
        j = (j + 1) & m;
        uint64_t v = s->slot[j];
        if (v == HO_EMPTY) break;
        uint64_t h = mix64(v) & m;
        /* can v move to i?  yes iff its home h is not in the cyclic interval (i, j] */
        int in_between = (i <= j) ? (h > i && h <= j) : (h > i || h <= j);
        if (!in_between) {
            s->slot[i] = v;
            i = j;
        }
    }
    s->slot[i] = HO_EMPTY;
    s->size--;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * BarcodeCache stand-in (classify.cpp:50-64): string -> {key0,key1,key-1} counts.
 * Row order is restored at print time (std::map<std::string> order = byte-wise compare).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    char *name;
    uint32_t len;
    int32_t c0, c1, cneg; /* `int` in the reference, classify.cpp:51 */
} bc_entry;

typedef struct {
    bc_entry *ent;
    size_t n, cap;
    uint32_t *index; /* hash -> ent idx+1 */
    size_t icap;
} bc_map;

static uint64_t hash_bytes(const char *p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < n; i++) h = (h ^ (uint8_t)p[i]) * 0x100000001b3ULL;
    return mix64(h);
}

static void bc_rehash(bc_map *m, size_t icap) {
    free(m->index);
    m->index = (uint32_t *)calloc(icap, sizeof(uint32_t));
    m->icap = icap;
    for (size_t e = 0; e < m->n; e++) {
        size_t i = hash_bytes(m->ent[e].name, m->ent[e].len) & (icap - 1);
        while (m->index[i]) i = (i + 1) & (icap - 1);
        m->index[i] = (uint32_t)(e + 1);
    }
}

static bc_entry *bc_get(bc_map *m, const char *name, size_t len) {
    if (m->icap == 0) bc_rehash(m, 1024);
    size_t i = hash_bytes(name, len) & (m->icap - 1);
    while (m->index[i]) {
        bc_entry *e = &m->ent[m->index[i] - 1];
        if (e->len == len && memcmp(e->name, name, len) == 0) return e;
        i = (i + 1) & (m->icap - 1);
    }
    if (m->n == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 1024;
        m->ent = (bc_entry *)realloc(m->ent, m->cap * sizeof(bc_entry));
    }
    bc_entry *e = &m->ent[m->n++];
    e->name = (char *)malloc(len + 1);
    memcpy(e->name, name, len);
    e->name[len] = 0;
    e->len = (uint32_t)len;
    e->c0 = e->c1 = e->cneg = 0;
    m->index[i] = (uint32_t)m->n;
    if (m->n * 2 > m->icap) {
        bc_rehash(m, m->icap * 2);
        return &m->ent[m->n - 1];
    }
    return e;
}

static void bc_free(bc_map *m) {
    for (size_t i = 0; i < m->n; i++) free(m->ent[i].name);
    free(m->ent);
    free(m->index);
    memset(m, 0, sizeof(*m));
}

/* ------------------------------------------------------------------------------------------
 * classifier state
 * ---------------------------------------------------------------------------------------- */
struct ho_classifier {
    int k;                 /* g_K classify.cpp:29 */
    ho_set set[2];         /* g_kmers classify.cpp:27 */
    uint64_t lines[2];     /* total_kmer classify.cpp:33,45 */
    double w0, w1;         /* g_hap0_fac/g_hap1_fac classify.cpp:22-23 */
    bc_map barcodes;       /* BarcodeCache data classify.cpp:439 */
};

ho_classifier *ho_new(void) {
    ho_classifier *c = (ho_classifier *)calloc(1, sizeof(*c));
    c->w0 = c->w1 = 1.0;
    return c;
}

void ho_free(ho_classifier *c) {
    if (!c) return;
    free(c->set[0].slot);
    free(c->set[1].slot);
    bc_free(&c->barcodes);
    free(c);
}

void ho_set_weights(ho_classifier *c, double w0, double w1) { c->w0 = w0; c->w1 = w1; }
int ho_k(const ho_classifier *c) { return c->k; }
uint64_t ho_set_size(const ho_classifier *c, int hap) { return c->set[hap & 1].size; }
uint64_t ho_lines_loaded(const ho_classifier *c, int hap) { return c->lines[hap & 1]; }
size_t ho_n_barcodes(const ho_classifier *c) { return c->barcodes.n; }
int ho_contains(const ho_classifier *c, int hap, uint64_t key) { return set_contains(&c->set[hap & 1], key); }

/* classify.cpp:30-46 load_kmers on an in-memory image of the file.
 *   :35-36  first line of hap 0 defines K (its length), and is inserted unconditionally
 *   :41     while(!getline(...).eof()): a final piece with no '\n' sets eof => dropped
 *   :38,42  insert(str2Kmer(...)), str2Kmer asserts line length == K (kmer.h:154)           */
int ho_load_kmers_text(ho_classifier *c, const char *text, size_t nbytes, int hap) {
    size_t pos = 0;
    uint64_t total = 0;
    ho_set *s = &c->set[hap];
    if (hap == 0) {
        const char *nl = (const char *)memchr(text, '\n', nbytes);
        size_t l = nl ? (size_t)(nl - text) : nbytes;
        if (l < 1 || l > 32) return -2;
        c->k = (int)l;
        if (set_reserve(s, nbytes / (l + 1) + 16)) return -3;
        set_insert(s, ho_canon_str(text, c->k));
        total++;
        pos = nl ? l + 1 : nbytes;
    } else {
        if (c->k == 0) return -4;
        if (set_reserve(s, nbytes / ((size_t)c->k + 1) + 16)) return -3;
    }
    while (pos < nbytes) {
        const char *nl = (const char *)memchr(text + pos, '\n', nbytes - pos);
        if (!nl) break; /* unterminated last piece: eof() => dropped */
        size_t l = (size_t)(nl - (text + pos));
        if (l != (size_t)c->k) return -5; /* reference: assert(str.size()==overlap) */
        if (set_insert(s, ho_canon_str(text + pos, c->k)) < 0) return -3;
        total++;
        pos += l + 1;
    }
    c->lines[hap] = total;
    return 0;
}

static char *slurp(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)sz + 1);
    if (buf && sz > 0 && fread(buf, 1, (size_t)sz, f) != (size_t)sz) {
        free(buf);
        buf = NULL;
    }
    fclose(f);
    *n = (size_t)sz;
    return buf;
}

int ho_load_kmers_file(ho_classifier *c, const char *path, int hap) {
    size_t n = 0;
    char *buf = slurp(path, &n);
    if (!buf) return -1; /* the reference spins forever here (classify.cpp:41 tests only eof) */
    int rc = ho_load_kmers_text(c, buf, n, hap);
    free(buf);
    return rc;
}

int ho_load_keys(ho_classifier *c, const uint64_t *keys, size_t n, int hap, int k) {
    if (k < 1 || k > 32) return -2;
    if (hap == 0) c->k = k;
    else if (c->k != k) return -4;
    ho_set *s = &c->set[hap];
    if (set_reserve(s, n + 16)) return -3;
    for (size_t i = 0; i < n; i++) set_insert_nogrow(s, keys[i]);
    c->lines[hap] = n;
    return 0;
}

/* Multi-threaded bulk build for bench.py's cpu_baseline on full-size (2 x 200M key) sets: same
 * exact set as ho_load_keys (insert-if-absent via compare-and-swap), only built by `threads`
 * workers so the untimed set-up stays short.  The reference builds single-threaded
 * (classify.cpp:30-46); build time is not part of any reported number. */
typedef struct {
    ho_set *s;
    const uint64_t *keys;
    size_t lo, hi;
    uint64_t added;
} build_job;

static void *build_worker(void *arg) {
    build_job *j = (build_job *)arg;
    ho_set *s = j->s;
    const uint64_t m = s->cap - 1;
    uint64_t added = 0;
    for (size_t r = j->lo; r < j->hi; r++) {
        const uint64_t key = j->keys[r];
        uint64_t i = mix64(key) & m;
        for (;;) {
            uint64_t v = __atomic_load_n(&s->slot[i], __ATOMIC_RELAXED);
            if (v == key) break;
            if (v == HO_EMPTY) {
                uint64_t expect = HO_EMPTY;
                if (__atomic_compare_exchange_n(&s->slot[i], &expect, key, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                    added++;
                    break;
                }
                if (expect == key) break;
            }
            i = (i + 1) & m;
        }
    }
    j->added = added;
    return NULL;
}

int ho_load_keys_mt(ho_classifier *c, const uint64_t *keys, size_t n, int hap, int k, int threads) {
    if (k < 1 || k > 32) return -2;
    if (hap == 0) c->k = k;
    else if (c->k != k) return -4;
    if (threads < 1) threads = 1;
    ho_set *s = &c->set[hap];
    if (set_reserve(s, n + 16)) return -3;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    build_job *jobs = (build_job *)malloc(sizeof(build_job) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (build_job){s, keys, n * (size_t)t / (size_t)threads, n * (size_t)(t + 1) / (size_t)threads, 0};
        pthread_create(&th[t], NULL, build_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        s->size += jobs[t].added;
    }
    free(th);
    free(jobs);
    c->lines[hap] = n;
    return 0;
}

/* classify.cpp:314-339 InitAdaptor.  (An adaptor shorter than K aborts the reference via
 * kmer.h:171; here it simply contributes nothing.) */
int ho_init_adaptor(ho_classifier *c, const char *af, const char *ar, FILE *log) {
    int erased = 0;
    const char *ad[2] = {af, ar};
    if (log) {
        fprintf(log, "Adaptor forward :%s\n", af);
        fprintf(log, "Adaptor reverse :%s\n", ar);
    }
    for (int a = 0; a < 2; a++) {
        size_t len = strlen(ad[a]);
        if (len < (size_t)c->k) continue;
        uint64_t *km = (uint64_t *)malloc((len - (size_t)c->k + 1) * sizeof(uint64_t));
        size_t n = ho_chop_read(ad[a], len, c->k, km);
        for (size_t i = 0; i < n; i++)
            for (int h = 0; h < 2; h++)
                if (set_erase(&c->set[h], km[i])) {
                    erased++;
                    if (log) {
                        char buf[40];
                        ho_kmer_to_str(km[i], c->k, buf);
                        fprintf(log, " INFO : erase a adaptor kmer from hap %d ; kmer= %s\n", h, buf);
                    }
                }
        free(km);
    }
    return erased;
}

/* classify.cpp:182-185 containN + :194-202 the vote loop. */
void ho_read_votes(const ho_classifier *c, const char *seq, size_t slen,
                   uint32_t *v0, uint32_t *v1, int *has_n) {
    *v0 = *v1 = 0;
    *has_n = memchr(seq, 'N', slen) != NULL;
    if (*has_n || slen < (size_t)c->k) return;
    const int k = c->k;
    const uint64_t mask = ho_mask(k);
    uint64_t w = ho_pack(seq, k), b = ho_revcomp(w, k);
    for (size_t index = 0;; index++) {
        uint64_t key = (w < b) ? w : b;
        *v0 += (uint32_t)set_contains(&c->set[0], key);
        *v1 += (uint32_t)set_contains(&c->set[1], key);
        if (index + (size_t)k >= slen) break;
        uint8_t ch = ho_base2int((uint8_t)seq[index + k]);
        w = ((w << 2) & mask) | ch;
        b = (b >> 2) | ((uint64_t)(ch ^ 2) << (2 * (k - 1)));
    }
}

/* classify.cpp:186-209 process_reads. */
int ho_process_read(ho_classifier *c, const char *head, size_t hlen, const char *seq, size_t slen) {
    size_t bs, bn;
    ho_parse_name(head, hlen, &bs, &bn);
    int has_n = memchr(seq, 'N', slen) != NULL;                 /* :190 */
    if (!has_n && slen < (size_t)c->k) return -1;               /* reference aborts, kmer.h:171 */
    bc_entry *e = bc_get(&c->barcodes, head + bs, bn);
    if (has_n) {
        e->cneg += 1;                                           /* :191 */
        return 0;
    }
    uint32_t v0, v1;
    ho_read_votes(c, seq, slen, &v0, &v1, &has_n);
    if (v0 > 0) e->c0 += (int32_t)v0;                           /* :203-204 */
    if (v1 > 0) e->c1 += (int32_t)v1;                           /* :205-206 */
    if (v0 == 0 && v1 == 0) e->cneg += 1;                       /* :207-208 */
    return 0;
}

/* --- line reader reproducing `std::getline(...).eof()` over plain or gz input ------------- */
typedef struct {
    FILE *fp;
    gzFile gz;
    char *buf;
    size_t cap, len, pos;
    int at_eof;
} lreader;

static int lr_fill(lreader *r) {
    if (r->at_eof) return 0;
    if (r->pos > 0) {
        memmove(r->buf, r->buf + r->pos, r->len - r->pos);
        r->len -= r->pos;
        r->pos = 0;
    }
    if (r->len == r->cap) {
        r->cap *= 2;
        r->buf = (char *)realloc(r->buf, r->cap);
    }
    size_t want = r->cap - r->len;
    long got = r->gz ? gzread(r->gz, r->buf + r->len, (unsigned)(want > (1u << 30) ? (1u << 30) : want))
                     : (long)fread(r->buf + r->len, 1, want, r->fp);
    if (got <= 0) {
        r->at_eof = 1;
        return 0;
    }
    r->len += (size_t)got;
    return 1;
}

/* Returns pointer/len of the next line; *hit_eof = 1 iff EOF was reached before a '\n'
 * (what std::getline reports through eofbit). */
static const char *lr_getline(lreader *r, size_t *n, int *hit_eof) {
    size_t scanned = 0; /* bytes after r->pos already known to hold no '\n' */
    for (;;) {
        const char *line = r->buf + r->pos;
        char *nl = (char *)memchr(line + scanned, '\n', r->len - r->pos - scanned);
        if (nl) {
            *n = (size_t)(nl - line);
            r->pos += *n + 1;
            *hit_eof = 0;
            return line;
        }
        scanned = r->len - r->pos;
        if (!lr_fill(r)) { /* lr_fill may move the data to the front; r->pos is updated */
            line = r->buf + r->pos;
            *n = r->len - r->pos;
            r->pos = r->len;
            *hit_eof = 1;
            return line;
        }
    }
}

/* classify.cpp:238-278 processFastq: .gz decided by the file-name suffix (:245-254); records
 * are 4 getlines, header must be newline-terminated (:257), lines 3-4 ignored (:267-268). */
int ho_process_fastq(ho_classifier *c, const char *path) {
    lreader r;
    memset(&r, 0, sizeof(r));
    size_t plen = strlen(path);
    int gz = plen > 3 && strcmp(path + plen - 3, ".gz") == 0;
    if (gz) {
        r.gz = gzopen(path, "rb");
        if (!r.gz) return -1;
        gzbuffer(r.gz, 1 << 20);
    } else {
        r.fp = fopen(path, "rb");
        if (!r.fp) return -1;
    }
    r.cap = 1 << 20;
    r.buf = (char *)malloc(r.cap);
    int rc = 0;
    char *head = NULL;
    size_t hcap = 0;
    for (;;) {
        size_t hn, sn, tn;
        int eof;
        const char *h = lr_getline(&r, &hn, &eof);
        if (eof) break;
        if (hn + 1 > hcap) {
            hcap = (hn + 1) * 2;
            head = (char *)realloc(head, hcap);
        }
        memcpy(head, h, hn); /* the next getline may move the buffer */
        const char *s = lr_getline(&r, &sn, &eof);
        if (ho_process_read(c, head, hn, s, sn) < 0) {
            rc = -6;
            break;
        }
        lr_getline(&r, &tn, &eof);
        lr_getline(&r, &tn, &eof);
    }
    free(head);
    free(r.buf);
    if (r.gz) gzclose(r.gz);
    if (r.fp) fclose(r.fp);
    return rc;
}

static int bc_cmp(const void *a, const void *b) {
    const bc_entry *x = (const bc_entry *)a, *y = (const bc_entry *)b;
    size_t m = x->len < y->len ? x->len : y->len;
    int r = memcmp(x->name, y->name, m); /* std::string::compare: traits compare, then length */
    if (r) return r;
    return (x->len > y->len) - (x->len < y->len);
}

/* classify.cpp:93-102 printBarcodeInfos. */
int ho_print(ho_classifier *c, FILE *out) {
    bc_map *m = &c->barcodes;
    qsort(m->ent, m->n, sizeof(bc_entry), bc_cmp);
    bc_rehash(m, m->icap ? m->icap : 1024);
    for (size_t i = 0; i < m->n; i++) {
        bc_entry *e = &m->ent[i];
        int hap = ho_get_hap(e->name, e->len, e->c0, e->c1, c->set[0].size, c->set[1].size, c->w0, c->w1);
        fwrite(e->name, 1, e->len, out);
        fprintf(out, "\t%d\t%d\t%d\n", hap, e->c0, e->c1);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * id-keyed bulk path
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const ho_classifier *c;
    const uint8_t *bases;
    const uint64_t *off;
    const uint32_t *ids;
    size_t lo, hi;
    uint32_t *c0, *c1, *neg, *seen;
    uint32_t *votes;   /* optional [n_reads][2]: the read's own (vote0, vote1), classify.cpp:195-202 */
} bulk_job;

static void *bulk_worker(void *arg) {
    bulk_job *j = (bulk_job *)arg;
    for (size_t r = j->lo; r < j->hi; r++) {
        const char *seq = (const char *)j->bases + j->off[r];
        size_t slen = (size_t)(j->off[r + 1] - j->off[r]);
        uint32_t v0, v1, id = j->ids[r];
        int has_n;
        ho_read_votes(j->c, seq, slen, &v0, &v1, &has_n);
        if (j->votes) {
            j->votes[2 * r] = v0;
            j->votes[2 * r + 1] = v1;
        }
        if (j->seen) __atomic_fetch_add(&j->seen[id], 1u, __ATOMIC_RELAXED);
        if (v0) __atomic_fetch_add(&j->c0[id], v0, __ATOMIC_RELAXED);
        if (v1) __atomic_fetch_add(&j->c1[id], v1, __ATOMIC_RELAXED);
        if (!v0 && !v1 && j->neg) __atomic_fetch_add(&j->neg[id], 1u, __ATOMIC_RELAXED);
    }
    return NULL;
}

int ho_classify_ids(const ho_classifier *c, const uint8_t *bases, const uint64_t *offsets,
                    const uint32_t *ids, size_t n_reads, uint32_t *c0, uint32_t *c1,
                    uint32_t *neg, uint32_t *seen, int threads) {
    return ho_classify_ids_votes(c, bases, offsets, ids, n_reads, c0, c1, neg, seen, NULL, threads);
}

int ho_classify_ids_votes(const ho_classifier *c, const uint8_t *bases, const uint64_t *offsets,
                          const uint32_t *ids, size_t n_reads, uint32_t *c0, uint32_t *c1,
                          uint32_t *neg, uint32_t *seen, uint32_t *votes, int threads) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > n_reads) threads = n_reads ? (int)n_reads : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    bulk_job *jobs = (bulk_job *)malloc(sizeof(bulk_job) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (bulk_job){c, bases, offsets, ids, n_reads * (size_t)t / (size_t)threads,
                             n_reads * (size_t)(t + 1) / (size_t)threads, c0, c1, neg, seen, votes};
        if (threads == 1) bulk_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, bulk_worker, &jobs[t]);
    }
    if (threads > 1)
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
    return 0;
}

/* ==========================================================================================
 * Secondary oracle: stage-03 per-read classifier (03.mkoutput_by_fabulous2.0/src_main/classify.cpp,
 * cited s03:N).  Strings, not 2-bit codes -- restated as such.
 * ======================================================================================== */
typedef struct {
    char *arena;      /* n strings of k bytes back to back */
    size_t n, cap;
    uint32_t *index;  /* open addressing: idx+1 */
    size_t icap;
} strset;

struct ho_s03 {
    int k;
    strset set[2];
    uint64_t lines[2];
    unsigned char oppo[256]; /* g_oppo s03:24-36; absent keys default-construct to '\0' (std::map::operator[]) */
};

static int strset_find_or_add(strset *s, const char *key, int k, int add) {
    if (s->icap == 0) {
        s->icap = 1 << 16;
        s->index = (uint32_t *)calloc(s->icap, sizeof(uint32_t));
    }
    size_t i = hash_bytes(key, (size_t)k) & (s->icap - 1);
    while (s->index[i]) {
        if (memcmp(s->arena + (size_t)(s->index[i] - 1) * (size_t)k, key, (size_t)k) == 0) return 1;
        i = (i + 1) & (s->icap - 1);
    }
    if (!add) return 0;
    if (s->n == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 4096;
        s->arena = (char *)realloc(s->arena, s->cap * (size_t)k);
    }
    memcpy(s->arena + s->n * (size_t)k, key, (size_t)k);
    s->n++;
    s->index[i] = (uint32_t)s->n;
    if (s->n * 2 > s->icap) {
        size_t ncap = s->icap * 2;
        uint32_t *ni = (uint32_t *)calloc(ncap, sizeof(uint32_t));
        for (size_t e = 0; e < s->n; e++) {
            size_t j = hash_bytes(s->arena + e * (size_t)k, (size_t)k) & (ncap - 1);
            while (ni[j]) j = (j + 1) & (ncap - 1);
            ni[j] = (uint32_t)(e + 1);
        }
        free(s->index);
        s->index = ni;
        s->icap = ncap;
    }
    return 0;
}

ho_s03 *ho_s03_new(void) {
    ho_s03 *c = (ho_s03 *)calloc(1, sizeof(*c));
    const char *from = "aAgGcCtTnN", *to = "TTCCGGAANN"; /* s03:26-35 */
    for (int i = 0; from[i]; i++) c->oppo[(unsigned char)from[i]] = (unsigned char)to[i];
    return c;
}
void ho_s03_free(ho_s03 *c) {
    if (!c) return;
    for (int h = 0; h < 2; h++) {
        free(c->set[h].arena);
        free(c->set[h].index);
    }
    free(c);
}
int ho_s03_k(const ho_s03 *c) { return c->k; }
uint64_t ho_s03_lines(const ho_s03 *c, int hap) { return c->lines[hap & 1]; }

static void s03_insert_line(ho_s03 *c, int hap, const char *line, size_t len) {
    /* s03:59-60 / 64-65: insert(line); insert(reverse_complement(line)) -- strings of the LINE's length.
     * Only lines of exactly K bytes can ever match a K-byte window, so others are counted but not stored. */
    if ((int)len != c->k) return;
    char rc[64];
    for (size_t i = 0; i < len; i++) rc[len - i - 1] = (char)c->oppo[(unsigned char)line[i]]; /* s03:37-42 */
    strset_find_or_add(&c->set[hap], line, c->k, 1);
    strset_find_or_add(&c->set[hap], rc, c->k, 1);
}

int ho_s03_load_text(ho_s03 *c, const char *text, size_t nbytes, int hap) { /* s03:51-70 */
    size_t pos = 0;
    uint64_t total = 0;
    if (hap == 0) {
        const char *nl = (const char *)memchr(text, '\n', nbytes);
        size_t l = nl ? (size_t)(nl - text) : nbytes;
        if (l < 1 || l > 63) return -2;
        c->k = (int)l; /* s03:58 */
        s03_insert_line(c, hap, text, l);
        total++;
        pos = nl ? l + 1 : nbytes;
    }
    while (pos < nbytes) {
        const char *nl = (const char *)memchr(text + pos, '\n', nbytes - pos);
        if (!nl) break; /* s03:63 eof() => dropped */
        s03_insert_line(c, hap, text + pos, (size_t)(nl - (text + pos)));
        total++;
        pos = (size_t)(nl - text) + 1;
    }
    c->lines[hap] = total; /* s03:68 */
    return 0;
}

int ho_s03_load_file(ho_s03 *c, const char *path, int hap) {
    size_t n = 0;
    char *buf = slurp(path, &n);
    if (!buf) return -1;
    int rc = ho_s03_load_text(c, buf, n, hap);
    free(buf);
    return rc;
}

void ho_s03_read_hits(const ho_s03 *c, const char *seq, size_t slen, uint32_t *h0, uint32_t *h1) { /* s03:209-214 */
    *h0 = *h1 = 0;
    if ((long)slen - c->k + 1 <= 0) return;
    for (size_t i = 0; i + (size_t)c->k <= slen; i++) {
        *h0 += (uint32_t)strset_find_or_add((strset *)&c->set[0], seq + i, c->k, 0);
        *h1 += (uint32_t)strset_find_or_add((strset *)&c->set[1], seq + i, c->k, 0);
    }
}

int ho_s03_format_row(const ho_s03 *c, const char *name, size_t nlen, uint32_t h0, uint32_t h1, char *out) {
    /* s03:215-216 then s03:104-135 */
    double hc[2] = {(double)h0, (double)h1};
    for (int j = 0; j < 2; j++) hc[j] /= (double)(int)c->lines[j]; /* total_kmers is int (s03:50) */
    double readHapCount = 0, secondBest = 0;
    int readHap = -1;
    for (int i = 0; i < 2; i++) {
        if (hc[i] > 0 && hc[i] < readHapCount && hc[i] > secondBest) secondBest = hc[i];
        if (hc[i] > 0 && hc[i] > readHapCount) {
            readHap = i;
            secondBest = readHapCount;
            readHapCount = hc[i];
        }
    }
    memcpy(out, name, nlen);
    char *p = out + nlen;
    if (secondBest == 0 && readHapCount != 0) return (int)nlen + sprintf(p, "\thaplotype%d\t%0.6f\n", readHap, readHapCount);
    if (readHapCount == 0 && secondBest == 0) return (int)nlen + sprintf(p, "\tambiguous\t0.0\n");
    if (readHapCount / secondBest > 1) return (int)nlen + sprintf(p, "\thaplotype%d\t%0.6f\n", readHap, readHapCount);
    return (int)nlen + sprintf(p, "\tambiguous\t%0.6f\n", readHapCount);
}

static void s03_emit(const ho_s03 *c, const char *head, size_t hlen, const char *seq, size_t slen, FILE *out) {
    uint32_t h0, h1;
    ho_s03_read_hits(c, seq, slen, &h0, &h1);
    char *row = (char *)malloc(hlen + 96);
    /* name = head.substr(1) (s03:207); an empty header would throw in the reference */
    int n = ho_s03_format_row(c, hlen ? head + 1 : head, hlen ? hlen - 1 : 0, h0, h1, row);
    fwrite(row, 1, (size_t)n, out);
    free(row);
}

int ho_s03_process_file(const ho_s03 *c, const char *path, int format, FILE *out) {
    lreader r;
    memset(&r, 0, sizeof(r));
    size_t plen = strlen(path);
    if (plen > 3 && strcmp(path + plen - 3, ".gz") == 0) { /* s03:235-242 */
        r.gz = gzopen(path, "rb");
        if (!r.gz) return -1;
    } else {
        r.fp = fopen(path, "rb");
        if (!r.fp) return -1;
    }
    r.cap = 1 << 20;
    r.buf = (char *)malloc(r.cap);
    char *head = NULL, *seq = NULL;
    size_t hcap = 0, hlen = 0, scap = 0, slen = 0;
    int rc = 0;
    if (format == 1) { /* processFastq s03:248-270: ids from 0, one record = 4 getlines */
        for (;;) {
            size_t n, sn, tn;
            int eof;
            const char *h = lr_getline(&r, &n, &eof);
            if (eof) break;
            if (n && h[0] == '>') { rc = -7; break; } /* s03:256-259 */
            if (n + 1 > hcap) { hcap = (n + 1) * 2; head = (char *)realloc(head, hcap); }
            memcpy(head, h, n);
            const char *s = lr_getline(&r, &sn, &eof);
            s03_emit(c, head, n, s, sn, out);
            lr_getline(&r, &tn, &eof);
            lr_getline(&r, &tn, &eof);
        }
    } else { /* processFasta s03:272-302: multi-line sequences, ids from 1, empty lines skipped */
        long long id = 0;
        for (;;) {
            size_t n;
            int eof;
            const char *t = lr_getline(&r, &n, &eof);
            if (eof) break; /* s03:279: a last line without '\n' is dropped */
            if (n == 0) continue;
            if (t[0] == '@' || t[0] == '+') { rc = -7; break; }
            if (t[0] == '>') {
                if (id > 0) s03_emit(c, head, hlen, seq ? seq : "", slen, out);
                if (n + 1 > hcap) { hcap = (n + 1) * 2; head = (char *)realloc(head, hcap); }
                memcpy(head, t, n);
                hlen = n;
                slen = 0;
                id++;
            } else {
                if (slen + n + 1 > scap) { scap = (slen + n + 1) * 2; seq = (char *)realloc(seq, scap); }
                memcpy(seq + slen, t, n);
                slen += n;
            }
        }
        if (rc == 0) s03_emit(c, head ? head : "", hlen, seq ? seq : "", slen, out); /* s03:297 unconditional submit */
    }
    free(head);
    free(seq);
    free(r.buf);
    if (r.gz) gzclose(r.gz);
    if (r.fp) fclose(r.fp);
    return rc;
}

/*
 * oracle_main_s00.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Front end of the stage-00 restatement, mirroring the reference script's options and products
 * (00.build_unshare_kmers_by_jellyfish/build_unshared_kmers.sh, cited s00.sh:N): writes
 * paternal.unique.filter.mer / maternal.unique.filter.mer (sorted) into the working directory and, with
 * --auto_bounds, {maternal,paternal}.histo and {maternal,paternal}.bounds.txt (ana.sh:7-13).
 */
#include "s00_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int ends_gz(const char *s) {
    size_t n = strlen(s);
    return n >= 3 && strcmp(s + n - 3, ".gz") == 0;            /* s00.sh:169 ${fname: -3} */
}
/* s00.sh:166-185: all files of one parent must be gz or all plain; returns 1 gz, 0 plain, -1 mixed */
static int gz_mode(const char **files, int n) {
    int gz = ends_gz(files[0]);
    for (int i = 1; i < n; i++)
        if (ends_gz(files[i]) != gz) return -1;
    return gz;
}
static int write_set(const ho_s00 *o, int parent, long lo, long hi, const char *path) {
    size_t n = ho_s00_select(o, parent, lo, hi, NULL);
    uint64_t *keys = malloc((n ? n : 1) * sizeof *keys);
    ho_s00_select(o, parent, lo, hi, keys);
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    char s[40];
    for (size_t i = 0; i < n; i++) {
        ho_s00_key_to_str(keys[i], ho_s00_k(o), s);
        fprintf(f, "%s\n", s);
    }
    fclose(f);
    free(keys);
    printf("%zu %s\n", n, path);                                /* s00.sh:300-303 wc -l */
    return 0;
}

int main(int argc, char **argv) {
    /* s00.sh:44-55 defaults; parents: 0 = paternal, 1 = maternal */
    long mer = 21, cpu = 8, memory = 10, lower[2] = {9, 9}, upper[2] = {33, 33};
    const char *files[2][1024];
    int nf[2] = {0, 0}, auto_bounds = 0;
    if (argc == 1) { fprintf(stderr, "usage\n"); return 0; }   /* s00.sh:60-63 */
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        const char *v = i + 1 < argc ? argv[i + 1] : "";
        if (!strcmp(a, "-h") || !strcmp(a, "--help")) { fprintf(stderr, "usage\n"); return 0; }
        else if (!strcmp(a, "--memory")) { memory = atol(v); i++; }
        else if (!strcmp(a, "--thread")) { cpu = atol(v); i++; }
        else if (!strcmp(a, "--m-lower")) { lower[1] = atol(v); i++; }
        else if (!strcmp(a, "--m-upper")) { upper[1] = atol(v); i++; }
        else if (!strcmp(a, "--p-lower")) { lower[0] = atol(v); i++; }
        else if (!strcmp(a, "--p-upper")) { upper[0] = atol(v); i++; }
        else if (!strcmp(a, "--mer")) { mer = atol(v); i++; }
        else if (!strcmp(a, "--auto_bounds")) auto_bounds = 1;
        else if (!strcmp(a, "--paternal")) { if (nf[0] < 1024) files[0][nf[0]++] = v; i++; }
        else if (!strcmp(a, "--maternal")) { if (nf[1] < 1024) files[1][nf[1]++] = v; i++; }
        else { printf("unknown option \"%s\"\n", a); return 0; }   /* s00.sh:113-116: message, then a bare `exit` = status 0 */
    }
    /* s00.sh:141-152 sanity check */
    if (memory < 1 || cpu < 1 || !nf[0] || !nf[1] || mer < 11 || lower[1] < 1 || upper[1] > 100000000 || lower[0] < 1 ||
        upper[0] > 100000000) {
        printf("ERROR: invalid arguments\n");
        return 1;
    }
    if (mer > 32) { fprintf(stderr, "mer > 32 is not supported by this restatement\n"); return 1; }
    for (int p = 1; p >= 0; p--)                                /* s00.sh:153-158 */
        for (int i = 0; i < nf[p]; i++)
            if (access(files[p][i], F_OK)) {
                printf("ERROR: input file \"%s\" does not exist\n", files[p][i]);
                return 1;
            }
    ho_s00 *o = ho_s00_new((int)mer);
    for (int p = 1; p >= 0; p--) {                              /* maternal first (s00.sh:165-225) */
        /* the script prepends each new file to the list (s00.sh:105,109): stream order = reverse argument order */
        const char *rev[1024];
        for (int i = 0; i < nf[p]; i++) rev[i] = files[p][nf[p] - 1 - i];
        int gz = gz_mode(rev, nf[p]);
        if (gz < 0) { printf("ERROR: gz and plain inputs mixed for one parent\n"); return 1; }
        int rc = ho_s00_add_files(o, p, rev, nf[p], gz);
        if (rc) { fprintf(stderr, "counting failed (%d)\n", rc); return 1; }
    }
    if (auto_bounds) {                                          /* s00.sh:241-252, ana.sh:7-13 */
        static uint64_t h[HO_S00_HISTO_HIGH + 2];
        const char *name[2] = {"paternal", "maternal"};
        for (int p = 1; p >= 0; p--) {
            char path[64];
            long mn, mx;
            ho_s00_histo(o, p, h);
            snprintf(path, sizeof path, "%s.histo", name[p]);
            FILE *f = fopen(path, "w");
            if (!f || ho_s00_write_histo(h, f)) return 1;
            fclose(f);
            ho_s00_find_bounds(h, &mn, &mx, &lower[p], &upper[p]);
            snprintf(path, sizeof path, "%s.bounds.txt", name[p]);
            f = fopen(path, "w");
            if (!f) return 1;
            fprintf(f, "MIN_INDEX=%ld\nMAX_INDEX=%ld\nLOWER_INDEX=%ld\nUPPER_INDEX=%ld\n", mn, mx, lower[p], upper[p]);
            fclose(f);
        }
    }
    printf("bounds used for maternal: [%ld, %ld]\n", lower[1], upper[1]);
    printf("bounds used for paternal: [%ld, %ld]\n", lower[0], upper[0]);
    if (write_set(o, 0, lower[0], upper[0], "paternal.unique.filter.mer")) return 1;
    if (write_set(o, 1, lower[1], upper[1], "maternal.unique.filter.mer")) return 1;
    ho_s00_free(o);
    return 0;
}

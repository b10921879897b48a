/*
 * oracle_main_s03.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Front end of the stage-03 per-read classifier restatement, mirroring the reference's main()
 * (03.mkoutput_by_fabulous2.0/src_main/classify.cpp:314-378): --hap F --hap F --read F [--format fasta|fastq].
 */
#include "hast_oracle.h"
#include <getopt.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
    static struct option lo[] = {{"hap", required_argument, NULL, 'p'},    {"read", required_argument, NULL, 'r'},
                                 {"format", required_argument, NULL, 'f'}, {"thread", required_argument, NULL, 't'},
                                 {"help", no_argument, NULL, 'h'},         {0, 0, 0, 0}};
    const char *haps[8], *reads[4096], *format = "fasta";
    int nh = 0, nr = 0, t = 8;
    for (;;) {
        int c = getopt_long(argc, argv, "p:r:t:f:h", lo, NULL);
        if (c < 0) break;
        switch (c) {
        case 'p': if (nh < 8) haps[nh++] = optarg; break;
        case 'r': if (nr < 4096) reads[nr++] = optarg; break;
        case 'f': format = optarg; break;
        case 't': t = atoi(optarg); break;
        default: fprintf(stderr, "usage\n"); return -1;
        }
    }
    if (nh != 2 || nr == 0 || t < 1) { fprintf(stderr, "usage\n"); return -1; }
    if (strcmp(format, "fasta") && strcmp(format, "fastq")) return -1;
    ho_s03 *c = ho_s03_new();
    for (int h = 0; h < 2; h++)
        if (ho_s03_load_file(c, haps[h], h)) { fprintf(stderr, "load failed\n"); return 2; }
    for (int i = 0; i < nr; i++)
        if (ho_s03_process_file(c, reads[i], strcmp(format, "fastq") == 0, stdout)) return 3;
    ho_s03_free(c);
    return 0;
}

/*
 * oracle_main.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Command-line front end of the CPU restatement, mirroring the reference's main()
 * (classify.cpp:373-450): same flags (:375-387), same phase order (:429-449), same stdout.
 * Used by tests to diff against golden outputs of the real reference binary.
 */
#include "hast_oracle.h"
#include <getopt.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
    static struct option lo[] = {
        {"hap0", required_argument, NULL, 'p'},      {"hap1", required_argument, NULL, 'm'},
        {"read", required_argument, NULL, 'r'},      {"thread", required_argument, NULL, 't'},
        {"weight0", required_argument, NULL, 'w'},   {"weight1", required_argument, NULL, 'u'},
        {"adaptor_f", required_argument, NULL, 'f'}, {"adaptor_r", required_argument, NULL, 'q'},
        {"help", no_argument, NULL, 'h'},            {0, 0, 0, 0}};
    const char *hap0 = NULL, *hap1 = NULL;
    const char *af = "CTGTCTCTTATACACATCTTAGGAAGACAAGCACTGACGACATGA"; /* classify.cpp:312 */
    const char *ar = "TCTGCTGAGTCGAGAACGTCTCTGTGAGCCAAGGAGTTGCTCTGG"; /* classify.cpp:313 */
    const char *reads[4096];
    int n_reads = 0, t = 8;
    double w0 = 1.0, w1 = 1.0;
    for (;;) {
        int c = getopt_long(argc, argv, "p:m:l:r:t:w:u:f:q:h", lo, NULL);
        if (c < 0) break;
        switch (c) {
        case 'f': af = optarg; break;
        case 'q': ar = optarg; break;
        case 'p': hap0 = optarg; break;
        case 'm': hap1 = optarg; break;
        case 'r': if (n_reads < 4096) reads[n_reads++] = optarg; break;
        case 't': t = atoi(optarg); break;
        case 'u': w1 = atof(optarg); break;
        case 'w': w0 = atof(optarg); break;
        default: fprintf(stderr, "usage: oracle_classify --hap0 F --hap1 F --read F [...]\n"); return -1;
        }
    }
    if (!hap0 || !hap1 || n_reads == 0 || t < 1) {
        fprintf(stderr, "usage: oracle_classify --hap0 F --hap1 F --read F [...]\n");
        return -1;
    }
    ho_classifier *c = ho_new();
    ho_set_weights(c, w0, w1);
    fprintf(stderr, "__START__\n");
    int rc;
    if ((rc = ho_load_kmers_file(c, hap0, 0)) != 0) { fprintf(stderr, "load hap0 failed (%d)\n", rc); return 2; }
    fprintf(stderr, "Recorded %llu haplotype 0 specific %d-mers\n", (unsigned long long)ho_lines_loaded(c, 0), ho_k(c));
    if ((rc = ho_load_kmers_file(c, hap1, 1)) != 0) { fprintf(stderr, "load hap1 failed (%d)\n", rc); return 2; }
    fprintf(stderr, "Recorded %llu haplotype 1 specific %d-mers\n", (unsigned long long)ho_lines_loaded(c, 1), ho_k(c));
    ho_init_adaptor(c, af, ar, stderr);
    for (int i = 0; i < n_reads; i++) {
        fprintf(stderr, "__process read: %s\n", reads[i]);
        if ((rc = ho_process_fastq(c, reads[i])) != 0) { fprintf(stderr, "read file failed (%d)\n", rc); return 3; }
    }
    ho_print(c, stdout);
    fprintf(stderr, "__END__\n");
    ho_free(c);
    return 0;
}

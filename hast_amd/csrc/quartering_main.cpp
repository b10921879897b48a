// quartering_main.cpp -- multi-threaded replacement for the step right after `classify` in HAST stage 01
// (SURVEY 8(f) #2): routing every FASTQ record to <prefix>.{paternal,maternal,homozygous,nobarcode}.fastq by the
// barcode lists, which the reference does with single-threaded awk
// (/root/reference/01.classify_stlfr_reads/quartering_fastq.awk, invoked at classify_stlfr_reads.sh:176-185 as
//   awk -v prefix=NAME -F '#|/' -f quartering_fastq.awk paternal.unique.barcodes maternal.unique.barcodes
//       homozygous.unique.barcodes READS   (READS may be "-" behind `gzip -dc`)).
//
//   quartering_fastq [-t N] --prefix NAME paternal.barcodes maternal.barcodes homozygous.barcodes READS|-|READS.gz
//
// (the routing itself: quartering.h, shared with `classify`)
// Same outputs byte for byte: the four FASTQ files (created only when something is routed to them), the
// "ERROR : unclassify barcode" lines on stderr (those records are dropped, awk :31-34) and the lines appended
// to filter_reads.log (awk :18-20, :51-57).  Semantics restated from the awk program: fields are split at '#'
// or '/' (-F '#|/'); a list line contributes its first field (:12-16); a header line (every 4th line from the
// first, :21) with more than one field and a second field other than "0_0_0" is looked up in the paternal, then
// maternal, then homozygous list (:23-35), otherwise it is a no-barcode read (:36-39); all four lines follow the
// header's class (:41-49).  Host-only code (no GPU work in this step); .gz input is inflated in-process.
#include <getopt.h>

#include "quartering.h"

int main(int argc, char **argv) {
    static struct option lo[] = {{"prefix", required_argument, NULL, 'p'}, {"thread", required_argument, NULL, 't'},
                                 {"block-mb", required_argument, NULL, 'b'}, {0, 0, 0, 0}};
    std::string prefix;
    int t_num = 8;
    size_t block_mb = 64;
    for (;;) {
        int c = getopt_long(argc, argv, "p:t:b:", lo, NULL);
        if (c < 0) break;
        if (c == 'p') prefix = optarg;
        else if (c == 't') t_num = atoi(optarg);
        else if (c == 'b') block_mb = (size_t)std::max(1L, atol(optarg));
        else return 2;
    }
    if (argc - optind != 4 || t_num < 1) {
        fprintf(stderr, "usage: quartering_fastq [-t N] --prefix NAME paternal.barcodes maternal.barcodes homozygous.barcodes READS|-\n");
        return 2;
    }
    hast::quartering::ClassMap cls_of;
    for (int i = 0; i < 3; i++)
        if (!hast::quartering::load_list(argv[optind + i], (uint8_t)(i + 1), cls_of)) {
            fprintf(stderr, "quartering_fastq: cannot read %s\n", argv[optind + i]);
            return 2;
        }
    const std::string reads = argv[optind + 3];
    hast::BlockSource src;
    if (!src.open(reads, block_mb << 20)) {
        fprintf(stderr, "quartering_fastq: cannot open %s\n", reads.c_str());
        return 2;
    }
    return hast::quartering::route(prefix, cls_of, src, reads, t_num, "quartering_fastq");
}

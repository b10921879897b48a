# the drop-in CLI on 20M reads (2 x 3.4 GB of plain FASTQ): GPU framing against --host-parse, block sizes, thread counts, gz; stdout md5 in every run
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
NP=10000000
tools/gen_fastq $D $NP 5000000 100000 21 150 64 0 || exit 1
ls -la $D | head
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq --read $D/r2.fq --weight0 1.04"
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats__ $D/err.$name | sed "s/.*load_s/load_s/") $(grep -h __stats_read_phase__ $D/err.$name | cut -d" " -f2-)"; }
cat $D/r1.fq $D/r2.fq > /dev/null
for i in 1; do
run fq_t32 hast_amd/classify $ARGS -t 32 --stats
run fq_t64 hast_amd/classify $ARGS -t 64 --stats
run fq_t8 hast_amd/classify $ARGS -t 8 --stats
run host_t32 hast_amd/classify $ARGS -t 32 --stats --host-parse
done
run fq_mb8 hast_amd/classify $ARGS -t 32 --stats --block-mb 8
run fq_mb32 hast_amd/classify $ARGS -t 32 --stats --block-mb 32
run fq_mb16 hast_amd/classify $ARGS -t 32 --stats --block-mb 16

(gzip -1 -k $D/r1.fq & gzip -1 -k $D/r2.fq & wait)
GZARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq.gz --read $D/r2.fq.gz --weight0 1.04"
run fq_gz_t32 hast_amd/classify $GZARGS -t 32 --stats
run host_gz_t32 hast_amd/classify $GZARGS -t 32 --stats --host-parse
rm -rf $D

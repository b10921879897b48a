# A/B on ONE box: ring of 512 symbols (16 waves per CU) against 2048 (14 waves per CU; tools/ab_old) in the decode kernel, 12M reads
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gz_gpu.py -x -q 2>&1 | tail -1
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 6000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
for rep in 1 2 3 4 5; do
  for v in ring512:hast_amd ring2048:tools/ab_old; do
    $(echo $v | cut -d: -f2)/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats 2> $D/err > $D/out
    echo "$(echo $v | cut -d: -f1) md5=$(md5sum < $D/out | cut -c1-8) $(grep -h __stats_phases__ $D/err | grep -o "read_phase_s=[0-9.]*")"
  done
done
rm -rf $D

# A/B on ONE box: the decode kernel with a copy per match (tools/ab_old = the commit before) against the batched one, gzip -6 files with
# constant and with noisy quality lines, alternating
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_phases__ $D/err.$name | cut -c1-300)"; }
for q in const noisy; do
  if [ $q = noisy ]; then export GEN_FASTQ_QUAL=noisy; fi
  tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
  (gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
  cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
  for rep in 1 2 3 4; do
    run ${q}_batched_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
    run ${q}_permatch_$rep tools/ab_old/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
  done
done
rm -rf $D

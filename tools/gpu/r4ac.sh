# A/B on ONE box: the device inflate before (tools/ab_old: two arenas of 8192 chunks, synchronous producer) and after this session's
# changes, alternating, with the open-time trace of the new one
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_phases__ $D/err.$name | cut -c1-300)"; grep -h "^gz open" $D/err.$name | cut -d: -f2 | tr '\n' ';'; echo; }
export HAST_GZ_TRACE=1
for rep in 1 2 3; do
  run new_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
  run old_$rep tools/ab_old/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
done
for rep in 4 5 6; do run new_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
for rep in 4 5 6; do run old_$rep tools/ab_old/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
rm -rf $D

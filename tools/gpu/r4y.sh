# device inflate on one box: default / no high-priority stream / no CU mask, with the open-time trace (where a 0.7-s stall of some boxes sits)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_read_phase__ $D/err.$name | cut -d" " -f2- | cut -c1-110)"; grep -h "__stats_phases__" $D/err.$name | cut -c1-420;  grep -h "^gz open" $D/err.$name | cut -d: -f2 | tr '\n' ';'; echo; }
export HAST_GZ_TRACE=1
for rep in 1 2 3; do run default_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
export HAST_GZ_POST_PRIORITY=0
for rep in 1 2 3; do run noprio_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
export HAST_GZ_FREE_CUS=0
for rep in 1 2 3; do run noprio_nomask_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
rm -rf $D

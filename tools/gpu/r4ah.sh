# round 4: device inflate with two literals per look-up: tests (corpus == zlib, CLI), 20M reads as two gzip -6 files (constant and noisy
# quality lines) whole process, kernel stats
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gz_gpu.py tests/test_cli_gpu.py -x -q > $O/r4ah_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/r4ah_pytest.log)"
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_read_phase__ $D/err.$name | cut -d" " -f2- | cut -c1-110)"; grep -h "__stats_phases__" $D/err.$name | cut -c1-420; }
for q in const noisy; do
  if [ $q = noisy ]; then export GEN_FASTQ_QUAL=noisy; fi
  tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
  (gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
  cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
  for rep in 1 2 3 4; do run ${q}_gz6_device$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4ah_prof_gz_$q -- hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats > $D/out.prof 2> $D/err.prof
  echo "under rocprofv3: md5=$(md5sum < $D/out.prof | cut -c1-12)"; grep -h "__stats_phases__" $D/err.prof | cut -c1-300
  python3 - <<PY
import csv,glob
f=glob.glob('$O/r4ah_prof_gz_$q/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:4]: print('  ', r['Name'][:40], r['Calls'], round(float(r['TotalDurationNs'])/1e6,1), 'ms')
PY
done
rm -rf $D

# round 4: fq device-block tests, partitioned stage-00 counting with per-tile reservations (tests, bench, kernel times), gz e2e again
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gz_gpu.py tests/test_fq_gpu.py -x -q -k "gz or device" > $O/r4g_pytest_gz.log 2>&1; echo "pytest gz/fq rc=$? $(tail -1 $O/r4g_pytest_gz.log)"
timeout -k 10 900 python -m pytest tests/test_kc_gpu.py -x -q -k "partition or trio or counts_hist or piled" > $O/r4g_pytest_kc.log 2>&1; echo "pytest kc rc=$? $(tail -1 $O/r4g_pytest_kc.log)"
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r4g_prof_s00 -o s00 -- python3 $GRAFT_REPO_ROOT/bench.py --workload s00 --cpu-seconds 0 --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$O/r4g_bench_s00.json 2> $GRAFT_REPO_ROOT/$O/r4g_bench_s00.err; cd $GRAFT_REPO_ROOT
python3 -c "
import json; d=json.load(open('$O/r4g_bench_s00.json')); print(round(d['value']/1e9,1), 'Gbp/s', d['seconds'], d['counting'])"
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_read_phase__ $D/err.$name | cut -d" " -f2- | cut -c1-110)"; grep -h "__stats_phases__\|__stats_gz__" $D/err.$name | cut -c1-460; if [ $rc != 0 ]; then tail -5 $D/err.$name; fi; }
for q in const; do
  D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
  tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
  (gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
  for rep in 1 2 3 4; do run ${q}_gz6_device$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
  cd /tmp && HAST_TEARDOWN=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r4g_prof_$q -o gz -- $GRAFT_REPO_ROOT/hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats > /dev/null 2> $GRAFT_REPO_ROOT/$O/r4g_prof_$q.err; cd $GRAFT_REPO_ROOT
  rm -rf $D
done

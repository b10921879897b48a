# device inflate after the per-symbol trims: tests, then 20M reads as two gzip -6 files four times
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gz_gpu.py tests/test_cli_gpu.py -x -q > $O/r4ao_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/r4ao_pytest.log)"
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_phases__ $D/err.$name | grep -o "read_phase_s=[0-9.]*\|total_s=[0-9.]*" | tr '\n' ' ')"; }
for rep in 1 2 3 4 5; do run gz6_device$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
rm -rf $D
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 3000000 5000000 100000 21 150 64 0 || exit 1
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4ao_prof -- hast_amd/classify --hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$O/r4ao_prof/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:4]: print('  ', r['Name'][:40], r['Calls'], round(float(r['TotalDurationNs'])/1e6,1), 'ms')
PY
rm -rf $D

# round 4, device inflate through the boundary: the CLI tests on .gz inputs, then 20M reads as two gzip -6 files through the
# device route and the host route (--stats: phases, gz stats), and the kernel times of the device route (rocprofv3)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gz_gpu.py -x -q > $O/r4c_pytest_gz.log 2>&1; echo "pytest gz rc=$? $(tail -1 $O/r4c_pytest_gz.log)"
timeout -k 10 600 python -m pytest tests/test_cli_gpu.py tests/test_fq_gpu.py -x -q -k "gz or device_inflate or parallel_ingest or striped" > $O/r4c_pytest_cli.log 2>&1; echo "pytest cli rc=$? $(tail -1 $O/r4c_pytest_cli.log)"
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_read_phase__ $D/err.$name | cut -d" " -f2- | cut -c1-150)"; grep -h "__stats_phases__\|__stats_gz__" $D/err.$name; }
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
ls -la $D/r1.fq.gz
cat $D/r1.fq.gz $D/r2.fq.gz $D/r1.fq $D/r2.fq > /dev/null
for rep in 1 2 3; do run gz6_device hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
HAST_INFLATE=host run gz6_host hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
run plain hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r4c_prof -o gz -- $GRAFT_REPO_ROOT/hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats > /dev/null 2> $GRAFT_REPO_ROOT/$O/r4c_prof.err; cd $GRAFT_REPO_ROOT
find $O/r4c_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -14 {}'
rm -rf $D

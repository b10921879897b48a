# round 5: ONE .gz stream over several contexts -- the new tests (striped device blocks fed by hast_gz_open_multi, the CLI's gz cases incl. damage
# behind the first pass and stdout on /dev/full), then `classify --devices` on 20M reads as two gzip -6 files: records per context, md5, times
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gz_gpu.py -x -q > $O/r5c_pytest_fq_gz.log 2>&1; rc=$?; echo "pytest fq+gz rc=$rc $(tail -1 $O/r5c_pytest_fq_gz.log)"
[ $rc = 0 ] || { tail -40 $O/r5c_pytest_fq_gz.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_cli_gpu.py -x -q -k "gz or inflate or output_errors or multi_gpu" > $O/r5c_pytest_cli.log 2>&1; rc=$?; echo "pytest cli rc=$rc $(tail -1 $O/r5c_pytest_cli.log)"
[ $rc = 0 ] || { tail -40 $O/r5c_pytest_cli.log; exit 1; }
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_phases__ $D/err.$name | grep -o "gpu_context_s=[0-9.]*\|scrub_sizes_clone_s=[0-9.]*\|read_phase_s=[0-9.]*\|total_s=[0-9.]*" | tr '\n' ' ') $(grep -h __stats_devices__ $D/err.$name | cut -d' ' -f3)"; }
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
for rep in 1 2; do
  run gz_one_ctx_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
  run gz_devices_0_0_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats --devices 0,0
  run gz_devices_0_0_0_0_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats --devices 0,0,0,0
done
HAST_GZ_SPLIT=contexts run gz_devices_0_0_split hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats --devices 0,0
run plain_one_ctx hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats
run plain_devices_0_0 hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats --devices 0,0
grep -h "__stats_gz__" $D/err.gz_devices_0_0_0_0_1 | cut -c1-400
rm -rf $D

# round 5, first GPU call: the token-parallel k_gz_decode -- device inflate tests (corpus == zlib), the CLI's gz tests, then an A/B on ONE box
# against round 4's tree (tools/ab_old: libhast.so + classify of commit 6d98496): 20M reads as two gzip -6 files, constant and noisy quality lines
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gz_gpu.py -x -q > $O/r5a_pytest_gz.log 2>&1; rc=$?; echo "pytest gz rc=$rc $(tail -1 $O/r5a_pytest_gz.log)"
[ $rc = 0 ] || { tail -30 $O/r5a_pytest_gz.log; exit 1; }
true
true
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_phases__ $D/err.$name | grep -o "gpu_context_s=[0-9.]*\|read_phase_s=[0-9.]*\|total_s=[0-9.]*" | tr '\n' ' ') $(grep -h __stats_gz__ $D/err.$name | grep -o "decode_s=[0-9.]*" | tr '\n' ' ')"; }
for q in const noisy; do
  if [ $q = noisy ]; then export GEN_FASTQ_QUAL=noisy; fi
  tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
  (gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
  echo "== quality lines: $q; $(stat -c %s $D/r1.fq) bytes per file, $(stat -c %s $D/r1.fq.gz) as gzip -6"
  cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
  for rep in 1 2 3; do
    run ${q}_new_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
    run ${q}_old_$rep ${AB_OLD:-tools/ab_old}/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
  done
  if [ $q = const ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5a_prof_gz -- hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats > $D/out.prof 2> $D/err.prof
    echo "under rocprofv3: md5=$(md5sum < $D/out.prof | cut -c1-12) $(grep -h __stats_phases__ $D/err.prof | cut -c1-300)"
    f=$(ls $O/r5a_prof_gz/*/*kernel_stats.csv | head -1); head -12 $f
  fi
done
rm -rf $D

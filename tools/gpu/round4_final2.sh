# round 4, the tree at the end of the round: whole -m gpu suite + smoke, then `classify` on 20M reads as two gzip -6 files (constant and
# noisy quality lines) inflated on the GPU and on the host, with the kernel stats of one device run -> profiles/round4_final2.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/round4_final2_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/round4_final2_pytest.log)"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_read_phase__ $D/err.$name | cut -d" " -f2- | cut -c1-80) $(grep -h __stats_phases__ $D/err.$name | grep -o "gpu_context_s=[0-9.]*\|read_phase_s=[0-9.]*\|total_s=[0-9.]*" | tr '\n' ' ')"; }
for q in const noisy; do
  if [ $q = noisy ]; then export GEN_FASTQ_QUAL=noisy; fi
  tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
  (gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
  echo "== quality lines: $q; $(stat -c %s $D/r1.fq) bytes per file, $(stat -c %s $D/r1.fq.gz) as gzip -6"
  cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
  for rep in 1 2 3 4; do run ${q}_gz6_device$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
  HAST_INFLATE=host run ${q}_gz6_host hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
  if [ $q = const ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/round4_prof_gz2 -- hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats > $D/out.prof 2> $D/err.prof
    echo "under rocprofv3: md5=$(md5sum < $D/out.prof | cut -c1-12) $(grep -h __stats_phases__ $D/err.prof | cut -c1-300)"
  fi
done
rm -rf $D

# round 5, the tree at the end of the round -> profiles/round5_final.txt (unedited):
#   the whole -m gpu suite + smoke; every bench line (default = BASELINE config 3, c2, c5, s00) -> profiles/round5_bench_*.json;
#   `classify` on 20M reads: plain, two gzip -6 files with constant and with noisy quality lines FIVE times each (<= 0.8 s
#   in every run of five), --devices 0,0 / 0,0,0,0 on the .gz files (records per context), the host route once; HAST_TRACE_INIT of five
#   starts (which call of the context's creation waits).
# usage: gpurun --timeout 1500 -- 'bash tools/gpu/round5_final.sh > gpurun_out/round5_final.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
if [ -z "$SKIP_SUITE" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/round5_final_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/round5_final_pytest.log)"
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
fi
if [ -z "$SKIP_BENCH" ]; then
  python bench.py > $O/round5_bench_default.json 2> $O/round5_bench_default.err; echo "bench default: $(python3 -c "import json;d=json.load(open('$O/round5_bench_default.json'));print(round(d['value']/1e9,1),'Gbp/s frac',d['roofline'].get('frac'),'frac_survey_8d',d['roofline'].get('frac_survey_8d'))")"
  for wl in c2 c5 s00; do
    python bench.py --workload $wl > $O/round5_bench_$wl.json 2> $O/round5_bench_$wl.err; echo "bench $wl: $(python3 -c "import json;d=json.load(open('$O/round5_bench_$wl.json'));print(round(d['value']/1e9,1),'Gbp/s frac',d['roofline'].get('frac'))")"
  done
fi
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_phases__ $D/err.$name | grep -o "gpu_context_s=[0-9.]*\|load_kmers_s=[0-9.]*\|scrub_sizes_clone_s=[0-9.]*\|read_phase_s=[0-9.]*\|total_s=[0-9.]*" | tr '\n' ' ') $(grep -h __stats_devices__ $D/err.$name | cut -d' ' -f3)"; }
for q in const noisy; do
  if [ $q = noisy ]; then export GEN_FASTQ_QUAL=noisy; fi
  tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
  (gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
  echo "== quality lines: $q; $(stat -c %s $D/r1.fq) bytes per file, $(stat -c %s $D/r1.fq.gz) as gzip -6"
  cat $D/r1.fq.gz $D/r2.fq.gz $D/r1.fq $D/r2.fq > /dev/null
  for rep in 1 2 3 4 5; do run ${q}_gz6_device$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
  HAST_INFLATE=host run ${q}_gz6_host hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
  if [ $q = const ]; then
    for rep in 1 2 3; do run plain$rep hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats; done
    for rep in 1 2; do
      run gz6_devices_0_0_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats --devices 0,0
      run gz6_devices_0_0_0_0_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats --devices 0,0,0,0
      run plain_devices_0_0_$rep hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats --devices 0,0
    done
    for rep in 1 2 3 4 5; do
      HAST_TRACE_INIT=1 hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats 2>&1 > /dev/null | grep "__trace_init__\|__stats_phases__" | sed "s/^/start $rep: /" | cut -c1-200
    done
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/round5_prof_gz -- hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats > $D/out.prof 2> $D/err.prof
    echo "under rocprofv3: md5=$(md5sum < $D/out.prof | cut -c1-12) $(grep -h __stats_phases__ $D/err.prof | cut -c1-300)"
    f=$(ls $O/round5_prof_gz/*/*kernel_stats.csv | head -1); cp $f $O/round5_cli_gz_device_kernel_stats.csv; head -14 $f
  fi
done
rm -rf $D

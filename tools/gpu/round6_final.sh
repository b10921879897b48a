# round 6, the tree at the end of the round -> profiles/round6_final.txt (unedited): the whole -m gpu suite + smoke; every bench line (default =
# BASELINE config 3, c2, c5, s00) -> profiles/round6_bench_*.json; `classify` on 20M reads over 10M barcodes with one dictionary per context
# (the merge by text that several GPUs need, on one GPU) beside the default.
# usage: gpurun --timeout 1200 -- 'bash tools/gpu/round6_final.sh > gpurun_out/round6_final.txt 2>&1'      SKIP_SUITE=1 / SKIP_BENCH=1 / SKIP_CLI=1
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
if [ -z "$SKIP_SUITE" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/round6_final_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/round6_final_pytest.log)"
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
fi
if [ -z "$SKIP_BENCH" ]; then
  python bench.py > $O/round6_bench_default.json 2> $O/round6_bench_default.err; echo "bench default: $(python3 -c "import json;d=json.load(open('$O/round6_bench_default.json'));print(round(d['value']/1e9,1),'Gbp/s frac',d['roofline'].get('frac'),'frac_survey_8d',d['roofline'].get('frac_survey_8d'),'cpu_baseline',d['cpu_baseline'].get('value'))")"
  for wl in c2 c5 s00; do
    python bench.py --workload $wl > $O/round6_bench_$wl.json 2> $O/round6_bench_$wl.err; echo "bench $wl: $(python3 -c "import json;d=json.load(open('$O/round6_bench_$wl.json'));print(round(d['value']/1e9,1),'Gbp/s frac',d['roofline'].get('frac'))")"
  done
fi
if [ -z "$SKIP_CLI" ]; then
  D=$(mktemp -d /dev/shm/hast_e2e.XXXXXX); trap 'rm -rf $D' EXIT
  now() { date +%s.%N; }
  run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
    echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s rows=$(wc -l < $D/out.$name) md5=$(md5sum < $D/out.$name | cut -c1-12)"
    grep -h "__stats_phases__\|__stats_dictionary__\|__stats_devices__" $D/err.$name | cut -c1-300 | sed 's/^/    /'; }
  tools/gen_fastq $D 10000000 5000000 10000000 21 150 64 0 || exit 1
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 --read $D/r1.fq --read $D/r2.fq -t 32 --stats"
  for rep in 1 2; do
    run one_context_$rep hast_amd/classify $ARGS
    run devices_0_0_0_shared_dictionary_$rep hast_amd/classify $ARGS --devices 0,0,0
    HAST_NAME_DICT=context run devices_0_0_0_dictionary_per_context_$rep hast_amd/classify $ARGS --devices 0,0,0
  done
fi

# stage 00: the tests of the count table, then `bench.py --workload s00` with the tree's kernels and with a switch set (A/B on ONE box,
# alternating), then the kernel stats of one step.  usage: gpurun -- 'AB_ENV="HAST_KC_EMIT=lanes" bash tools/gpu/s00_ab.sh > gpurun_out/s00_ab.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/s00_ab
mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 900 python -m pytest tests/test_kc_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?
  echo "pytest kc rc=$rc $(tail -1 $O/pytest.log)"
  [ $rc -ne 0 ] && { tail -40 $O/pytest.log; exit 1; }
fi
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], '%.1f Gbp/s' % (d['value']/1e9), '%.1f ms/step' % d['ms_per_step'], d.get('counting'))
" "$1" "$2"; }
for i in 1 2; do
  timeout -k 10 300 python3 bench.py --workload s00 --cpu-seconds 0 > $O/new_$i.json 2> $O/new_$i.err && line $O/new_$i.json "tree_$i" || { tail -5 $O/new_$i.err; exit 1; }
  if [ -n "$AB_ENV" ]; then
    timeout -k 10 300 env $AB_ENV python3 bench.py --workload s00 --cpu-seconds 0 > $O/old_$i.json 2> $O/old_$i.err && line $O/old_$i.json "[$AB_ENV]_$i" || { tail -5 $O/old_$i.err; exit 1; }
  fi
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload s00 --cpu-seconds 0 --steps 3 --warmup 1 > $O/prof.json 2> $O/prof.err || tail -3 $O/prof.err
f=$(ls $O/prof/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_kc_" in r["Name"]: print("%-60s calls %5s  avg %9.3f ms  total %9.1f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e6, float(r["TotalDurationNs"])/1e6))
PY
cp "$f" gpurun_out/s00_ab_kernel_stats.csv
rm -rf $O/prof

cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_kc_gpu.py -x -q > $O/r5i_pytest_kc.log 2>&1; rc=$?; echo "pytest kc rc=$rc $(tail -1 $O/r5i_pytest_kc.log)"
[ $rc = 0 ] || { tail -40 $O/r5i_pytest_kc.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5i_s00 -- python3 bench.py --workload s00 --cpu-seconds 0 > $O/r5i_bench_s00.json 2> $O/r5i_bench_s00.err
python3 -c "
import json;d=json.load(open('$O/r5i_bench_s00.json'));print('s00', round(d['value']/1e9,1),'Gbp/s', round(d['ms_per_step'],1),'ms/step', d['counting'].get('spilled_windows'))"
f=$(ls $O/r5i_s00/*/*kernel_stats.csv | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_kc' in r['Name']: print(r['Name'][:40], r['Calls'], round(float(r['TotalDurationNs'])/1e6,1), round(float(r['AverageNs'])/1e6,2))
PY
timeout -k 10 600 python -m pytest tests/test_cli_gpu.py -x -q -k "phase_reads or gz_decoders" > $O/r5i_pytest_cli.log 2>&1; rc=$?; echo "pytest cli rc=$rc $(tail -1 $O/r5i_pytest_cli.log)"
[ $rc = 0 ] || { tail -40 $O/r5i_pytest_cli.log; exit 1; }
bash tests/e2e/s00_e2e.sh 2>&1 | grep -o '"name": "[a-z0-9_]*", "rc": [0-9]*, "seconds": [0-9.]*\|table [0-9.]* s' | paste - - | head

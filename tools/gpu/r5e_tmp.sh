cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_kc_gpu.py -x -q > $O/r5e_pytest_kc.log 2>&1; rc=$?; echo "pytest kc rc=$rc $(tail -1 $O/r5e_pytest_kc.log)"
[ $rc = 0 ] || { tail -40 $O/r5e_pytest_kc.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_cli_gpu.py -x -q -k "one_million or output_errors or damaged_behind" > $O/r5e_pytest_cli.log 2>&1; rc=$?; echo "pytest cli rc=$rc $(tail -1 $O/r5e_pytest_cli.log)"
[ $rc = 0 ] || { tail -40 $O/r5e_pytest_cli.log; exit 1; }
python bench.py --workload s00 --cpu-seconds 0 > $O/r5e_bench_s00.json 2> $O/r5e_bench_s00.err; python3 -c "
import json;d=json.load(open('$O/r5e_bench_s00.json'));print('s00', round(d['value']/1e9,1),'Gbp/s', d['ms_per_step'],'ms/step', {k:v for k,v in d.get('roofline',{}).items() if k in ('frac','kernels_ms')})"
tail -3 $O/r5e_bench_s00.err
bash tests/e2e/s00_e2e.sh 2>&1 | tail -6

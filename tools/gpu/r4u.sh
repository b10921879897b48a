# round 4: stage 00 once more with the exact record cursor in front of a flush decision (one flush a step): tests, counters, bench line
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_kc_gpu.py -x -q > $O/r4u_pytest_kc.log 2>&1; echo "pytest kc rc=$? $(tail -1 $O/r4u_pytest_kc.log)"
bash profiles/collect_s00.sh round4_s00 > $O/round4_collect_s00.log 2>&1; tail -1 $O/round4_collect_s00.log
python bench.py --workload s00 > $O/round4_bench_s00.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/round4_bench_s00.json')); print('s00', round(d['value']/1e9,1), 'Gbp/s', d['seconds'], d['counting']['flushes_per_step'], json.dumps(d['roofline'])[:700])"

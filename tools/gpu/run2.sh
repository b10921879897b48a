cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02b_pytest.log 2>&1; tail -15 gpurun_out/r02b_pytest.log
for mode in filter exact; do
  HAST_CLASSIFY=$mode timeout 600 python bench.py --cpu-seconds 0 > gpurun_out/r02b_bench_$mode.json 2> gpurun_out/r02b_bench_$mode.err; tail -2 gpurun_out/r02b_bench_$mode.err; python3 -c "
import json; d=json.load(open('gpurun_out/r02b_bench_$mode.json')); print('$mode', d['value']/1e9, d['roofline']['kernel_ms_avg'], d['hits'])"
done

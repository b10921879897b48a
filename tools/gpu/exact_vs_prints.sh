# exact filter entries against prints: parity tests, then the bench in both modes (random and clustered keys, C2)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "filter" > gpurun_out/r02q_filtertests.log 2>&1; echo "filter tests rc=$?"; tail -3 gpurun_out/r02q_filtertests.log
python -m pytest tests -m gpu -x -q > gpurun_out/r02q_gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02q_gputest.log
one() { # name, env, flags
  local name=$1; shift; local envs=$1; shift
  env $envs python bench.py --cpu-seconds 0 --steps 10 "$@" > gpurun_out/r02q_bench_$name.json 2> gpurun_out/r02q_bench_$name.err || tail -3 gpurun_out/r02q_bench_$name.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r02q_bench_$name.json')); print('$name', round(d['value']/1e9,1), 'Gbp/s', round(d['roofline']['kernel_ms_avg'],2), 'ms', d['config']['filter'], d['hits'])"
}
one c3_exact "X=1"
one c3_prints "HAST_FILTER_EXACT=0"
one c3cl_exact "X=1" --clustered
one c3cl_prints "HAST_FILTER_EXACT=0" --clustered
one c2_exact "X=1" --workload c2
one c2_prints "HAST_FILTER_EXACT=0" --workload c2
one c5 "X=1" --workload c5

cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
nproc; free -g | head -2
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r02a_pytest.log 2>&1; tail -5 gpurun_out/r02a_pytest.log
( time python bench.py ) > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err; tail -3 gpurun_out/r02a_bench.err; cat gpurun_out/r02a_bench.json | head -c 3000
bash profiles/collect.sh r02a > gpurun_out/r02a_collect.log 2>&1; tail -3 gpurun_out/r02a_collect.log

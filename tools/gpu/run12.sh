cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
( time timeout 2000 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02j_pytest.log 2>&1; tail -4 gpurun_out/r02j_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time python bench.py ) > gpurun_out/r02j_bench.json 2> gpurun_out/r02j_bench.err; tail -2 gpurun_out/r02j_bench.err; head -c 2500 gpurun_out/r02j_bench.json

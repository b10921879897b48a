cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_fq_gpu.py tests/test_cli_gpu.py -m gpu -x -q ) > gpurun_out/r02i_pytest.log 2>&1; tail -4 gpurun_out/r02i_pytest.log
bash tools/gpu/run9.sh

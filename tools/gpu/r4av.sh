# the --devices golden test with its new assertion on the __stats_switches__ line
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 200 python -m pytest tests/test_cli_gpu.py -x -q -k "multi_gpu" 2>&1 | tail -3

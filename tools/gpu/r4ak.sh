# tests/test_gz_gpu.py alone (the run that found the hang of a stream closed early)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gz_gpu.py -x -q 2>&1 | tail -3

# round 4, device inflate: first run of the kernels (tests/test_gz_gpu.py) + the counters test that failed on its own arithmetic
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gz_gpu.py -x -q > $O/r4b_pytest_gz.log 2>&1; echo "pytest gz rc=$? $(tail -1 $O/r4b_pytest_gz.log)"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "counters_past or offsets_that or get_hap_and" > $O/r4b_pytest_misc.log 2>&1; echo "pytest misc rc=$? $(tail -1 $O/r4b_pytest_misc.log)"

# the early-close test of the device inflate six times in fresh processes (an intermittent hang: CU-masked stream re-created), then the whole file
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for v in 1 2 3 4 5 6; do
  timeout -k 5 60 python -m pytest tests/test_gz_gpu.py -x -q -s -k "closed_early" > gpurun_out/r4am_$v.log 2>&1; echo "run $v rc=$? $(grep -c 'early close' gpurun_out/r4am_$v.log) cases started; $(tail -1 gpurun_out/r4am_$v.log | cut -c1-80)"
done
timeout -k 10 300 python -m pytest tests/test_gz_gpu.py -x -q > gpurun_out/r4am_all.log 2>&1; echo "whole file rc=$? $(tail -1 gpurun_out/r4am_all.log)"

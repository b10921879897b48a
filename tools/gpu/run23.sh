cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
bash profiles/collect.sh r03a > gpurun_out/r03a_collect.log 2>&1; tail -1 gpurun_out/r03a_collect.log
bash profiles/collect.sh r03a_clustered --clustered > gpurun_out/r03a_collect_cl.log 2>&1; tail -1 gpurun_out/r03a_collect_cl.log
bash profiles/collect.sh r03a_c5 --workload c5 > gpurun_out/r03a_collect_c5.log 2>&1; tail -1 gpurun_out/r03a_collect_c5.log
bash profiles/collect.sh r03a_c2 --workload c2 > gpurun_out/r03a_collect_c2.log 2>&1; tail -1 gpurun_out/r03a_collect_c2.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -2

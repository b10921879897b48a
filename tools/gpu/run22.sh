cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
bash profiles/collect.sh r02x > gpurun_out/r02x_collect.log 2>&1; tail -1 gpurun_out/r02x_collect.log
bash profiles/collect.sh r02x_clustered --clustered > gpurun_out/r02x_collect_cl.log 2>&1; tail -1 gpurun_out/r02x_collect_cl.log
bash profiles/collect.sh r02x_c5 --workload c5 > gpurun_out/r02x_collect_c5.log 2>&1; tail -1 gpurun_out/r02x_collect_c5.log
bash profiles/collect.sh r02x_c2 --workload c2 > gpurun_out/r02x_collect_c2.log 2>&1; tail -1 gpurun_out/r02x_collect_c2.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -2

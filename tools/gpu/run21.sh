cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
bash profiles/collect.sh r02v_c5 --workload c5 > gpurun_out/r02v_collect_c5.log 2>&1; tail -1 gpurun_out/r02v_collect_c5.log
bash profiles/collect.sh r02v_c2 --workload c2 > gpurun_out/r02v_collect_c2.log 2>&1; tail -1 gpurun_out/r02v_collect_c2.log

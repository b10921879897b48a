cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
bash profiles/collect.sh r02v > gpurun_out/r02v_collect.log 2>&1; tail -2 gpurun_out/r02v_collect.log
bash profiles/collect.sh r02v_clustered --clustered > gpurun_out/r02v_collect_cl.log 2>&1; tail -2 gpurun_out/r02v_collect_cl.log

# round 4 profiles, part 1: the default command and --clustered (profiles/collect.sh: kernel trace + stats, then one --pmc pass per group)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=round4
bash profiles/collect.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1; tail -1 gpurun_out/${TAG}_collect.log
bash profiles/collect.sh ${TAG}_clustered --clustered > gpurun_out/${TAG}_collect_cl.log 2>&1; tail -1 gpurun_out/${TAG}_collect_cl.log

# one part of tools/gpu/collect_all.sh (a gpurun call is limited to 20 minutes): bash tools/gpu/collect_part.sh <tag> <part> with part = a (the default
# command + --clustered), b (--workload c5, --workload c2), c (stage 00: the default path and the direct kernel)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:?tag}; PART=${2:?part}
case $PART in
a) bash profiles/collect.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1; tail -1 gpurun_out/${TAG}_collect.log
   bash profiles/collect.sh ${TAG}_clustered --clustered > gpurun_out/${TAG}_collect_cl.log 2>&1; tail -1 gpurun_out/${TAG}_collect_cl.log;;
b) bash profiles/collect.sh ${TAG}_c5 --workload c5 > gpurun_out/${TAG}_collect_c5.log 2>&1; tail -1 gpurun_out/${TAG}_collect_c5.log
   bash profiles/collect.sh ${TAG}_c2 --workload c2 > gpurun_out/${TAG}_collect_c2.log 2>&1; tail -1 gpurun_out/${TAG}_collect_c2.log;;
c) bash profiles/collect_s00.sh ${TAG}_s00 > gpurun_out/${TAG}_collect_s00.log 2>&1; tail -1 gpurun_out/${TAG}_collect_s00.log
   bash profiles/collect_s00.sh ${TAG}_s00_atomic atomic > gpurun_out/${TAG}_collect_s00a.log 2>&1; tail -1 gpurun_out/${TAG}_collect_s00a.log;;
esac

# everything that ties the committed profiles to the device sources of the tree, in one gpurun call:
#     bash tools/gpu/collect_all.sh <tag>
# profiles/collect.sh for the default command, --clustered, --workload c5 and --workload c2 (kernel trace + stats, then one --pmc
# pass per counter group), profiles/collect_s00.sh for stage 00 (the default path of the table's size and, with S00_ATOMIC=1, the
# direct kernel), then -- with SUITE=1 -- the GPU test suite.  Afterwards, in the build container:
#     for t in <tag> <tag>_clustered <tag>_c5 <tag>_c2; do python3 profiles/summarize.py $t; done
#     python3 profiles/summarize_s00.py <tag>_s00 [; python3 profiles/summarize_s00.py <tag>_s00_atomic]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:?tag}
bash profiles/collect.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1; tail -1 gpurun_out/${TAG}_collect.log
bash profiles/collect.sh ${TAG}_clustered --clustered > gpurun_out/${TAG}_collect_cl.log 2>&1; tail -1 gpurun_out/${TAG}_collect_cl.log
bash profiles/collect.sh ${TAG}_c5 --workload c5 > gpurun_out/${TAG}_collect_c5.log 2>&1; tail -1 gpurun_out/${TAG}_collect_c5.log
bash profiles/collect.sh ${TAG}_c2 --workload c2 > gpurun_out/${TAG}_collect_c2.log 2>&1; tail -1 gpurun_out/${TAG}_collect_c2.log
bash profiles/collect_s00.sh ${TAG}_s00 > gpurun_out/${TAG}_collect_s00.log 2>&1; tail -1 gpurun_out/${TAG}_collect_s00.log
[ -n "${S00_ATOMIC:-}" ] && { bash profiles/collect_s00.sh ${TAG}_s00_atomic atomic > gpurun_out/${TAG}_collect_s00a.log 2>&1; tail -1 gpurun_out/${TAG}_collect_s00a.log; }
[ -n "${SUITE:-}" ] && python -m pytest tests -m gpu -x -q 2>&1 | tail -2
true

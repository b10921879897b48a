# round 4 profiles, part 2: config 5, C2, stage 00 (default = partitioned, and the direct kernel)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=round4
bash profiles/collect.sh ${TAG}_c5 --workload c5 > gpurun_out/${TAG}_collect_c5.log 2>&1; tail -1 gpurun_out/${TAG}_collect_c5.log
bash profiles/collect.sh ${TAG}_c2 --workload c2 > gpurun_out/${TAG}_collect_c2.log 2>&1; tail -1 gpurun_out/${TAG}_collect_c2.log
bash profiles/collect_s00.sh ${TAG}_s00 > gpurun_out/${TAG}_collect_s00.log 2>&1; tail -1 gpurun_out/${TAG}_collect_s00.log
bash profiles/collect_s00.sh ${TAG}_s00_atomic atomic > gpurun_out/${TAG}_collect_s00a.log 2>&1; tail -1 gpurun_out/${TAG}_collect_s00a.log

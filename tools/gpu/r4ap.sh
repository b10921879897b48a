# stage 00's counters again on the final kc sources (comments changed: the profiles are tied to a hash of the files), both paths; bench lines
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
bash profiles/collect_s00.sh round4_s00 > $O/round4_collect_s00.log 2>&1; tail -1 $O/round4_collect_s00.log
bash profiles/collect_s00.sh round4_s00_atomic atomic > $O/round4_collect_s00a.log 2>&1; tail -1 $O/round4_collect_s00a.log

cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
bash tools/gpu/collect_part.sh round5 c
bash tools/gpu/gz_pmc.sh > $O/round5_gz_kernels_pmc.txt 2>&1; tail -2 $O/round5_gz_kernels_pmc.txt | cut -c1-200
AB_OLD=tools/ab_old bash tools/gpu/gz_ab.sh > $O/round5_ab_gz_vs_round4.txt 2>&1; grep "_new_\|_old_" $O/round5_ab_gz_vs_round4.txt | cut -c1-200

cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
bash profiles/collect.sh r02u > gpurun_out/r02u_collect.log 2>&1; tail -2 gpurun_out/r02u_collect.log
bash profiles/collect.sh r02u_clustered --clustered > gpurun_out/r02u_collect_cl.log 2>&1; tail -2 gpurun_out/r02u_collect_cl.log
for wl in c2 c5; do python bench.py --workload $wl --cpu-seconds 0 > gpurun_out/r02u_bench_$wl.json 2>/dev/null; done
python bench.py --clustered --cpu-seconds 0 > gpurun_out/r02u_bench_c3_clustered.json 2>/dev/null
HAST_CLASSIFY=exact python bench.py --cpu-seconds 0 > gpurun_out/r02u_bench_c3_exact_kernel.json 2>/dev/null
python bench.py --steps 60 --cpu-seconds 0 > gpurun_out/r02u_bench_c3_2p9Breads.json 2>/dev/null
for f in gpurun_out/r02u_bench_*.json; do python3 -c "
import json,sys; d=json.load(open('$f')); print('$f', round(d['value']/1e9,1), round(d['roofline']['kernel_ms_avg'],2))"; done

# round 4: stage-00 partitioned counting after (1) 8 waves/SIMD in the partition kernels, (2) half-tile LDS staging in the emit kernel,
# (3) record loads one step ahead in k_kc_apply; tile sizes of the emit kernel
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kc_gpu.py -x -q > $O/r4q_pytest_kc.log 2>&1; echo "pytest kc rc=$? $(tail -1 $O/r4q_pytest_kc.log)"
for tile in 4096; do
HAST_KC_TILE=$tile HAST_KC_COUNT=partition timeout -k 10 300 python3 bench.py --workload s00 --cpu-seconds 0 --steps 2 --warmup 1 > $O/r4q_s00_$tile.json 2> $O/r4q_s00_$tile.err
python3 -c "
import json; d=json.load(open('$O/r4q_s00_$tile.json')); print('tile $tile:', round(d['value']/1e9,1), 'Gbp/s', round(d['seconds']['count'],4), d['counting']['spilled_windows'])" | cut -c1-600
done
HAST_KC_COUNT=partition rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4q_prof_s00 -- python3 bench.py --workload s00 --cpu-seconds 0 --steps 2 --warmup 1 > /dev/null 2> $O/r4q_prof.err
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4q_prof_s00/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r['Name'][:45], r['Calls'], round(float(r['TotalDurationNs'])/1e6,1), round(float(r['AverageNs'])/1e6,2))
PY

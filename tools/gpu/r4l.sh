# round 4: k_kc_apply with a lane per window (bucket-wide probes): stage-00 tests, then the s00 bench both ways
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kc_gpu.py -x -q > $O/r4l_pytest_kc.log 2>&1; echo "pytest kc rc=$? $(tail -1 $O/r4l_pytest_kc.log)"
for mode in partition atomic; do
HAST_KC_COUNT=$mode timeout -k 10 300 python3 bench.py --workload s00 --cpu-seconds 0 --steps 2 --warmup 1 > $O/r4l_s00_$mode.json 2> $O/r4l_s00_$mode.err
python3 -c "
import json; d=json.load(open('$O/r4l_s00_$mode.json')); print('$mode:', round(d['value']/1e9,1), 'Gbp/s', d['seconds'], d['counting'])" | cut -c1-600
done
HAST_KC_COUNT=partition rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4l_prof_s00 -- python3 bench.py --workload s00 --cpu-seconds 0 --steps 2 --warmup 1 > /dev/null 2> $O/r4l_prof.err
f=$(find $O/r4l_prof_s00 -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | cut -c1-50,200- | head -12

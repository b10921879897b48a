# round 4: stage 00 with partitioned counting as the default of large tables (slice-local placement by the key's hash, select through
# LDS, small flushes through the atomic path): tests, the bench both ways, the 60-Mbp trio through the program
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kc_gpu.py -x -q > $O/r4r_pytest_kc.log 2>&1; echo "pytest kc rc=$? $(tail -1 $O/r4r_pytest_kc.log)"
for mode in default atomic; do
if [ $mode = atomic ]; then export HAST_KC_COUNT=atomic; fi
timeout -k 10 300 python3 bench.py --workload s00 --cpu-seconds 0 > $O/r4r_s00_$mode.json 2> $O/r4r_s00_$mode.err
python3 -c "
import json; d=json.load(open('$O/r4r_s00_$mode.json')); print('$mode:', round(d['value']/1e9,1), 'Gbp/s', {k: round(v,4) for k,v in d['seconds'].items()}, d['counting']['spilled_windows'], d['counting']['flushes'])" | cut -c1-600
done
unset HAST_KC_COUNT
if [ -f tests/e2e/s00_e2e.sh ]; then timeout -k 10 500 bash tests/e2e/s00_e2e.sh 20000000 > $O/r4r_s00_e2e.log 2>&1; echo "e2e rc=$?"; tail -15 $O/r4r_s00_e2e.log | cut -c1-300; fi

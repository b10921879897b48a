# stage 00, instrumented build of the first partitioned version: emit only / all passes but the LDS probes / everything
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
for dbg in 1 2 0; do
HAST_KC_DEBUG=$dbg python3 bench.py --workload s00 --cpu-seconds 0 --steps 2 --warmup 1 > $O/r4j_s00_dbg$dbg.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/r4j_s00_dbg$dbg.json')); print('debug $dbg:', round(d['value']/1e9,1), 'Gbp/s', round(d['seconds']['count'],4))"
done

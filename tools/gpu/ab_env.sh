# A/B of one library under two environments on ONE box:  bash tools/gpu/ab_env.sh "<bench flags>" "VAR=VAL" ["VAR2=VAL2" ...]
# every setting (and the unset default, "-") is benched twice, interleaved
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
FLAGS=$1; shift
for round in 1 2; do
  for setting in - "$@"; do
    ( if [ "$setting" != "-" ]; then export "$setting"; fi
      timeout 900 python bench.py --cpu-seconds 0 --steps 10 $FLAGS > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
      python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$setting [$FLAGS]', round(d['value']/1e9,1), 'Gbp/s kernel', round(d['roofline']['kernel_ms_avg'],3), 'ms (min', round(d['roofline']['kernel_ms_min'],3), ') ceiling', round(d['roofline'].get('request_ceiling_this_run',{}).get('requests_per_s',0)/1e9,2), 'hits', d['hits']['c0'], d['hits']['c1'])" )
  done
done

cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for round in 1 2; do
 for setting in - HAST_COMMIT=atomic; do
  for flags in "" "--workload c2" "--workload c1"; do
  ( if [ "$setting" != "-" ]; then export "$setting"; fi
    timeout 900 python bench.py --cpu-seconds 0 --steps 10 $flags > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
    python3 -c "
import json; d=json.load(open('/tmp/ab.json')); r=d['roofline']; print('$setting [$flags]', round(d['value']/1e9,1), 'Gbp/s step', round(d['ms_per_step'],3), 'kernel', round(r['kernel_ms_avg'],3), 'commit', round(r['commit_kernel_ms_avg'],3), 'hits', d['hits'])" )
  done
 done
done

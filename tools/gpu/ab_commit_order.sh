cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for setting in HAST_COMMIT=atomic HAST_COMMIT=partition HAST_COMMIT=atomic HAST_COMMIT=partition HAST_COMMIT=atomic HAST_COMMIT=partition; do
  ( export "$setting"
    python bench.py --cpu-seconds 0 --steps 20 > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
    python3 -c "
import json; d=json.load(open('/tmp/ab.json')); r=d['roofline']; print('$setting', round(d['value']/1e9,1), 'Gbp/s step', round(d['ms_per_step'],3), 'kernel', round(r['kernel_ms_avg'],3), 'min', round(r['kernel_ms_min'],3), 'commit', round(r['commit_kernel_ms_avg'],3))" )
done

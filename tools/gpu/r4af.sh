# config 5: tile size / waves per SIMD (77 VGPRs allow 6 waves per SIMD; the 32-KB tiles allow 5 workgroups per CU)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
for lds in 0 26000 22800 0 26000; do
  if [ $lds = 0 ]; then unset HAST_TILE_LDS; else export HAST_TILE_LDS=$lds; fi
  python bench.py --cpu-seconds 0 --workload c5 --steps 10 > $O/r4af_c5_$lds.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$O/r4af_c5_$lds.json')); r=d['roofline']; print('c5 tile_lds=$lds', round(d['value']/1e9,1), 'Gbp/s kernel', round(r.get('kernel_ms_avg',0),2), 'ms; of the ceiling', (r.get('request_rate') or {}).get('frac_of_ceiling_this_run'))"
done
for lds in 0 26000; do
  if [ $lds = 0 ]; then unset HAST_TILE_LDS; else export HAST_TILE_LDS=$lds; fi
  python bench.py --cpu-seconds 0 --no-secondary --steps 5 > $O/r4af_c3_$lds.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$O/r4af_c3_$lds.json')); r=d['roofline']; print('c3 tile_lds=$lds', round(d['value']/1e9,1), 'Gbp/s kernel', round(r.get('kernel_ms_avg',0),2), 'ms; of the ceiling', (r.get('request_rate') or {}).get('frac_of_ceiling_this_run'))"
done

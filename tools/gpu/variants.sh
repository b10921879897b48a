# builds libhast variants with extra -D flags on the GPU box and benches each:  bash tools/gpu/variants.sh "name:-DFLAG=.. -D.." ...
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p /tmp/variants gpurun_out
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( cd hast_amd/csrc && for f in hast_kernels hast_filter; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $f.hip -o /tmp/variants/${f}_$name.o; done;
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/variants/libhast_$name.so /tmp/variants/hast_kernels_$name.o /tmp/variants/hast_filter_$name.o hast_api.o fq_kernels.o fq_api.o kc_kernels.o kc_api.o -ldl ) 2>&1 | grep -E "error" | head -3
done
for spec in "base:" "$@"; do
  name=${spec%%:*}
  if [ $name = base ]; then unset HAST_LIB; else export HAST_LIB=/tmp/variants/libhast_$name.so; fi
  for wl in "" "--clustered"; do
    timeout 600 python bench.py --cpu-seconds 0 --steps 10 $wl > /tmp/variants/b.json 2> /tmp/variants/b.err || tail -3 /tmp/variants/b.err
    python3 -c "
import json; d=json.load(open('/tmp/variants/b.json')); print('$name $wl', round(d['value']/1e9,1), 'Gbp/s', round(d['roofline']['kernel_ms_avg'],2), 'ms', d['hits']['c0'])"
  done
done

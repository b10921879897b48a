# HBM read requests per read of k_classify_f for the tree's library and for variants built with extra -D flags:
#   bash tools/gpu/rdreq.sh "name:-DFLAG=.." ...      (one rocprofv3 --pmc pass each; prints requests per read and kernel ms)
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
  rm -rf /tmp/variants/pmc_$name
  rocprofv3 --pmc TCC_EA0_RDREQ_sum SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/variants/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 $RDREQ_FLAGS > /tmp/variants/pmc_$name.json 2> /tmp/variants/pmc_$name.err
  python3 - "$name" <<'PY'
import csv, glob, sys, collections
name = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob('/tmp/variants/pmc_%s/**/*counter_collection.csv' % name, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_classify_f' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in tot:
    print(name, k, 'per launch', tot[k] / n[k], 'launches', n[k], 'per read (48M)', tot[k] / n[k] / 48e6)
PY
done

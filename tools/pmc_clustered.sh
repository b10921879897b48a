cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for V in default split; do
if [ $V = default ]; then unset HAST_LIB; else export HAST_LIB=$PWD/hast_amd/variants/libhast_$V.so; fi
for C in "" "--clustered --no-plants"; do
tag=$(echo "$V$C" | tr -d ' -')
rocprofv3 --pmc TCC_EA0_RDREQ_sum SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmccl_$tag -- python3 bench.py --workload c3 --steps 2 --warmup 1 --cpu-seconds 0 $C > /dev/null 2>&1
done; done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmccl_*")):
    f=glob.glob(d+"/*/*_counter_collection.csv")
    if not f: print(d,"no data"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_classify" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d.split("pmccl_")[1], {k: "%.4g"%v[-1] for k,v in sorted(agg.items())})
PY

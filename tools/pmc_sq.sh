cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for W in c1 c3; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/pmc2_$W -- python3 bench.py --workload $W --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc3_$W -- python3 bench.py --workload $W --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc[23]_c*")):
    f=glob.glob(d+"/*/*_counter_collection.csv")
    if not f: print(d,"no data"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_classify" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d, {k: "%.3g"%v[-1] for k,v in sorted(agg.items())})
PY

# what bounds stage 00's kernels: SQ cycle counters (busy / VALU / scalar / LDS active, waits) of one step of `bench.py --workload s00`
# usage: gpurun -- 'bash tools/gpu/s00_pmc.sh > gpurun_out/s00_pmc.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/s00_pmc
mkdir -p $O
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
            "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_$name -- python3 bench.py --workload s00 --cpu-seconds 0 --steps 1 --warmup 0 > $O/$name.json 2> $O/$name.err || tail -3 $O/$name.err
done
python3 - "$O" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hast::", "")
        if "k_kc_" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(agg):
    print(k, {c: "%.4g" % v for c, v in sorted(agg[k].items())})
PY
rm -rf $O

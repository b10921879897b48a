# round 4: what the partitioned stage-00 kernels wait for (SQ counters per kernel)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r4p
mkdir -p $O
CMD="python3 bench.py --workload s00 --cpu-seconds 0 --steps 1 --warmup 0"
export HAST_KC_COUNT=partition
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_ACTIVE_INST_SCA" \
            "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_VALU_MFMA_BUSY_CYCLES"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_$name -- $CMD > $O/pmc_$name.json 2> $O/pmc_$name.err
done
python3 - "$O" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_kc" not in k: continue
        k = k.split("(")[0].replace("void ", "").replace("hast::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in agg:
    print(k, {c: "%.3g" % v for c, v in sorted(agg[k].items())})
PY

#!/bin/bash
# rocprofv3 evidence for the stage-00 count kernel (run through gpurun from the repo root): kernel stats, then PMC passes
# (each in its own run, --kernel-trace only) of bench.py --workload s00 on a 50-Mbp trio with the table at load factor ~0.5.
set -u
TAG=${1:-s00}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python3 bench.py --workload s00 --genome 50e6 --table-gb 9.5 --cpu-seconds 0 --steps 1 --warmup 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats_bench.json 2> $OUT/stats_bench.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_ATOMIC_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_VMEM"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- $CMD > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
res = {}
for f in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith(("void hast::k_kc", "hast::k_kc", "void rocprim")) or "k_kc" in r["Name"]:
            res.setdefault("stats", []).append({k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")})
for d in sorted(glob.glob(out + "/pmc_*")):
    f = glob.glob(d + "/*/*_counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(float)
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(f[0])):
        if "k_kc_count" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
    for k in agg:
        res.setdefault("pmc_sum_over_launches", {})[k] = agg[k]
        res.setdefault("launches", {})[k] = n[k]
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY

cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "filter or baseline_size or randomized or strict or perread or long" ) > gpurun_out/r02g_pytest.log 2>&1; tail -3 gpurun_out/r02g_pytest.log
for lds in 32000 40000 52000 78000; do
HAST_TILE_LDS=$lds python bench.py --cpu-seconds 0 --steps 10 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('lds$lds', d['value']/1e9, d['roofline']['kernel_ms_avg'])"
done
python bench.py --cpu-seconds 0 --steps 10 --clustered 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('clustered', d['value']/1e9, d['roofline']['kernel_ms_avg'])"
python bench.py --workload c5 --cpu-seconds 0 --steps 10 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('c5', d['value']/1e9, d['roofline']['kernel_ms_avg'], d['config']['filter'])"
OUT=gpurun_out/prof_r02g; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS TCC_EA0_RDREQ_sum SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --batch-reads 16000000 > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/prof_r02g/pmc_*")):
    f=glob.glob(d+"/*/*_counter_collection.csv")
    if not f: continue
    agg=collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]]=float(r["Counter_Value"])
    for k,v in agg.items():
        if "classify" in k: print(k, {a: "%.4g"%b for a,b in sorted(v.items())})
PY

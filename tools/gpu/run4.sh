cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cli_gpu.py -m gpu -x -q ) > gpurun_out/r02d_pytest.log 2>&1; tail -4 gpurun_out/r02d_pytest.log
for mode in filter exact; do
  HAST_CLASSIFY=$mode timeout 600 python bench.py --cpu-seconds 0 > gpurun_out/r02d_bench_$mode.json 2> gpurun_out/r02d_bench_$mode.err; tail -1 gpurun_out/r02d_bench_$mode.err; python3 -c "
import json; d=json.load(open('gpurun_out/r02d_bench_$mode.json')); print('$mode', d['value']/1e9, d['roofline']['kernel_ms_avg'], d['hits'])"
done
OUT=gpurun_out/prof_r02d; mkdir -p $OUT
for name in rdsize sq sq2; do
  case $name in
    rdsize) P="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum";;
    sq) P="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES";;
    sq2) P="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT";;
  esac
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --batch-reads 16000000 > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/prof_r02d/pmc_*")):
    f=glob.glob(d+"/*/*_counter_collection.csv")
    if not f: continue
    agg=collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]]=float(r["Counter_Value"])
    for k,v in agg.items():
        if "classify" in k or "filter" in k: print(k, {a: "%.4g"%b for a,b in sorted(v.items())})
PY

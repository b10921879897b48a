# what the kernels of the device inflate issue: SQ counters of one `classify` run over two gzip -6 files of 6M reads each
# (2.05 GB of FASTQ per file); prints one line per k_gz_* kernel.  usage: gpurun -- 'bash tools/gpu/gz_pmc.sh > gpurun_out/gz_pmc.txt'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/gz_pmc
mkdir -p $O
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
[ -n "$GZ_PMC_NOISY" ] && export GEN_FASTQ_QUAL=noisy
tools/gen_fastq $D 6000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
echo "inflated bytes per file: $(stat -c %s $D/r1.fq), compressed: $(stat -c %s $D/r1.fq.gz)"
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_$name -- hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 > $D/out 2> $D/err; echo "md5 $(md5sum < $D/out | cut -c1-12)"
done
python3 - "$O" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hast::", "")
        if "k_gz_" not in k: continue
        agg[k][r["Counter_Name"] + ("" if "pmc_SQ_INSTS_VALU" in f or r["Counter_Name"] not in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES") else "@" + f.split("/pmc_")[1].split("/")[0])] += float(r["Counter_Value"])
for k in sorted(agg):
    print(k, {c: "%.4g" % v for c, v in sorted(agg[k].items())})
PY
rm -rf $D $O

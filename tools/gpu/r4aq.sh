# device inflate: table entries taken as wave-uniform values (scalar tests): gz tests, then 6M reads under rocprofv3 for the kernel times
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gz_gpu.py -x -q > $O/r4aq_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/r4aq_pytest.log)"
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 3000000 5000000 100000 21 150 64 0 || exit 1
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
for rep in 1 2; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4aq_prof$rep -- hast_amd/classify --hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 > $D/out 2>/dev/null
md5sum < $D/out | cut -c1-12
python3 - <<PY
import csv,glob
f=glob.glob('$O/r4aq_prof$rep/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:3]: print('  ', r['Name'][:40], r['Calls'], round(float(r['TotalDurationNs'])/1e6,1), 'ms')
PY
done
rm -rf $D

# round 4: whole -m gpu suite on the final tree, stage 00's counters again (one flush a step: the host follows the device's record
# cursor), the kernels of one `classify` run over two .gz files inflated on the GPU
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r4t_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/r4t_pytest.log)"
bash profiles/collect_s00.sh round4_s00 > $O/round4_collect_s00.log 2>&1; tail -1 $O/round4_collect_s00.log
python bench.py --workload s00 > $O/round4_bench_s00.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/round4_bench_s00.json')); print('s00', round(d['value']/1e9,1), 'Gbp/s', d['seconds'], d['counting']['flushes_per_step'], json.dumps(d['roofline'])[:900])"
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/round4_prof_gz -- hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats > $D/out.prof 2> $D/err.prof
echo "under rocprofv3: md5=$(md5sum < $D/out.prof | cut -c1-12)"; grep -h "__stats_gz__\|__stats_phases__" $D/err.prof | cut -c1-400
hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats --devices 0,0 2>&1 > /dev/null | grep "__stats_devices__"
rm -rf $D

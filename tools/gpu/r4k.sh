# round 4, status pass of the second session: whole -m gpu suite, config 5 after the tile-level dirty flag, what --devices costs
# with one name cache per GPU (20M reads, plain FASTQ, page cache warm)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r4k_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/r4k_pytest.log)"
timeout -k 10 300 python bench.py --cpu-seconds 0 --workload c5 > $O/r4k_bench_c5.json 2> $O/r4k_bench_c5.err; echo "c5 rc=$?"
python3 -c "
import json; d=json.load(open('$O/r4k_bench_c5.json')); r=d['roofline']; print('c5', round(d['value']/1e9,1), 'Gbp/s', round(d['ms_per_step'],2), 'ms/step kernel', round(r.get('kernel_ms_avg',0) or 0,2), 'ceiling', r.get('request_ceiling_this_run'), r.get('request_rate'))"
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq --read $D/r2.fq --weight0 1.04"
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_read_phase__ $D/err.$name | cut -d" " -f2- | cut -c1-150)"; grep -h "__stats_phases__" $D/err.$name; }
cat $D/r1.fq $D/r2.fq > /dev/null
for rep in 1 2; do
run one_ctx hast_amd/classify $ARGS -t 32 --stats
run devices_0_0 hast_amd/classify $ARGS -t 32 --stats --devices 0,0
run devices_0_0_0_0 hast_amd/classify $ARGS -t 32 --stats --devices 0,0,0,0
done
run one_file_one_ctx hast_amd/classify --hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq -t 32 --stats
run one_file_4_ctx hast_amd/classify --hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq -t 32 --stats --devices 0,0,0,0
rm -rf $D

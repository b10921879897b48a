# round-3 measurements on one box: the default bench line, the other workloads, the 2.9-B-read run, and the CLI on 20M reads
# with the blocks of each file on one context and striped over 2 / 4 contexts of the same GPU (what --devices costs when the
# GPUs are one: framing from the newline count, the overlap upload, one more counting pass)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
python bench.py > $O/round3_bench_default.json 2> $O/round3_bench_default.err; tail -2 $O/round3_bench_default.err
python bench.py --cpu-seconds 0 --workload c5 > $O/round3_bench_c5.json 2>/dev/null
python bench.py --cpu-seconds 0 --workload c2 > $O/round3_bench_c2.json 2>/dev/null
python bench.py --cpu-seconds 0 --clustered > $O/round3_bench_c3_clustered.json 2>/dev/null
python bench.py --cpu-seconds 0 --steps 60 > $O/round3_bench_c3_2p9Breads.json 2>/dev/null
HAST_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --cpu-seconds 0 > $O/round3_bench_c3_one_rank_rccl.json 2>/dev/null
python bench.py --workload s00 > $O/round3_bench_s00.json 2>/dev/null
for f in default c5 c2 c3_clustered c3_2p9Breads c3_one_rank_rccl s00; do python3 -c "
import json; d=json.load(open('$O/round3_bench_$f.json')); r=d['roofline']; print('$f', round(d['value']/1e9,1), 'Gbp/s', round(d['ms_per_step'],2), 'ms/step kernel', round(r.get('kernel_ms_avg',0) or 0,2), 'frac', r.get('frac'), 'req', (r.get('request_rate') or {}).get('frac_of_ceiling_this_run'), 'allreduce_ms', d.get('allreduce_ms'))"; done
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq --read $D/r2.fq --weight0 1.04"
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats__ $D/err.$name | sed "s/.*load_s/load_s/") $(grep -h __stats_read_phase__ $D/err.$name | cut -d" " -f2- | cut -c1-140) $(grep -h __stats_devices__ $D/err.$name | cut -d" " -f2-)"; }
cat $D/r1.fq $D/r2.fq > /dev/null
for rep in 1 2; do
run one_ctx hast_amd/classify $ARGS -t 32 --stats
run devices_0_0 hast_amd/classify $ARGS -t 32 --stats --devices 0,0
run devices_0_0_0_0 hast_amd/classify $ARGS -t 32 --stats --devices 0,0,0,0
HAST_DEAL=files run files_0_0 hast_amd/classify $ARGS -t 32 --stats --devices 0,0
done
run one_file_one_ctx hast_amd/classify --hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq -t 32 --stats
run one_file_4_ctx hast_amd/classify --hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq -t 32 --stats --devices 0,0,0,0
rm -rf $D

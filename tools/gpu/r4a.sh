# round 4, first GPU pass: the whole -m gpu suite on the 64-bit counters / pinned switches, the default bench line with its
# long steps + secondary object, and the CLI's phase accounting (--stats) on 20M reads, plain and gz
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r4a_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/r4a_pytest.log)"
timeout -k 10 600 python bench.py > $O/r4a_bench_default.json 2> $O/r4a_bench_default.err; echo "bench rc=$?"; tail -3 $O/r4a_bench_default.err
python3 -c "
import json; d=json.load(open('$O/r4a_bench_default.json')); r=d['roofline']; print(round(d['value']/1e9,1), 'Gbp/s', round(d['ms_per_step'],2), 'ms/step kernel', round(r.get('kernel_ms_avg',0) or 0,2), 'commit', round(r.get('commit_kernel_ms_avg',0),3), 'frac', r.get('frac'), 'req', (r.get('request_rate') or {}).get('frac_of_ceiling_this_run'), 'secondary', (d.get('secondary') or {}).get('value'))"
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_read_phase__ $D/err.$name | cut -d" " -f2- | cut -c1-150)"; grep -h "__stats_phases__\|__stats_switches__" $D/err.$name; }
cat $D/r1.fq $D/r2.fq > /dev/null
for rep in 1 2; do run plain_t32 hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats; done
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
ls -la $D/r1.fq.gz
for rep in 1 2; do run gz6_default hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
rm -rf $D

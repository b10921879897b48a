# round 4, final tree: the whole -m gpu suite, the stage-00 and default bench lines, and `classify` on 20M reads as .gz (inflated on the
# GPU / on the host; gzip -6 with constant and noisy quality lines, gzip -1) and plain -- unedited output -> profiles/round4_final.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/round4_final_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/round4_final_pytest.log)"
python bench.py --workload s00 > $O/round4_bench_s00.json 2>/dev/null
python bench.py > $O/round4_final_bench_default.json 2>/dev/null
for f in round4_bench_s00 round4_final_bench_default; do python3 -c "
import json; d=json.load(open('$O/$f.json')); r=d['roofline']; print('$f', round(d['value']/1e9,1), 'Gbp/s; roofline.frac', r.get('frac'), '; traffic', r.get('traffic'), '; useful', json.dumps(r.get('useful'))[:200])"; done
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_read_phase__ $D/err.$name | cut -d" " -f2- | cut -c1-150)"; grep -h "__stats_phases__" $D/err.$name | cut -c1-420; }
for q in const noisy; do
  if [ $q = noisy ]; then export GEN_FASTQ_QUAL=noisy; fi
  tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
  (gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
  echo "== quality lines: $q; $(stat -c %s $D/r1.fq) bytes per file, $(stat -c %s $D/r1.fq.gz) as gzip -6"
  cat $D/r1.fq $D/r2.fq $D/r1.fq.gz $D/r2.fq.gz > /dev/null
  for rep in 1 2; do run ${q}_plain$rep hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats; done
  for rep in 1 2 3 4; do run ${q}_gz6_device$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
  for rep in 1 2; do HAST_INFLATE=host run ${q}_gz6_host$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
  if [ $q = const ]; then
    (gzip -1 -c $D/r1.fq > $D/r1.l1.fq.gz & gzip -1 -c $D/r2.fq > $D/r2.l1.fq.gz & wait)
    for rep in 1 2; do run const_gz1_device$rep hast_amd/classify $ARGS --read $D/r1.l1.fq.gz --read $D/r2.l1.fq.gz -t 32 --stats; done
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/round4_prof_gz -- hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats > $D/out.prof 2> $D/err.prof
    echo "under rocprofv3: md5=$(md5sum < $D/out.prof | cut -c1-12) $(grep -h __stats_phases__ $D/err.prof | cut -c1-300)"
  fi
done
rm -rf $D

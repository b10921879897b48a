# round-4 measurements on one box (unedited output -> profiles/round4_measure.txt): the bench lines of every workload, the CLI on 20M
# reads (plain, .gz inflated on the GPU, .gz inflated on the host, gzip -1, noisy quality lines; --devices on one GPU) with the
# rocprofv3 kernel stats of one device-inflate run, and stage 00 through its program
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python bench.py > $O/round4_bench_default.json 2> $O/round4_bench_default.err; tail -2 $O/round4_bench_default.err
python bench.py --cpu-seconds 0 --workload c5 > $O/round4_bench_c5.json 2>/dev/null
python bench.py --cpu-seconds 0 --workload c2 > $O/round4_bench_c2.json 2>/dev/null
python bench.py --cpu-seconds 0 --clustered > $O/round4_bench_c3_clustered.json 2>/dev/null
python bench.py --workload s00 > $O/round4_bench_s00.json 2>/dev/null
HAST_KC_COUNT=atomic python bench.py --workload s00 --cpu-seconds 0 > $O/round4_bench_s00_atomic.json 2>/dev/null
for f in default c5 c2 c3_clustered s00 s00_atomic; do python3 -c "
import json; d=json.load(open('$O/round4_bench_$f.json')); r=d['roofline']; print('$f', round(d['value']/1e9,1), 'Gbp/s', round(d['ms_per_step'],2) if 'ms_per_step' in d else d.get('seconds'), 'ms/step; kernel', round(r.get('kernel_ms_avg',0) or 0,2), 'ms; roofline.frac', r.get('frac'), '; of the request ceiling', (r.get('request_rate') or {}).get('frac_of_ceiling_this_run'), '; useful', json.dumps(r.get('useful'))[:300])"; done
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_read_phase__ $D/err.$name | cut -d" " -f2- | cut -c1-150)"; grep -h "__stats_phases__\|__stats_gz__\|__stats_devices__" $D/err.$name | cut -c1-420; }
for q in const noisy; do
  if [ $q = noisy ]; then export GEN_FASTQ_QUAL=noisy; fi
  tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
  (gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
  echo "== quality lines: $q; $(stat -c %s $D/r1.fq) bytes per file, $(stat -c %s $D/r1.fq.gz) as gzip -6"
  cat $D/r1.fq $D/r2.fq $D/r1.fq.gz $D/r2.fq.gz > /dev/null
  for rep in 1 2 3; do run ${q}_plain$rep hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats; done
  for rep in 1 2 3 4; do run ${q}_gz6_device$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
  for rep in 1 2; do HAST_INFLATE=host run ${q}_gz6_host$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
  if [ $q = const ]; then
    (gzip -1 -c $D/r1.fq > $D/r1.l1.fq.gz & gzip -1 -c $D/r2.fq > $D/r2.l1.fq.gz & wait)
    for rep in 1 2; do run const_gz1_device$rep hast_amd/classify $ARGS --read $D/r1.l1.fq.gz --read $D/r2.l1.fq.gz -t 32 --stats; done
    for rep in 1 2; do
      run devices_0_0 hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats --devices 0,0
      run devices_0_0_0_0 hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats --devices 0,0,0,0
    done
    run gz6_devices_0_0 hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats --devices 0,0
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/round4_prof_gz -- hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats > $D/out.prof 2> $D/err.prof
    echo "under rocprofv3: md5=$(md5sum < $D/out.prof | cut -c1-12) $(grep -h __stats_gz__ $D/err.prof | cut -c1-300)"
  fi
done
rm -rf $D
timeout -k 10 400 bash tests/e2e/s00_e2e.sh 20000000 30 2 round4_s00_20Mbp > $O/round4_s00_e2e.log 2>&1; echo "s00 e2e rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/cli_e2e_round4_s00_20Mbp.json"))
for r in d["runs"]:
    print(r["name"], r["seconds"], "s", r["Mbp_per_s"], "Mbp/s", r["md5_paternal"][:8], r["md5_maternal"][:8], r["md5_histo"][:8], (r["stats"][-1:] or [""])[0][:160])
PY

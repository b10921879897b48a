# ONE command for the first node with more than one MI355X (VERDICT r5 "next" #7): everything of the N > 1 path that has
# only ever run with both ends on ONE GPU -- the RCCL group call with n > 1 communicators, hipMemcpyPeerAsync / event waits between devices in
# the striped framer and in hast_gz_open_multi, the merge by text of several GPUs' barcode dictionaries, `--devices` of all three programs,
# bench.py over N ranks -- each with an md5 against the 1-GPU output of the same input, and one summary table at the end.
#   bash tools/gpu/first_multi_gpu.sh            N = the GPUs this box shows (N = 1: the same commands over one device: must be green too)
#   N=4 bash tools/gpu/first_multi_gpu.sh        the first 4 of them
#   SKIP_SUITE=1 / SKIP_BENCH=1 / SKIP_CLI=1     leave a part out;   BENCH_GPUS="2 4 8"   the rank counts bench.py is run with
# Not a scaling measurement (the driver's SCALE run is that): a does-it-work-and-agree run.  Results in gpurun_out/first_multi_gpu.txt.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
O=gpurun_out; mkdir -p $O
ROOT=$PWD
HAVE=$(python3 -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 1)
N=${N:-$HAVE}; [ "$N" -gt "$HAVE" ] && { echo "first_multi_gpu.sh: N=$N but this box shows $HAVE GPUs"; exit 2; }
DEVS=$(seq -s, 0 $((N-1)))
SUMMARY=$(mktemp /tmp/fmg_summary.XXXXXX)
row() { printf "%-46s %-8s %s\n" "$1" "$2" "$3" | tee -a $SUMMARY; }
now() { date +%s.%N; }
el() { python3 -c "print(round($2-$1,2))"; }
echo "== $N of $HAVE GPUs: devices $DEVS"

# 1. the suite: on a box with >= 2 GPUs the "0,1" / two-GPU parametrisations are no longer skipped
if [ -z "$SKIP_SUITE" ]; then
  t0=$(now); timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/fmg_pytest.log 2>&1; rc=$?; t1=$(now)
  row "pytest -m gpu (two-GPU cases included)" "$([ $rc = 0 ] && echo ok || echo FAIL)" "$(tail -1 $O/fmg_pytest.log) [$(el $t0 $t1) s]"
fi

# 2. bench.py: the BASELINE metric and config 5 over 1 .. N ranks (one process per GPU, one all-reduce over RCCL / xGMI)
if [ -z "$SKIP_BENCH" ]; then
  for wl in default c5; do
    for g in 1 ${BENCH_GPUS:-$(for x in 2 4 8; do [ $x -le $N ] && echo $x; done)}; do
      [ $g -gt $N ] && continue
      flag=""; [ $wl != default ] && flag="--workload $wl"
      python bench.py --gpus $g $flag > $O/fmg_bench_${wl}_$g.json 2> $O/fmg_bench_${wl}_$g.err; rc=$?
      row "bench.py $flag --gpus $g" "$([ $rc = 0 ] && echo ok || echo FAIL)" "$(python3 -c "
import json
try:
    d = json.load(open('$O/fmg_bench_${wl}_$g.json')); print('%.1f Gbp/s, n_gpus %d, %.1f ms/step, allreduce %s ms' % (d['value'] / 1e9, d['n_gpus'], d['ms_per_step'], d.get('allreduce_ms')))
except Exception as e: print('no JSON line:', e)")"
    done
  done
fi

# 3. the three programs with --devices, md5 against one GPU
if [ -z "$SKIP_CLI" ]; then
  D=$(mktemp -d /dev/shm/hast_fmg.XXXXXX); trap 'rm -rf $D $SUMMARY' EXIT
  tools/gen_fastq $D ${NPAIRS:-10000000} 5000000 1000000 21 150 32 0 || exit 1
  tools/pgzip1 $D/r1.fq $D/r1.fq.gz 6 16 32 && tools/pgzip1 $D/r2.fq $D/r2.fq.gz 6 16 32 || exit 1
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 -t 32 --stats"
  cl() { local name=$1; shift; local w=$D/w.$name; mkdir -p $w; local t0=$(now); (cd $w && $ROOT/hast_amd/classify $ARGS "$@" > $D/out.$name 2> $D/err.$name); RC=$?; local t1=$(now)
    MD5=$(md5sum < $D/out.$name | cut -c1-12); T=$(el $t0 $t1)
    PER=$(grep -h __stats_devices__ $D/err.$name | sed 's/.*records_per_context=//')
    ROUTED=$(cd $w && cat *.fastq filter_reads.log *.barcodes 2>/dev/null | md5sum | cut -c1-12); rm -rf $w; }
  for kind in plain gz; do
    R="--read $D/r1.fq --read $D/r2.fq"; [ $kind = gz ] && R="--read $D/r1.fq.gz --read $D/r2.fq.gz"
    cl ${kind}_one $R; ref=$MD5; row "classify $kind, one GPU" "$([ $RC = 0 ] && echo ok || echo FAIL)" "md5 $MD5 [$T s]"
    cl ${kind}_devs $R --devices $DEVS
    row "classify $kind --devices $DEVS" "$([ $RC = 0 ] && [ $MD5 = $ref ] && echo ok || echo FAIL)" "md5 $MD5, records per context $PER [$T s]"
    cl ${kind}_devs_files $R --devices $DEVS --deal files
    row "classify $kind --devices $DEVS --deal files" "$([ $RC = 0 ] && [ $MD5 = $ref ] && echo ok || echo FAIL)" "md5 $MD5 [$T s]"
    cl ${kind}_route_one $R --phase-reads; rref=$ROUTED
    cl ${kind}_route_devs $R --phase-reads --devices $DEVS
    row "classify $kind --phase-reads --devices $DEVS" "$([ $RC = 0 ] && [ $MD5 = $ref ] && [ $ROUTED = $rref ] && echo ok || echo FAIL)" "routed files md5 $ROUTED (one GPU: $rref) [$T s]"
  done
  # every GPU twice in the list: contexts that share a device are summed there first, then the all-reduce over the distinct devices
  DD=$(for d in $(seq 0 $((N-1))); do printf "%s,%s," $d $d; done | sed 's/,$//')
  cl gz_doubled --read $D/r1.fq.gz --read $D/r2.fq.gz --devices $DD
  row "classify gz --devices $DD" "$([ $RC = 0 ] && [ $MD5 = $ref ] && echo ok || echo FAIL)" "md5 $MD5, records per context $PER [$T s]"
  # config 5's boundary: classify_read over the GPUs
  E=$(mktemp -d /dev/shm/hast_fmg5.XXXXXX); trap 'rm -rf $D $E $SUMMARY' EXIT
  tools/gen_fastq $E 20000 2000000 1 31 20000 32 0 || exit 1
  RA="--hap $E/hap0.mer --hap $E/hap1.mer --read $E/r1.fq --format fastq --thread 32"
  hast_amd/classify_read $RA > $E/out.one 2> $E/err.one; r1=$?; hast_amd/classify_read $RA --devices $DEVS > $E/out.devs 2> $E/err.devs; r2=$?
  row "classify_read --devices $DEVS (20k reads of 20 kb)" "$([ $r1 = 0 ] && [ $r2 = 0 ] && cmp -s $E/out.one $E/out.devs && echo ok || echo FAIL)" "md5 $(md5sum < $E/out.devs | cut -c1-12) (one GPU: $(md5sum < $E/out.one | cut -c1-12))"
  # stage 00: the key space split over the GPUs, no exchange at all
  tools/gen_trio $E 20000000 30 150 2 32 || exit 1
  UA=""; for i in 0 1; do UA="$UA --paternal $E/paternal_$i.fq --maternal $E/maternal_$i.fq"; done
  mkdir -p $E/u1 $E/uN
  (cd $E/u1 && $ROOT/hast_amd/unshared_kmers $UA --thread 8 --auto_bounds > out.txt 2> err.txt); r1=$?
  (cd $E/uN && $ROOT/hast_amd/unshared_kmers $UA --thread 8 --auto_bounds --devices $DEVS > out.txt 2> err.txt); r2=$?
  m1=$(cat $E/u1/*.mer $E/u1/*.histo | md5sum | cut -c1-12); mN=$(cat $E/uN/*.mer $E/uN/*.histo | md5sum | cut -c1-12)
  row "unshared_kmers --devices $DEVS (20-Mbp trio at 30x)" "$([ $r1 = 0 ] && [ $r2 = 0 ] && [ $m1 = $mN ] && echo ok || echo FAIL)" "products md5 $mN (one GPU: $m1)"
fi
echo; echo "== summary ($N GPUs)"; cat $SUMMARY
grep -q FAIL $SUMMARY && exit 1
exit 0

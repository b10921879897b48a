# round 6: a change to k_gz_decode against the tree in front of it (AB_OLD=tools/ab_old: libhast.so + classify built from that tree) on ONE box:
# the device inflate's tests first (corpus == zlib, damaged inputs, rings), then 20M reads as two single-member gzip -6 files, constant and noisy
# quality lines: runs alternating new / old (read phase, decode_s) and the kernel statistics of one run of each under rocprofv3.
# usage: gpurun -- 'bash tools/gpu/gz_decode_ab.sh > gpurun_out/gz_decode_ab.txt 2>&1'        SKIP_TESTS=1: the A/B only
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
OLD=${AB_OLD:-tools/ab_old}
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 700 python -m pytest tests/test_gz_gpu.py -x -q > $O/gz_decode_ab_pytest.log 2>&1; rc=$?; echo "pytest tests/test_gz_gpu.py rc=$rc $(tail -1 $O/gz_decode_ab_pytest.log)"
  [ $rc = 0 ] || { tail -30 $O/gz_decode_ab_pytest.log; exit 1; }
fi
D=$(mktemp -d /dev/shm/hast_dab.XXXXXX); trap 'rm -rf $D' EXIT
run() { local name=$1; shift; "$@" > $D/out.$name 2> $D/err.$name; local rc=$?
  echo "$name rc=$rc md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_phases__ $D/err.$name | grep -o "read_phase_s=[0-9.]*\|total_s=[0-9.]*" | tr '\n' ' ') $(grep -h __stats_gz__ $D/err.$name | grep -o "decode_s=[0-9.]*\|followup_jobs=[0-9]*" | tr '\n' ' ')"; }
stats() { python3 - "$1" <<'PYEOF'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "gz::" in n: print("     %-16s calls %5s total %8.1f ms avg %8.3f ms max %8.3f ms" % (n.split("(")[0].split("::")[-1], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, int(r["MaxNs"]) / 1e6))
PYEOF
}
for q in ${QUALS:-const noisy}; do
  [ $q = noisy ] && export GEN_FASTQ_QUAL=noisy
  tools/gen_fastq $D 10000000 5000000 100000 21 150 32 0 || exit 1
  tools/pgzip1 $D/r1.fq $D/r1.fq.gz 6 16 32; tools/pgzip1 $D/r2.fq $D/r2.fq.gz 6 16 32
  echo "== quality lines: $q; $(stat -c %s $D/r1.fq) bytes per file, $(stat -c %s $D/r1.fq.gz) as one gzip -6 member"
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 -t 32 --stats --read $D/r1.fq.gz --read $D/r2.fq.gz"
  for rep in 1 2 3; do
    run ${q}_new_$rep hast_amd/classify $ARGS
    run ${q}_old_$rep $OLD/classify $ARGS
  done
  for v in new old; do
    exe=hast_amd/classify; [ $v = old ] && exe=$OLD/classify
    rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_$v -- $exe $ARGS > $D/out.prof_$v 2> $D/err.prof_$v
    f=$(ls $D/prof_$v/*/*kernel_stats.csv | head -1)
    echo "-- kernel stats, $v ($q): md5=$(md5sum < $D/out.prof_$v | cut -c1-12) $(grep -o "read_phase_s=[0-9.]*" $D/err.prof_$v)"; stats $f
    cp $f $O/gz_decode_ab_${q}_${v}_kernel_stats.csv; rm -rf $D/prof_$v
  done
done

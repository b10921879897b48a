# round 6: two passes of a file on the GPU together (HAST_GZ_AHEAD=1: two decode streams, three arenas) against one at a time (=0), same binary,
# one box: the inflate tests in both modes, then 20M reads as two single-member gzip -6 files (constant and noisy quality lines), runs alternating,
# and for each mode one traced run: sum and union of the decode launches' intervals.  ONE_FILE=1 adds runs with r1 as .gz and r2 plain.
# usage: gpurun -- 'bash tools/gpu/gz_ahead_ab.sh > gpurun_out/gz_ahead_ab.txt 2>&1'        SKIP_TESTS=1: the A/B only
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
  for a in 1 0; do
    HAST_GZ_AHEAD=$a timeout -k 10 500 python -m pytest tests/test_gz_gpu.py -x -q > $O/gz_ahead_pytest_$a.log 2>&1; rc=$?; echo "HAST_GZ_AHEAD=$a pytest tests/test_gz_gpu.py rc=$rc $(tail -1 $O/gz_ahead_pytest_$a.log)"
    [ $rc = 0 ] || { tail -40 $O/gz_ahead_pytest_$a.log; exit 1; }
  done
fi
D=$(mktemp -d /dev/shm/hast_aab.XXXXXX); trap 'rm -rf $D' EXIT
run() { local name=$1; shift; "$@" > $D/out.$name 2> $D/err.$name; local rc=$?
  echo "$name rc=$rc md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_phases__ $D/err.$name | grep -o "read_phase_s=[0-9.]*\|total_s=[0-9.]*" | tr '\n' ' ') $(grep -h __stats_gz__ $D/err.$name | grep -o "followup_jobs=[0-9]*" | tr '\n' ' ') hbm=$(grep -o "in_use_peak_bytes=[0-9]*" $D/err.$name)"; }
for q in ${QUALS:-const noisy}; do
  [ $q = noisy ] && export GEN_FASTQ_QUAL=noisy
  tools/gen_fastq $D ${NPAIRS:-10000000} 5000000 100000 21 150 32 0 || exit 1
  tools/pgzip1 $D/r1.fq $D/r1.fq.gz 6 16 32; tools/pgzip1 $D/r2.fq $D/r2.fq.gz 6 16 32
  echo "== quality lines: $q; $(stat -c %s $D/r1.fq) bytes per file, $(stat -c %s $D/r1.fq.gz) as one gzip -6 member"
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 -t 32 --stats"
  for rep in 1 2 3 4; do
    for a in 1 0; do HAST_GZ_AHEAD=$a run ${q}_ahead${a}_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz; done
  done
  if [ -n "$ONE_FILE" ]; then
    for rep in 1 2 3; do
      for a in 1 0; do HAST_GZ_AHEAD=$a run ${q}_onegz_ahead${a}_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq; done
    done
  fi
  for a in 1 0; do
    export HAST_GZ_AHEAD=$a
    rocprofv3 --kernel-trace --output-format csv -d $D/prof -- hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz > $D/out.p 2> $D/err.p
    unset HAST_GZ_AHEAD
    echo "-- traced, HAST_GZ_AHEAD=$a ($q): md5=$(md5sum < $D/out.p | cut -c1-12) $(grep -o "read_phase_s=[0-9.]*" $D/err.p)"
    python3 - $(ls $D/prof/*/*kernel_trace.csv | head -1) <<'PYEOF'
import csv, sys, collections
iv = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].split("::")[-1].replace("void ", "")
    iv[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
def union(v):
    v = sorted(v); tot = 0; cs, ce = v[0]
    for s, e in v[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
for n in ("k_gz_decode", "k_gz_search"):
    v = iv.get(n)
    if v: print("     %-14s launches %4d  sum %8.1f ms  union %8.1f ms  first start to last end %8.1f ms" % (n, len(v), sum(e - s for s, e in v) / 1e6, union(v) / 1e6, (max(e for s, e in v) - min(s for s, e in v)) / 1e6))
PYEOF
    rm -rf $D/prof
  done
done

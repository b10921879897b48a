# The boundary at BASELINE config 2's size (VERDICT r5 "next" #1): `classify` on 200M reads of 150 bp as two FASTQ files, 50M + 50M 21-mers as
# k-mer TEXT files (so that the CLI itself builds the exact-entries filter), 1M barcodes -- plain, as two single-member .gz files of > 2 GB
# each (the device inflate's ring engages at its default), over four contexts, and with --phase-reads (the wrapper's steps 10-11 on the GPU).
# Reference shape: classify.cpp:238-278 streams inputs of any size; HAST.sh:162-166 hands it two .fq.gz; classify_stlfr_reads.sh:148-185.
# stdout md5 against the oracle's multi-threaded program over the same files and against the REAL reference binary on a subsample.
# The files live in /dev/shm (the box's / has 79 GB; /dev/shm is RAM, the job's memory limit is 300 GB): 68 GB of FASTQ + 13 GB of .gz
# + 68 GB of routed output at a time.
# A gpurun call lasts 20 minutes at most and nothing stays on the box between calls, so the job comes in two halves over the same
# (deterministic) files:   STEPS="runs route prof"  the measurements;   STEPS="check1" / STEPS="check2"  the checkers: the oracle's
# multi-threaded program takes ~13 minutes per 100M reads on the box's 16-core quota, so it is run over ONE input file per call and
# compared with `classify` on that file; that the two files' counts add up to the run over both is checked with merge_counts (check1);
# check2 also runs the REAL reference binary on a 2M-read subsample.
# usage: gpurun --timeout 1200 -- 'STEPS="runs route prof" bash tools/gpu/cli_c2.sh > gpurun_out/round6_cli_c2.txt 2>&1'
#        gpurun --timeout 1200 -- 'STEPS=check1 bash tools/gpu/cli_c2.sh > gpurun_out/round6_cli_c2_check1.txt 2>&1'      (and check2)
#   NPAIRS (default 100000000 = 200M reads), KEYS (50000000 per haplotype), BARCODES (1000000), LEVEL (6)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
NPAIRS=${NPAIRS:-100000000}; KEYS=${KEYS:-50000000}; BARCODES=${BARCODES:-1000000}; LEVEL=${LEVEL:-6}
STEPS=${STEPS:-runs route prof}
has() { case " $STEPS " in *" $1 "*) return 0;; esac; return 1; }
D=$(mktemp -d /dev/shm/hast_c2.XXXXXX)
( while sleep 45; do echo "[still running $(date +%T)]"; done ) &
HB=$!
trap 'kill $HB 2>/dev/null; rm -rf $D' EXIT
now() { date +%s.%N; }
el() { python3 -c "print(round($2-$1,2))"; }
echo "== box: $(nproc) hardware threads, cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), memory.max $(cat /sys/fs/cgroup/memory.max 2>/dev/null), /dev/shm $(df -h /dev/shm | tail -1 | awk '{print $4}') free"
t0=$(now); GEN_FASTQ_MAX_GB=90 tools/gen_fastq $D $NPAIRS $KEYS $BARCODES 21 150 32 0 || exit 1; t1=$(now)
echo "== generated in $(el $t0 $t1) s: $((2*NPAIRS)) reads of 150 bp, $KEYS + $KEYS 21-mers, $BARCODES barcodes; $(stat -c %s $D/r1.fq) bytes per FASTQ file, $(stat -c %s $D/hap0.mer) per k-mer file"
if has runs || has route || has prof || has blocks || has ab || has abname || has abcus || has abids || has abahead || has paused; then
t0=$(now); tools/pgzip1 $D/r1.fq $D/r1.fq.gz $LEVEL 16 32 && tools/pgzip1 $D/r2.fq $D/r2.fq.gz $LEVEL 16 32 || exit 1; t1=$(now)
echo "== compressed in $(el $t0 $t1) s (tools/pgzip1 level $LEVEL: ONE gzip member per file): $(stat -c %s $D/r1.fq.gz) + $(stat -c %s $D/r2.fq.gz) bytes"
fi
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 -t 32 --stats"
run() { local name=$1; shift; local w=$D/w.$name; mkdir -p $w; local t0=$(now); (cd $w && "$@" > $D/out.$name 2> $D/err.$name); local rc=$?; local t1=$(now)
  echo "-- $name rc=$rc whole process $(el $t0 $t1) s rows=$(wc -l < $D/out.$name) stdout md5=$(md5sum < $D/out.$name | cut -c1-12)"
  grep -h "__stats_phases__\|__stats_read_phase__\|__stats_gz__\|__stats_devices__\|__stats_phase_reads__\|__stats_hbm__\|__stats_filter__\|__stats__ \|WARN\|ERROR" $D/err.$name | cut -c1-700 | sed 's/^/     /'
  python3 - "$D/err.$name" <<'EOF'
import re, sys
t = open(sys.argv[1]).read()
m = re.search(r"__stats__ .*bases=(\d+)", t); p = re.search(r"read_phase_s=([0-9.]+)", t)
if m and p and float(p.group(1)) > 0:
    print("     read phase: %.2f Gbp/s (%d bases in %s s)" % (int(m.group(1)) / float(p.group(1)) / 1e9, int(m.group(1)), p.group(1)))
EOF
  if ls $w/*.fastq > /dev/null 2>&1; then
    echo "     routed files: $(cd $w && ls -l *.fastq | awk '{printf "%s=%s ", $9, $5}')"; echo "     filter_reads.log: $(tr '\n' '|' < $w/filter_reads.log)"
    echo "     lists: $(cd $w && wc -l *.barcodes | tr '\n' ' ')"
  fi
  if [ -n "$KEEP" ] && [ "$KEEP" = "$name" ]; then :; else rm -rf $w; fi; }
PY=$PWD/hast_amd/classify
run plain $PY $ARGS --read $D/r1.fq --read $D/r2.fq
for half in 1 2; do
  if has check$half; then
    run only_r$half $PY $ARGS --read $D/r$half.fq
    cp $D/out.only_r$half $D/half$half.tsv
    if [ $half = 1 ]; then
      # additivity: the rows of r1 alone + the rows of r2 alone, merged by barcode, must be the rows of the run over both
      run only_r2 $PY $ARGS --read $D/r2.fq
      n0=$(grep -o "set0=[0-9]*" $D/err.plain | head -1 | cut -d= -f2); n1=$(grep -o "set1=[0-9]*" $D/err.plain | head -1 | cut -d= -f2)
      hast_amd/merge_counts --set0 $n0 --set1 $n1 --weight0 1.04 $D/out.only_r1 $D/out.only_r2 > $D/out.merged 2> $D/err.merged
      echo "-- merge_counts(r1 alone, r2 alone) rc=$? rows=$(wc -l < $D/out.merged) md5=$(md5sum < $D/out.merged | cut -c1-12)   (the run over both files: $(md5sum < $D/out.plain | cut -c1-12))"
    fi
    t0=$(now); oracle/oracle_classify --hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 --read $D/r$half.fq -t 32 > $D/out.oracle 2> /dev/null; t1=$(now)
    echo "-- oracle_classify -t 32 over r$half.fq ($NPAIRS reads): $(el $t0 $t1) s rows=$(wc -l < $D/out.oracle) md5=$(md5sum < $D/out.oracle | cut -c1-12)   (classify on the same file: $(md5sum < $D/half$half.tsv | cut -c1-12))"
  fi
done
if has runs; then
run plain_again $PY $ARGS --read $D/r1.fq --read $D/r2.fq
run gz $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz
run gz_again $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz
run gz_devices_0_0_0_0 $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz --devices 0,0,0,0
run gz_host_inflate $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz --inflate host
fi
if has blocks; then          # the ingest block size (default 16 MB): what a block costs besides its bytes -- a dozen small kernels and two host round trips
  for mb in 16 32 64 128; do
    run plain_block_${mb}mb $PY $ARGS --read $D/r1.fq --read $D/r2.fq --block-mb $mb
    run gz_block_${mb}mb $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz --block-mb $mb
  done
fi
if has ab; then              # the block size that follows the input's size (default) against 16 MB, alternating, with the cgroup's CPU throttle counters
  thr() { awk '/nr_throttled|throttled_usec/ {printf "%s=%s ", $1, $2}' /sys/fs/cgroup/cpu.stat 2>/dev/null; }
  for rep in 1 2 3; do
    echo "   cpu.stat before: $(thr)"
    run gz_auto_$rep $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz
    echo "   cpu.stat: $(thr)"
    run gz_16mb_$rep $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz --block-mb 16
    echo "   cpu.stat: $(thr)"
    run plain_auto_$rep $PY $ARGS --read $D/r1.fq --read $D/r2.fq
    echo "   cpu.stat: $(thr)"
    run plain_16mb_$rep $PY $ARGS --read $D/r1.fq --read $D/r2.fq --block-mb 16
    echo "   cpu.stat: $(thr)"
    run plain_auto_t8_$rep $PY ${ARGS/-t 32/-t 8} --read $D/r1.fq --read $D/r2.fq
    echo "   cpu.stat: $(thr)"
  done
fi
if has abname; then          # a dictionary names a block behind its framing (HAST_NAME_EARLY=1, tried in round 6) against in hast_fq_next on the context's stream (the default)
  for rep in 1 2 3; do
    HAST_NAME_EARLY=1 run gz_name_early_$rep $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz
    HAST_NAME_EARLY=0 run gz_name_in_next_$rep $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz
    HAST_NAME_EARLY=1 run plain_name_early_$rep $PY $ARGS --read $D/r1.fq --read $D/r2.fq
    HAST_NAME_EARLY=0 run plain_name_in_next_$rep $PY $ARGS --read $D/r1.fq --read $D/r2.fq
  done
fi
if has abcus; then           # CUs the decode passes leave to the kernels behind them (HAST_GZ_FREE_CUS, default 32), alternating
  for rep in ${CUS_REPS:-1 2 3}; do
    for cus in ${CUS_LIST:-32 16 0 64}; do
      HAST_GZ_FREE_CUS=$cus run gz_free_cus_${cus}_$rep $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz
    done
  done
fi
if has paused; then          # every run `sleep $PAUSE` (5 s) behind the exit of the one in front: what a process costs that does not start while the driver takes the last one apart (profiles/round6_gz_slow_hunt.txt)
  for rep in 1 2 3 4; do
    sleep ${PAUSE:-5}; run plain_paused_$rep $PY $ARGS --read $D/r1.fq --read $D/r2.fq
    sleep ${PAUSE:-5}; run gz_paused_$rep $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz
  done
  for rep in 1 2; do
    run plain_back_to_back_$rep $PY $ARGS --read $D/r1.fq --read $D/r2.fq
    run gz_back_to_back_$rep $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz
  done
fi
if has abahead; then         # two passes of a file on the GPU together (HAST_GZ_AHEAD=1, the default since round 6's last day) against one at a time (=0), alternating
  for rep in ${AHEAD_REPS:-1 2 3 4}; do
    HAST_GZ_AHEAD=1 run gz_ahead1_$rep $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz
    HAST_GZ_AHEAD=0 run gz_ahead0_$rep $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz
  done
  HAST_GZ_AHEAD=1 run gz_ahead1_devices_0_0 $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz --devices 0,0
  HAST_GZ_AHEAD=1 run gz_ahead1_one_gz $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq
  HAST_GZ_AHEAD=0 run gz_ahead0_one_gz $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq
fi
if has abids; then           # the bookkeeping kernel reads a block's ids in device memory (default, round 6) against out of pinned host memory (HAST_COMMIT_IDS=host)
  for rep in 1 2 3; do
    run gz_ids_device_$rep $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz
    HAST_COMMIT_IDS=host run gz_ids_host_$rep $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz
    run plain_ids_device_$rep $PY $ARGS --read $D/r1.fq --read $D/r2.fq
    HAST_COMMIT_IDS=host run plain_ids_host_$rep $PY $ARGS --read $D/r1.fq --read $D/r2.fq
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/round6_prof_c2 -- $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz > $D/out.prof 2> $D/err.prof
  f=$(ls $O/round6_prof_c2/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/round6_cli_c2_gz_kernel_stats_ids_on_device.csv && head -12 $f | cut -c1-200
  rm -rf $O/round6_prof_c2
fi
if has route; then
  KEEP=plain_route run plain_route $PY $ARGS --read $D/r1.fq --read $D/r2.fq --phase-reads
  # the routed files against the inputs: every record went somewhere (sizes add up), and one class against a grep of the input
  w=$D/w.plain_route
  for f in r1.fq r2.fq; do
    tot=0; for c in nobarcode paternal maternal homozygous; do [ -f $w/$f.$c.fastq ] && tot=$((tot + $(stat -c %s $w/$f.$c.fastq))); done
    echo "     $f: routed bytes $tot of $(stat -c %s $D/$f)"
  done
  rm -rf $w
  run gz_route $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz --phase-reads
  run gz_route_devices_0_0_0_0 $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz --phase-reads --devices 0,0,0,0
  if [ -n "$ROUTE_HOST" ]; then run plain_route_host $PY $ARGS --read $D/r1.fq --read $D/r2.fq --phase-reads --route host; fi
fi
# kernel statistics of one .gz run (rocprofv3 on the program itself)
if has prof; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/round6_prof_c2 -- $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz > $D/out.prof 2> $D/err.prof
echo "-- under rocprofv3 (gz): rc=$? md5=$(md5sum < $D/out.prof | cut -c1-12) $(grep -h __stats_phases__ $D/err.prof | cut -c1-300)"
grep -v "^[EW]2026" $D/err.prof | tail -4 | cut -c1-300 | sed 's/^/     /'
f=$(ls $O/round6_prof_c2/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/round6_cli_c2_gz_kernel_stats.csv && head -16 $f
# sum against UNION of every kernel's launches (two files decode at once, and since the round's last day two passes of each)
t=$(ls $O/round6_prof_c2/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$t" ] && python3 - "$t" <<'PYEOF'
import csv, sys, collections
iv = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    iv[r["Kernel_Name"].split("(")[0].split("::")[-1].replace("void ", "")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
def union(v):
    v = sorted(v); tot = 0; cs, ce = v[0]
    for s, e in v[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
allv = [x for v in iv.values() for x in v]
print("     all kernels: %d launches, sum %.0f ms, union %.0f ms, first start to last end %.0f ms" % (len(allv), sum(e - s for s, e in allv) / 1e6, union(allv) / 1e6, (max(e for s, e in allv) - min(s for s, e in allv)) / 1e6))
for n, v in sorted(iv.items(), key=lambda kv: -sum(e - s for s, e in kv[1]))[:12]:
    print("     %-44s launches %6d  sum %8.1f ms  union %8.1f ms" % (n[:44], len(v), sum(e - s for s, e in v) / 1e6, union(v) / 1e6))
PYEOF
rm -rf $O/round6_prof_c2
fi
# the checkers: the oracle's program over all the reads; the real reference binary on the first 2M reads of each file
if has check2; then
head -n 4000000 $D/r1.fq > $D/s1.fq; head -n 4000000 $D/r2.fq > $D/s2.fq
if [ -x oracle/_ref/classify_O2 ]; then
  t0=$(now); timeout -k 5 420 oracle/_ref/classify_O2 --hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 --read $D/s1.fq --read $D/s2.fq -t 32 > $D/out.ref 2> /dev/null; t1=$(now)
  echo "-- REAL reference binary (-O2, -t 32) on 2M reads, same k-mer files: $(el $t0 $t1) s rows=$(wc -l < $D/out.ref) md5=$(md5sum < $D/out.ref | cut -c1-12)"
fi
run subsample $PY $ARGS --read $D/s1.fq --read $D/s2.fq
fi

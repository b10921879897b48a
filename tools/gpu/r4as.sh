# does the process BEHIND a classify run stall because that run left without freeing its device memory?  12M reads as two gzip -6
# files, back to back: leaving at once (default) against an orderly teardown (HAST_TEARDOWN=1), in blocks of four
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 6000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
now() { date +%s.%N; }
for mode in exit teardown exit teardown; do
  for rep in 1 2 3 4; do
    t0=$(now)
    if [ $mode = teardown ]; then HAST_TEARDOWN=1 hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats 2> $D/err > $D/out
    else hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats 2> $D/err > $D/out; fi
    t1=$(now)
    echo "$mode $rep wall=$(python3 -c "print(round($t1-$t0,3))") $(grep -h __stats_phases__ $D/err | grep -o "gpu_context_s=[0-9.]*\|load_kmers_s=[0-9.]*\|scrub_sizes_clone_s=[0-9.]*\|read_phase_s=[0-9.]*\|teardown_s=[0-9a-z.]*\|total_s=[0-9.]*" | tr '\n' ' ')"
  done
done
rm -rf $D

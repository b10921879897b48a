# `classify` on the 20M-read pair as two gzip -6 files with an environment switch set and not set, alternating on ONE box
# usage: gpurun -- 'AB_ENV="HAST_GZ_FREE_CUS=0" bash tools/gpu/gz_env_ab.sh > gpurun_out/gz_env_ab.txt 2>&1'     (QUAL=noisy: noisy quality lines)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_phases__ $D/err.$name | grep -o "gpu_context_s=[0-9.]*\|load_kmers_s=[0-9.]*\|scrub_sizes_clone_s=[0-9.]*\|read_phase_s=[0-9.]*\|total_s=[0-9.]*" | tr '\n' ' ') $(grep -h __stats_setup__ $D/err.$name | grep -o "waited_for_stream_setup_s=[0-9.]*")"; }
if [ "${QUAL:-const}" = noisy ]; then export GEN_FASTQ_QUAL=noisy; fi
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
for rep in $(seq 1 ${RUNS:-8}); do
  run tree_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
  run "[$AB_ENV]_$rep" env $AB_ENV hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
done
rm -rf $D

# which HIP call makes `classify` wait: the 20M-read pair as two gzip -6 files, RUNS times (default 12) under
# tools/hipstall.so (every runtime call longer than HIPSTALL_MS is printed with its time in the process), then plain FASTQ 6 times;
# the per-entry-point table of the first and of the slowest run.  usage: gpurun -- 'bash tools/gpu/gz_stalls.sh > gpurun_out/gz_stalls.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
g++ -O2 -shared -fPIC -o tools/hipstall.so tools/hipstall.cpp -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -ldl || exit 1
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); LD_PRELOAD=$PWD/tools/hipstall.so HIPSTALL_MS=${HIPSTALL_MS:-20} "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_phases__ $D/err.$name | grep -o "gpu_context_s=[0-9.]*\|load_kmers_s=[0-9.]*\|scrub_sizes_clone_s=[0-9.]*\|read_phase_s=[0-9.]*\|counters_back_s=[0-9.]*\|sort_print_s=[0-9.]*\|total_s=[0-9.]*" | tr '\n' ' ') $(grep -h "__hipstall_total__ the table" $D/err.$name | grep -o "[0-9.]* s after")"
  grep -h "^__hipstall__" $D/err.$name | sed 's/^/      /'; }
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
cat $D/r1.fq.gz $D/r2.fq.gz $D/r1.fq $D/r2.fq > /dev/null
for rep in $(seq 1 ${RUNS:-12}); do run gz_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
echo "== totals of run gz_1"; grep -h "^__hipstall_total__" $D/err.gz_1
for rep in 1 2 3 4 5 6; do run plain_$rep hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats; done
echo "== totals of run plain_1"; grep -h "^__hipstall_total__" $D/err.plain_1
echo "== without the library"; for rep in 1 2 3; do t0=$(now); hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 > $D/o 2> $D/e; t1=$(now); echo "gz_bare_$rep $(python3 -c "print(round($t1-$t0,3))") s"; done
rm -rf $D

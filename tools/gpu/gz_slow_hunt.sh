# round 6: the .gz run that takes 0.6-1.7 s instead of 0.2 s, one in five to twenty-five on some boxes: N runs of the 20M-read pair (noisy quality
# lines) with HAST_GZ_TRACE=1; of every run its read phase, of the slow ones the producer's steps that took more than 20 ms.
# usage: gpurun -- 'bash tools/gpu/gz_slow_hunt.sh > gpurun_out/gz_slow_hunt.txt 2>&1'       N=30
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/hast_slow.XXXXXX); trap 'rm -rf $D' EXIT
export GEN_FASTQ_QUAL=${QUAL:-noisy}
tools/gen_fastq $D 10000000 5000000 100000 21 150 32 0 || exit 1
tools/pgzip1 $D/r1.fq $D/r1.fq.gz 6 16 32; tools/pgzip1 $D/r2.fq $D/r2.fq.gz 6 16 32
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 -t 32 --stats --read $D/r1.fq.gz --read $D/r2.fq.gz"
# ALT="A=1 A=0": the runs alternate between these settings of one environment variable (e.g. HAST_GZ_PREALLOC=1 HAST_GZ_PREALLOC=0)
for rep in $(seq 1 ${N:-30}); do
 for alt in ${ALT:-HAST_UNUSED=1}; do
  case $alt in SLEEP_BEFORE=*) sleep ${alt#SLEEP_BEFORE=};; esac      # (ALT="SLEEP_BEFORE=0 SLEEP_BEFORE=3": does a pause behind the previous process's exit matter?)
  env $alt HAST_GZ_TRACE=1 hast_amd/classify $ARGS > $D/out 2> $D/err
  rp=$(grep -o "read_phase_s=[0-9.]*" $D/err | cut -d= -f2)
  echo "rep=$rep $alt read_phase_s=$rp $(grep -o "total_s=[0-9.]*\|scrub_sizes_clone_s=[0-9.]*\|load_kmers_s=[0-9.]*" $D/err | tr '\n' ' ') open_s=$(grep -o "open_s=[0-9.]*" $D/err | cut -d= -f2 | tr '\n' '/') md5=$(md5sum < $D/out | cut -c1-8)"
  if python3 -c "import sys; sys.exit(0 if float('$rp') > ${SLOW:-0.4} else 1)"; then
    echo "---- SLOW RUN $rep: steps of more than 20 ms"
    grep "^gz seg\|^gz open" $D/err | awk '{ if ($(NF-1) + 0 > 0.02) print "     " $0 }' | sed "s|$D/||g"
    grep -h "__stats_gz__\|__stats_read_phase__" $D/err | sed "s|$D/||g" | cut -c1-520 | sed 's/^/     /'
  fi
 done
done

# the stall of ~1 s that one `.gz` run in ten shows in its read phase: RUNS (default 40) runs of the 20M-read pair under tools/hipstall.so
# (HIP calls above 100 ms) with HAST_GZ_TRACE=1; a run whose read phase exceeds 0.5 s is printed with its stall lines and its trace
# usage: gpurun -- 'bash tools/gpu/gz_rare_stall.sh > gpurun_out/gz_rare_stall.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
g++ -O2 -shared -fPIC -o tools/hipstall.so tools/hipstall.cpp -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -ldl || exit 1
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
now() { date +%s.%N; }
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
slow=0
# the box's CPU quota (a cgroup: 16 cores' worth of a 256-thread host): how long the group was throttled during a run
thr() { cat /sys/fs/cgroup/cpu.stat 2>/dev/null | awk '/^nr_throttled/ {n=$2} /^throttled_usec/ {u=$2} END {print n+0, u+0}' ; }
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)"
for rep in $(seq 1 ${RUNS:-40}); do
  th0=($(thr)); t0=$(now); LD_PRELOAD=$PWD/tools/hipstall.so HIPSTALL_MS=100 HAST_GZ_TRACE=1 hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats > $D/out 2> $D/err; rc=$?; t1=$(now); th1=($(thr))
  rp=$(grep -h __stats_phases__ $D/err | grep -o "read_phase_s=[0-9.]*" | cut -d= -f2)
  echo "gz_$rep rc=$rc $(python3 -c "print(round($t1-$t0,3))") s read_phase_s=$rp $(grep -h __stats_read_phase__ $D/err | cut -d' ' -f2-5) throttled: $((th1[0]-th0[0])) periods, $(( (th1[1]-th0[1]) / 1000 )) ms"
  if python3 -c "import sys; sys.exit(0 if float('${rp:-0}') > 0.5 else 1)"; then
    slow=$((slow+1))
    echo "---- slow run gz_$rep: HIP calls above 100 ms, then the inflate's trace"
    grep -h "^__hipstall__" $D/err
    grep -hv "^__hipstall__" $D/err | cut -c1-260 | head -${TRACE_LINES:-150}
    echo "----"
    [ $slow -ge 3 ] && break
  fi
done
rm -rf $D

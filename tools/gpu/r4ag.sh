# --devices on one GPU with a 10-us idle wait of the striped relay
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq --read $D/r2.fq --weight0 1.04"
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_read_phase__ $D/err.$name | cut -d" " -f2- | cut -c1-110) $(grep -h __stats_phases__ $D/err.$name | grep -o "read_phase_s=[0-9.]*")"; }
cat $D/r1.fq $D/r2.fq > /dev/null
for rep in 1 2 3; do
run one_ctx hast_amd/classify $ARGS -t 32 --stats
run devices_0_0 hast_amd/classify $ARGS -t 32 --stats --devices 0,0
run devices_0_0_0_0 hast_amd/classify $ARGS -t 32 --stats --devices 0,0,0,0
done
HAST_TRACE_BLOCKS=1 hast_amd/classify $ARGS -t 32 --devices 0,0 2>&1 > /dev/null | grep trace | head -60 > gpurun_out/r4ag_trace.txt
rm -rf $D

# round 6: why did `rocprofv3 --kernel-trace` over the C2-size .gz run die?  20M reads, ring forced / not, teardown path / not, under the profiler / not
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/hast_dbg.XXXXXX); trap 'rm -rf $D' EXIT
tools/gen_fastq $D ${NPAIRS:-10000000} 5000000 100000 21 150 32 0 || exit 1
tools/pgzip1 $D/r1.fq $D/r1.fq.gz 1 16 32; tools/pgzip1 $D/r2.fq $D/r2.fq.gz 1 16 32
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 -t 32 --stats --read $D/r1.fq.gz --read $D/r2.fq.gz"
which gdb; ulimit -c 0
run() { local name=$1; shift; "$@" > $D/out.$name 2> $D/err.$name; echo "-- $name rc=$? md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_phases__ $D/err.$name | cut -c1-200)"; tail -3 $D/err.$name | cut -c1-300 | sed 's/^/     /'; }
run plain_exit hast_amd/classify $ARGS
HAST_TEARDOWN=1 run teardown hast_amd/classify $ARGS
HAST_TEARDOWN=1 run teardown_ring hast_amd/classify $ARGS --gz-ring-bytes 268435456
run prof rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof1 -- hast_amd/classify $ARGS
run prof_ring rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof2 -- hast_amd/classify $ARGS --gz-ring-bytes 268435456
ls $D/prof1/*/ $D/prof2/*/ 2>/dev/null | head

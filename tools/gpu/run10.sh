cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 3000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq --read $D/r2.fq --weight0 1.04"
cat $D/r1.fq $D/r2.fq > /dev/null
HAST_TRACE_BLOCKS=1 hast_amd/classify $ARGS -t 32 --stats > /dev/null 2> gpurun_out/trace.log
grep -E "^trace|__stats" gpurun_out/trace.log | head -150
rm -rf $D

# rocprofv3 kernel trace of the drop-in CLI itself (3M read pairs = 2 GB of plain FASTQ, 5M + 5M 21-mers): which kernels the program
# launches and what they cost (FASTQ framing, name cache, table build, filter build, classify, commit)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 3000000 5000000 100000 21 150 64 0 || exit 1
cat $D/r1.fq $D/r2.fq > /dev/null
rm -rf gpurun_out/prof_cli; mkdir -p gpurun_out/prof_cli
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cli -- hast_amd/classify --hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq --read $D/r2.fq --weight0 1.04 -t 32 --stats > $D/out.tsv 2> gpurun_out/prof_cli/stderr.txt
echo "rc=$? md5=$(md5sum < $D/out.tsv | cut -c1-12)"; grep __stats gpurun_out/prof_cli/stderr.txt
f=$(ls gpurun_out/prof_cli/*/*kernel_stats.csv | head -1); cp $f gpurun_out/cli_kernel_stats.csv; cut -d, -f1-5 $f | cut -c1-150 | head -20
rm -rf $D

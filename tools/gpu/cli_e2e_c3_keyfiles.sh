# the drop-in CLI with BASELINE-size key files: 200M + 200M 21-mers as text (2 x 4.4 GB), 2M reads
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
t0=$(date +%s.%N); tools/gen_fastq $D 1000000 200000000 100000 21 150 64 0 || exit 1; t1=$(date +%s.%N)
echo "generated in $(python3 -c "print(round($t1-$t0,1))") s"; ls -la $D
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq --read $D/r2.fq --weight0 1.04"
cat $D/hap0.mer $D/hap1.mer > /dev/null
for i in 1 2; do
  t0=$(date +%s.%N); hast_amd/classify $ARGS -t 32 --stats > $D/out.txt 2> $D/err.txt; rc=$?; t1=$(date +%s.%N)
  echo "classify rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.txt | cut -c1-12)"; grep -h "__stats" $D/err.txt; grep -v "__stats" $D/err.txt | tail -12
done
t0=$(date +%s.%N); hast_amd/classify $ARGS -t 32 --stats --save-table $D/table.bin > $D/out2.txt 2> $D/err2.txt; t1=$(date +%s.%N); echo "with --save-table: $(python3 -c "print(round($t1-$t0,3))") s"; ls -la $D/table.bin
t0=$(date +%s.%N); hast_amd/classify --load-table $D/table.bin --read $D/r1.fq --read $D/r2.fq --weight0 1.04 -t 32 --stats > $D/out3.txt 2> $D/err3.txt; rc=$?; t1=$(date +%s.%N); echo "--load-table rc=$rc: $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out3.txt | cut -c1-12)"; grep -h "__stats" $D/err3.txt; tail -5 $D/err3.txt
rm -rf $D

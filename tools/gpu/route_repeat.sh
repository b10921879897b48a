# the routing pass N times over several contexts and small blocks: do the routed files ever differ from the host router's?
# usage: gpurun -- 'N=40 bash tools/gpu/route_repeat.sh > gpurun_out/route_repeat.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
D=$(mktemp -d /tmp/hast_rr.XXXXXX); trap 'rm -rf $D' EXIT
cp tests/golden/rand_k21/* $D/
(cd $D && gunzip -k hap0.mer.gz hap1.mer.gz r1.fq.gz)
N=${N:-30}
sum() { (cd $1 && cat *.fastq *.barcodes filter_reads.log 2>/dev/null | md5sum | cut -c1-12); }
mkdir $D/ref && (cd $D/ref && $OLDPWD/hast_amd/classify --hap0 ../hap0.mer --hap1 ../hap1.mer --read ../r1.fq --read ../r2.fq.gz --weight0 1.04 --phase-reads --route host > out.tsv 2> err)
want=$(sum $D/ref); wout=$(md5sum < $D/ref/out.tsv | cut -c1-12)
for cfg in "--devices 0,0,0 --batch-reads 40" "--devices 0,0 --batch-reads 150" "--batch-reads 25" "--devices 0,0,0,0 --batch-reads 60 --deal files"; do
  bad=0
  for i in $(seq 1 $N); do
    rm -rf $D/w; mkdir $D/w
    (cd $D/w && $OLDPWD/hast_amd/classify --hap0 ../hap0.mer --hap1 ../hap1.mer --read ../r1.fq --read ../r2.fq.gz --weight0 1.04 --phase-reads $cfg > out.tsv 2> err)
    if [ "$(sum $D/w)" != "$want" ] || [ "$(md5sum < $D/w/out.tsv | cut -c1-12)" != "$wout" ]; then bad=$((bad+1)); fi
  done
  echo "[$cfg] $bad of $N runs differ from the host router's files ($want)"
done

# a golden case of the CLI run N times per environment: how often does stdout differ from the reference's?
# usage: gpurun -- 'ENVS="HAST_DEAL=files HAST_DEAL=files,HAST_PARK_GB=0" bash tools/gpu/flake_hunt.sh > gpurun_out/flake_hunt.txt 2>&1'   (an entry = comma-separated settings)
cd "${GRAFT_REPO_ROOT:-.}"
D=$(mktemp -d /tmp/hast_flake.XXXXXX)
cp tests/golden/rand_k21/* $D/
(cd $D && gunzip -k hap0.mer.gz hap1.mer.gz)
want=$(md5sum < $D/expected.pair_w104.tsv | cut -c1-12)
N=${N:-25}
for envs in ${ENVS:-"HAST_DEAL=files"}; do
  bad=0
  for i in $(seq 1 $N); do
    got=$(cd $D && env ${envs//,/ } $OLDPWD/hast_amd/classify --hap0 hap0.mer --hap1 hap1.mer --read r1.fq.gz --read r2.fq.gz --thread 8 --weight0 1.04 --devices 0,0,0 --batch-reads 50 --initial-barcodes 50 2> $D/err | md5sum | cut -c1-12)
    if [ "$got" != "$want" ]; then bad=$((bad+1)); cp $D/err $D/err.bad; fi
  done
  echo "[$envs] $bad of $N runs differ"
done
rm -rf $D

# device inflate: CUs left free by the decode passes (HAST_GZ_FREE_CUS), 12M reads as two gzip -6 files, read phase of three runs each
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 6000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
for fc in 32 16 64 0 32 16 64 0 32 16 64 0; do
  HAST_GZ_FREE_CUS=$fc hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats 2> $D/err > $D/out
  echo "free=$fc md5=$(md5sum < $D/out | cut -c1-8) $(grep -h __stats_phases__ $D/err | grep -o "read_phase_s=[0-9.]*") $(grep -h __stats_read_phase__ $D/err | grep -o "waiting_for_gpu_framing_s=[0-9.]*")"
done
rm -rf $D

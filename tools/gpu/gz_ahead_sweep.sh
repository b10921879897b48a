# round 6: passes of a .gz file side by side (HAST_GZ_AHEAD = 0, 1, 2) x chunks per pass: read phase and HBM at the peak over the 20M-read pairs
# (two single-member gzip -6 files, constant and noisy quality lines), variants in turn, REPS times.
# usage: gpurun -- 'bash tools/gpu/gz_ahead_sweep.sh > gpurun_out/gz_ahead_sweep.txt 2>&1'      VARIANTS="ahead:pass[:free CUs] ..."  (pass 0 = the default, 6144)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/hast_asw.XXXXXX); trap 'rm -rf $D' EXIT
VARIANTS=${VARIANTS:-1:0 2:0 1:4096 2:4096 1:3072 2:3072 0:0}
for q in ${QUALS:-const noisy}; do
  [ $q = noisy ] && export GEN_FASTQ_QUAL=noisy
  tools/gen_fastq $D 10000000 5000000 100000 21 150 32 0 || exit 1
  tools/pgzip1 $D/r1.fq $D/r1.fq.gz 6 16 32; tools/pgzip1 $D/r2.fq $D/r2.fq.gz 6 16 32
  echo "== quality lines: $q; $(stat -c %s $D/r1.fq.gz) bytes per .gz"
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 -t 32 --stats --read $D/r1.fq.gz --read $D/r2.fq.gz"
  hast_amd/classify $ARGS > /dev/null 2>&1       # (the box's first run)
  for rep in $(seq 1 ${REPS:-4}); do
    for v in $VARIANTS; do
      IFS=: read a p cus <<< "$v"
      envs="HAST_GZ_AHEAD=$a"; [ "$p" != 0 ] && envs="$envs HAST_GZ_PASS_CHUNKS=$p HAST_GZ_ROOM=20"; [ -n "$cus" ] && envs="$envs HAST_GZ_FREE_CUS=$cus"
      env $envs hast_amd/classify $ARGS > $D/out 2> $D/err
      echo "ahead=$a pass=$p${cus:+ free_cus=$cus} rep=$rep rc=$? md5=$(md5sum < $D/out | cut -c1-8) $(grep -o "read_phase_s=[0-9.]*" $D/err) $(grep -h __stats_gz__ $D/err | grep -o "followup_jobs=[0-9]*" | tr '\n' ' ') $(grep -o "in_use_peak_bytes=[0-9]*" $D/err)"
    done
  done
done

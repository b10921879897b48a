# round 6: passes of one .gz stream side by side (HAST_GZ_AHEAD) are new concurrency -- two or three decode kernels of a file in flight, arenas and
# job arrays taking turns, the ring's release moving behind them.  N runs per configuration of `classify` over a small .gz pair cut into HUNDREDS of
# passes (4-KB chunks, 8 a pass), with and without the ring, one and two contexts (one decode unit, and a unit per context): does stdout ever differ
# from the plain files' output?
# usage: gpurun -- 'N=25 bash tools/gpu/gz_ahead_repeat.sh > gpurun_out/gz_ahead_repeat.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/hast_gar.XXXXXX); trap 'rm -rf $D' EXIT
tools/gen_fastq $D 100000 200000 5000 21 150 32 0 || exit 1
tools/pgzip1 $D/r1.fq $D/r1.fq.gz 6 4 4; tools/pgzip1 $D/r2.fq $D/r2.fq.gz 9 4 4
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 -t 8 --stats"
hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq > $D/want 2> /dev/null
want=$(md5sum < $D/want | cut -c1-12)
echo "two .gz files of $(stat -c %s $D/r1.fq.gz) / $(stat -c %s $D/r2.fq.gz) bytes (levels 6 / 9), the plain files' stdout md5 $want"
N=${N:-25}
export HAST_GZ_CHUNK_BYTES=4096 HAST_GZ_PASS_CHUNKS=8
for ahead in 1 2 0; do
  for ring in "" "HAST_GZ_RING_BYTES=262144 HAST_GZ_PIECE_BYTES=65536"; do
    for dev in "" "--devices 0,0" "--devices 0,0 SPLIT"; do
      split=""; case "$dev" in *SPLIT) split="HAST_GZ_SPLIT=contexts"; dev="--devices 0,0";; esac
      bad=0; passes=0; laps=0
      for i in $(seq 1 $N); do
        env HAST_GZ_AHEAD=$ahead $ring $split hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz $dev > $D/out 2> $D/err || bad=$((bad+1000))
        [ "$(md5sum < $D/out | cut -c1-12)" = "$want" ] || bad=$((bad+1))
      done
      laps=$(grep -o "ring_laps=[0-9]*" $D/err | head -1); ch=$(grep -o "chunks=[0-9]*" $D/err | head -1)
      echo "[HAST_GZ_AHEAD=$ahead ${ring:+ring 256 KB }${split:+a unit per context }$dev] $bad of $N runs differ ($ch = $(( ${ch#chunks=} / 8 )) passes${laps:+, $laps})"
    done
  done
done

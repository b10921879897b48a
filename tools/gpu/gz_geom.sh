# round 6: what a decode pass costs is its SLOWEST wave -- the counters say the decode kernel's waves issue for 39 % of their
# lifetime and the kernel takes ~4 x its instruction-issue bound: the loss is occupancy, not the wave's own speed.  Finer / more work units per
# pass: HAST_GZ_CHUNK_BYTES x HAST_GZ_PASS_CHUNKS (arena bytes = chunks x chunk bytes x 24 stay the same along a diagonal), 20M reads as two
# single-member gzip -6 files, constant and noisy quality lines; read phase + decode kernel time (rocprofv3 kernel stats) per variant.
# usage: gpurun -- 'bash tools/gpu/gz_geom.sh > gpurun_out/round6_gz_geom.txt 2>&1'      GEOMS="32768:4096 16384:8192 ..."
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
D=$(mktemp -d /dev/shm/hast_geom.XXXXXX); trap 'rm -rf $D' EXIT
GEOMS=${GEOMS:-32768:4096:12 16384:8192:12 16384:8192:24 16384:4096:24 32768:8192:12 8192:16384:48 65536:2048:12}
for q in ${QUALS:-const noisy}; do
  [ $q = noisy ] && export GEN_FASTQ_QUAL=noisy
  tools/gen_fastq $D 10000000 5000000 100000 21 150 32 0 || exit 1
  tools/pgzip1 $D/r1.fq $D/r1.fq.gz 6 16 32; tools/pgzip1 $D/r2.fq $D/r2.fq.gz 6 16 32
  echo "== quality lines: $q; $(stat -c %s $D/r1.fq.gz) bytes per .gz"
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 -t 32 --stats --read $D/r1.fq.gz --read $D/r2.fq.gz"
  for rep in 1 2; do
    for g in $GEOMS; do
      IFS=: read c p r fr <<< "$g"
      envs="HAST_GZ_SLOT_FRACTION=${fr:-0.7} HAST_GZ_ROOM=$r HAST_GZ_CHUNK_BYTES=$c HAST_GZ_PASS_CHUNKS=$p"; [ "$g" = default ] && envs="HAST_UNUSED=1"
      env $envs hast_amd/classify $ARGS > $D/out 2> $D/err
      echo "chunk=$c pass=$p room=$r slots=${fr:-0.7} rep=$rep rc=$? md5=$(md5sum < $D/out | cut -c1-8) $(grep -o "read_phase_s=[0-9.]*\|total_s=[0-9.]*" $D/err | tr '\n' ' ') $(grep -h __stats_gz__ $D/err | head -1 | grep -o "decode_s=[0-9.]*\|chain_walk_s=[0-9.]*\|followup_jobs=[0-9]*\|chunks=[0-9]*\|accepted=[0-9]*" | tr '\n' ' ') hbm=$(grep -o "in_use_peak_bytes=[0-9]*" $D/err)"
    done
  done
  for g in ${PROF_GEOMS:-32768:4096:12 16384:8192:24}; do
    IFS=: read c p r fr <<< "$g"
    if [ "$g" = default ]; then unset HAST_GZ_SLOT_FRACTION HAST_GZ_ROOM HAST_GZ_CHUNK_BYTES HAST_GZ_PASS_CHUNKS; else export HAST_GZ_SLOT_FRACTION=${fr:-0.7} HAST_GZ_ROOM=$r HAST_GZ_CHUNK_BYTES=$c HAST_GZ_PASS_CHUNKS=$p; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -- hast_amd/classify $ARGS > $D/out 2> $D/err
    f=$(ls $D/prof/*/*kernel_stats.csv | head -1)
    echo "-- kernel stats chunk=$c pass=$p room=$r slots=${fr:-0.7} ($q): $(grep -o "read_phase_s=[0-9.]*" $D/err) $(grep -h __stats_gz__ $D/err | head -1 | grep -o "followup_jobs=[0-9]*") $(grep -o "in_use_peak_bytes=[0-9]*" $D/err)"
    python3 - "$f" <<'PYEOF'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "gz::" in n:
        print("     %-16s calls %5s total %8.1f ms avg %8.3f ms max %8.3f ms" % (n.split("(")[0].split("::")[-1], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, int(r["MaxNs"]) / 1e6))
PYEOF
    cp $f $O/round6_gz_geom_${q}_${c}_${p}_${r}_${fr:-0.7}_kernel_stats.csv; rm -rf $D/prof
    unset HAST_GZ_SLOT_FRACTION HAST_GZ_ROOM HAST_GZ_CHUNK_BYTES HAST_GZ_PASS_CHUNKS
  done
done

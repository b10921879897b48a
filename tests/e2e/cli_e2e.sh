#!/bin/bash
# TEST INFRASTRUCTURE (it executes oracle/_ref, the real reference binaries): end-to-end (M2) measurement of the drop-in CLI against the REAL reference binary on the same files,
# run on the GPU box through gpurun:   bash tests/e2e/cli_e2e.sh [n_pairs] [keys_per_hap] [barcodes] [tag]
# Generates the synthetic C1-style inputs (tools/gen_fastq), runs oracle/_ref/classify (as shipped, -g) and
# classify_O2 on the host cores, then hast_amd/classify at several -t, compares stdout md5, and writes
# gpurun_out/cli_e2e_<tag>.json.  File I/O + parse + H2D are inside every timing (this is not the bench metric).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
NP=${1:-1000000}; NK=${2:-1000000}; NB=${3:-10000}; TAG=${4:-c1}
# each read pair is ~680 bytes of FASTQ, each key 22 bytes of text: refuse sizes that would fill the box's /tmp
if [ "$NP" -gt 30000000 ] || [ "$NK" -gt 500000000 ]; then echo "cli_e2e.sh: $NP read pairs / $NK keys is more than this script may write" >&2; exit 2; fi
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
OUT=gpurun_out/cli_e2e_$TAG.json
mkdir -p gpurun_out
tools/gen_fastq $D $NP $NK $NB 21 150 32 ${CLUSTERED:-0} || exit 1
BYTES=$(stat -c %s $D/r1.fq); READS=$((NP*2)); BP=$((READS*150))
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq --read $D/r2.fq --weight0 1.04"
now() { date +%s.%N; }
run() { # name cmd...
  local name=$1; shift
  local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  local md5=$(md5sum < $D/out.$name | cut -d' ' -f1)
  python3 -c "import json,sys; print(json.dumps({'name':'$name','rc':$rc,'seconds':round($t1-$t0,3),'Mbp_per_s':round($BP/($t1-$t0)/1e6,1),'md5':'$md5','rows':sum(1 for _ in open('$D/out.$name'))}))"
}
{
echo "{\"pairs\": $NP, \"reads\": $READS, \"bp\": $BP, \"keys_per_hap\": $NK, \"barcodes\": $NB, \"fastq_bytes_each\": $BYTES, \"host_threads\": $(nproc), \"runs\": ["
if [ -x oracle/_ref/classify_O2 ]; then
  [ "$NP" -le 2000000 ] && { run ref_shipped_t8 timeout 900 oracle/_ref/classify $ARGS -t 8; echo ","; }
  run ref_O2_t8 timeout 900 oracle/_ref/classify_O2 $ARGS -t 8; echo ","
  run ref_O2_t32 timeout 900 oracle/_ref/classify_O2 $ARGS -t 32; echo ","
fi
for T in 1 8 32 64; do run hast_t$T hast_amd/classify $ARGS -t $T --stats; echo ","; done
run hast_t64_again hast_amd/classify $ARGS -t 64 --stats
if [ "${GZ:-0}" = 1 ]; then
  (gzip -1 -k $D/r1.fq & gzip -1 -k $D/r2.fq & wait)
  GZARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --read $D/r1.fq.gz --read $D/r2.fq.gz --weight0 1.04"
  echo ","; run hast_gz_t32 hast_amd/classify $GZARGS -t 32 --stats
  [ -x oracle/_ref/classify_O2 ] && { echo ","; run ref_O2_gz_t32 timeout 1800 oracle/_ref/classify_O2 $GZARGS -t 32; }
fi
if [ "${QUARTER:-0}" = 1 ]; then
  # the step after classify with OUR partitioner (the reference's awk program does not travel to this box;
  # its timing is taken in the build container)
  ( cd $D && awk '{if($2 == 0) print $1;}' out.hast_t64 > p.bc && awk '{if($2 == 1) print $1;}' out.hast_t64 > m.bc && awk '{if($2 == "-1") print $1;}' out.hast_t64 > h.bc )
  for T in 1 8 32; do echo ","; ( cd $D && rm -f r1.fq.*.fastq; run quarter_t$T $OLDPWD/hast_amd/quartering_fastq -t $T --prefix r1.fq p.bc m.bc h.bc r1.fq ); done
fi
echo "]}"
} > $OUT
grep -h "__stats__" $D/err.hast_t* | tail -5
cat $OUT
rm -rf $D

#!/bin/bash
# TEST INFRASTRUCTURE (executes oracle/_ref/classify_s03, the real stage-03 reference binary): end-to-end timing of
# the per-read classifier on long reads, same files for both programs, stdout md5 compared.
#   bash tests/e2e/cli_e2e_s03.sh [n_reads] [keys_per_hap] [read_len] [tag]      (run through gpurun)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
NR=${1:-5000}; NK=${2:-2000000}; L=${3:-20000}; TAG=${4:-s03}
# the FASTQ written below is ~2 x NR x L bytes: refuse sizes that would fill the box's /tmp (argument order: reads, KEYS, length)
if [ $((NR * L)) -gt 8000000000 ] || [ "$NK" -gt 500000000 ]; then echo "cli_e2e_s03.sh: $NR reads x $L bases / $NK keys is more than this script may write" >&2; exit 2; fi
D=$(mktemp -d /tmp/hast_e2e_s03.XXXXXX)
OUT=gpurun_out/cli_e2e_$TAG.json
mkdir -p gpurun_out
tools/gen_fastq $D $NR $NK 1 31 $L 32 || exit 1
BP=$((NR*L))
ARGS="--hap $D/hap0.mer --hap $D/hap1.mer --read $D/r1.fq --format fastq"
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  python3 -c "import json; print(json.dumps({'name':'$name','rc':$rc,'seconds':round($t1-$t0,3),'Mbp_per_s':round($BP/($t1-$t0)/1e6,1),'md5':'$(md5sum < $D/out.$name | cut -d' ' -f1)','rows':sum(1 for _ in open('$D/out.$name'))}))"; }
{
echo "{\"reads\": $NR, \"read_len\": $L, \"bp\": $BP, \"keys_per_hap\": $NK, \"k\": 31, \"fastq_bytes\": $(stat -c %s $D/r1.fq), \"host_threads\": $(nproc), \"runs\": ["
[ -x oracle/_ref/classify_s03 ] && { run ref_s03_t32 timeout 1500 oracle/_ref/classify_s03 $ARGS --thread 32; echo ","; }
for T in 1 8 32; do run hast_read_t$T hast_amd/classify_read $ARGS --thread $T; [ $T != 32 ] && echo ","; done
echo "]}"
} > $OUT
cat $OUT
rm -rf $D

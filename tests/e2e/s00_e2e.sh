#!/bin/bash
# TEST INFRASTRUCTURE (executes oracle/oracle_unshared, the pinned CPU restatement of stage 00): end-to-end run of the
# stage-00 drop-in on FASTQ files, same files for both programs, products compared byte for byte.
#   bash tests/e2e/s00_e2e.sh [genome_len] [coverage] [files_per_parent] [tag]      (run through gpurun)
# The real reference (build_unshared_kmers.sh + the jellyfish binary vendored next to it) only exists in the build
# container; its wall time on the same generator settings is recorded by hand in DESIGN.md.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
G=${1:-5000000}; COV=${2:-30}; NF=${3:-2}; TAG=${4:-s00}
# the four FASTQ sets hold ~4 x G x COV bytes: refuse sizes that would fill the box's /tmp
if [ $((G * COV)) -gt 4000000000 ]; then echo "s00_e2e.sh: a $G-base genome at ${COV}x is more than this script may write" >&2; exit 2; fi
D=$(mktemp -d /tmp/hast_e2e_s00.XXXXXX)
OUT=$PWD/gpurun_out/cli_e2e_$TAG.json
mkdir -p gpurun_out
[ -x tools/gen_trio ] || g++ -O2 -std=c++17 -pthread -o tools/gen_trio tools/gen_trio.cpp || exit 1
tools/gen_trio $D $G $COV 150 $NF 32 || exit 1
BP=$(python3 -c "print(2*int($G*$COV/150)*150)")
ARGS=""
for i in $(seq 0 $((NF-1))); do ARGS="$ARGS --paternal $D/paternal_$i.fq --maternal $D/maternal_$i.fq"; done
now() { date +%s.%N; }
run() { local name=$1; shift; mkdir -p $D/$name; local t0=$(now); (cd $D/$name && "$@" > out.txt 2> err.txt); local rc=$?; local t1=$(now)
  python3 -c "import json; print(json.dumps({'name':'$name','rc':$rc,'seconds':round($t1-$t0,3),'Mbp_per_s':round($BP/($t1-$t0)/1e6,1),'md5_paternal':'$(sort $D/$name/paternal.unique.filter.mer 2>/dev/null | md5sum | cut -d' ' -f1)','md5_maternal':'$(sort $D/$name/maternal.unique.filter.mer 2>/dev/null | md5sum | cut -d' ' -f1)','md5_histo':'$(cat $D/$name/maternal.histo $D/$name/paternal.histo $D/$name/*.bounds.txt 2>/dev/null | md5sum | cut -d' ' -f1)','rows':[sum(1 for _ in open('$D/$name/paternal.unique.filter.mer')),sum(1 for _ in open('$D/$name/maternal.unique.filter.mer'))], 'stats':[l.strip() for l in open('$D/$name/err.txt') if l.startswith('[stats]')]}))"; }
{
echo "{\"genome\": $G, \"coverage\": $COV, \"bp\": $BP, \"files_per_parent\": $NF, \"fastq_bytes_each\": $(stat -c %s $D/paternal_0.fq), \"host_threads\": $(nproc), \"runs\": ["
run oracle_cpu_1thread timeout 3000 $PWD/oracle/oracle_unshared $ARGS --thread 1 --auto_bounds; echo ","
for T in 1 8; do run hast_t$T $PWD/hast_amd/unshared_kmers $ARGS --thread $T --auto_bounds --stats; echo ","; done
run hast_t8_again $PWD/hast_amd/unshared_kmers $ARGS --thread 8 --auto_bounds --stats
if [ "${GZ:-1}" = 1 ]; then
  for f in $D/*.fq; do gzip -1 -k $f & done; wait
  echo ","; run hast_gz_t8 $PWD/hast_amd/unshared_kmers ${ARGS//.fq/.fq.gz} --thread 8 --auto_bounds --stats
fi
echo "]}"
} > $OUT
cat $OUT
rm -rf $D

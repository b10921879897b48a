# BASELINE config 3's TABLE through the boundary (VERDICT r5 "missing" #2: no run had held the 50 GB of table + filter next to two open .gz streams):
# 200M + 200M 21-mers as two text files of 4.4 GB, 10M barcodes, 20M reads as plain FASTQ and as two single-member .gz files; `classify`
# plain / .gz / .gz over two contexts (the table cloned) / --phase-reads, --stats (HBM in use at the peak), stdout md5 of every run against
# the oracle's program on the same files (its two hash sets of 200M keys each: ~25 GB of host memory, minutes to load).
# usage: gpurun --timeout 1200 -- 'bash tools/gpu/cli_c3_table.sh > gpurun_out/round6_cli_c3_table.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/hast_c3.XXXXXX)
( while sleep 45; do echo "[still running $(date +%T)]"; done ) &
HB=$!
trap 'kill $HB 2>/dev/null; rm -rf $D' EXIT
now() { date +%s.%N; }
el() { python3 -c "print(round($2-$1,2))"; }
t0=$(now); GEN_FASTQ_MAX_GB=40 tools/gen_fastq $D ${NPAIRS:-10000000} ${KEYS:-200000000} ${BARCODES:-10000000} 21 150 32 0 || exit 1; t1=$(now)
echo "== generated in $(el $t0 $t1) s: $(stat -c %s $D/hap0.mer) bytes per k-mer file, $(stat -c %s $D/r1.fq) per FASTQ file"
tools/pgzip1 $D/r1.fq $D/r1.fq.gz 6 16 32; tools/pgzip1 $D/r2.fq $D/r2.fq.gz 6 16 32
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 -t 32 --stats"
run() { local name=$1; shift; local w=$D/w.$name; mkdir -p $w; local t0=$(now); (cd $w && "$@" > $D/out.$name 2> $D/err.$name); local rc=$?; local t1=$(now)
  echo "-- $name rc=$rc whole process $(el $t0 $t1) s rows=$(wc -l < $D/out.$name) stdout md5=$(md5sum < $D/out.$name | cut -c1-12)"
  grep -h "__stats_phases__\|__stats_hbm__\|__stats_filter__\|__stats__ \|__stats_phase_reads__\|WARN\|ERROR" $D/err.$name | cut -c1-400 | sed 's/^/     /'; rm -rf $w; }
PY=$PWD/hast_amd/classify
run plain $PY $ARGS --read $D/r1.fq --read $D/r2.fq
run gz $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz
run gz_devices_0_0 $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz --devices 0,0
run gz_route $PY $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz --phase-reads
run plain_again $PY $ARGS --read $D/r1.fq --read $D/r2.fq
t0=$(now); timeout -k 5 ${ORACLE_TIMEOUT:-800} oracle/oracle_classify --hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 --read $D/r1.fq --read $D/r2.fq -t 32 > $D/out.oracle 2> /dev/null; rc=$?; t1=$(now)
echo "-- oracle_classify -t 32 (200M + 200M keys, 20M reads): rc=$rc $(el $t0 $t1) s rows=$(wc -l < $D/out.oracle) md5=$(md5sum < $D/out.oracle | cut -c1-12)"

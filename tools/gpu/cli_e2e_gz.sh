# ordinary .gz input through the drop-in CLI: the parallel inflate (par_inflate.h) against the serial decoder and zlib,
# gzip levels 1 and 6; 20M reads (2 x 3.4 GB of FASTQ); stdout md5 must be the same in every run.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
NP=10000000
tools/gen_fastq $D $NP 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_read_phase__ $D/err.$name | cut -d" " -f2-)"; }
cat $D/r1.fq $D/r2.fq > /dev/null
run plain_t32 hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats
g++ -O2 -std=c++17 -o /tmp/ti_fast tests/native/test_inflate.cpp -lz -pthread
for lvl in 6 1; do
  (gzip -$lvl -c $D/r1.fq > $D/r1.l$lvl.fq.gz & gzip -$lvl -c $D/r2.fq > $D/r2.l$lvl.fq.gz & wait)
  ls -la $D/r1.l$lvl.fq.gz
  GZ="--read $D/r1.l$lvl.fq.gz --read $D/r2.l$lvl.fq.gz"
  for t in 1 4 8 16 32; do echo "inflate only, level $lvl, $t threads: $(/tmp/ti_fast -q -P -t $t -p 16777216 $D/r1.l$lvl.fq.gz 2>&1 | tail -1)"; done
  echo "inflate only, level $lvl, serial decoder: $(/tmp/ti_fast -q -p 16777216 $D/r1.l$lvl.fq.gz 2>&1 | tail -1)"
  echo "inflate only, level $lvl, zlib: $(/tmp/ti_fast -q -z -p 16777216 $D/r1.l$lvl.fq.gz 2>&1 | tail -1)"
  HAST_GZ_THREADS=1 run gz${lvl}_serial hast_amd/classify $ARGS $GZ -t 32 --stats
  for t in 4 8 16 32; do HAST_GZ_THREADS=$t run gz${lvl}_par$t hast_amd/classify $ARGS $GZ -t 32 --stats; done
  run gz${lvl}_default hast_amd/classify $ARGS $GZ -t 32 --stats
  HAST_GZ_THREADS=16 run gz${lvl}_par16_hostparse hast_amd/classify $ARGS $GZ -t 32 --stats --host-parse
done
HAST_GZ_TRACE=1 HAST_GZ_THREADS=16 hast_amd/classify $ARGS --read $D/r1.l6.fq.gz -t 32 > /dev/null 2> $D/trace.txt; grep pargz $D/trace.txt | head -12
rm -rf $D

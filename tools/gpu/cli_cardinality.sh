# The boundary at BASELINE's barcode cardinalities (classify.cpp:50-64,93-102: one map entry and one output row per barcode): `classify` on
# 20M reads (2 x 3.4 GB of FASTQ, plain and as gzip -6) over 1M (config 2) and 10M (config 3) barcodes, --stats phases, stdout md5 against
# the oracle's line-by-line program on the same files and against the REAL reference binary on a 2M-read subsample.
# usage: gpurun -- 'bash tools/gpu/cli_cardinality.sh > gpurun_out/cli_cardinality.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/hast_e2e.XXXXXX)
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s rows=$(wc -l < $D/out.$name) md5=$(md5sum < $D/out.$name | cut -c1-12)"
  grep -h "__stats_phases__\|__stats_read_phase__\|__stats_phase_reads__" $D/err.$name | sed 's/^/    /'; }
for nbc in ${BARCODES:-1000000 10000000}; do
  echo "== $nbc barcodes, 10M read pairs of 150 bp, 5M + 5M 21-mers"
  tools/gen_fastq $D 10000000 5000000 $nbc 21 150 64 0 || exit 1
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
  cat $D/r1.fq $D/r2.fq > /dev/null
  for rep in 1 2 3; do run plain_$rep hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats; done
  tools/pgzip1 $D/r1.fq $D/r1.fq.gz 6 16 32; tools/pgzip1 $D/r2.fq $D/r2.fq.gz 6 16 32      # (ONE gzip member each, as `gzip -6` writes; 16 threads)
  cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
  for rep in 1 2; do run gz_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
  if [ -n "$CHUNK_AB" ]; then      # the device inflate's chunk size (compressed bytes a wave decodes), alternating with the default
    for rep in 1 2; do
      HAST_GZ_CHUNK_BYTES=65536 run gz_chunk64k_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
      run gz_chunk32k_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
      HAST_GZ_CHUNK_BYTES=49152 run gz_chunk48k_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
    done
  fi
  run devices_0_0 hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats --devices 0,0
  # round 6: who numbers the barcodes -- the GPU's dictionary (default), one dictionary per context merged by text (what several GPUs do), the host's (round 5)
  HAST_NAME_DICT=context run devices_0_0_0_dict_per_context hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats --devices 0,0,0
  HAST_NAME_DICT=0 run host_dictionary_1 hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats
  HAST_NAME_DICT=0 run host_dictionary_2 hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 --stats
  # round 6: the wrapper's steps 10-11, routed on the GPU and by the host threads: the same files
  mkdir -p $D/wd $D/wh
  (cd $D/wd && run route_device $OLDPWD/hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq.gz -t 32 --stats --phase-reads)
  (cd $D/wh && run route_host $OLDPWD/hast_amd/classify $ARGS --read $D/r1.fq --read $D/r2.fq.gz -t 32 --stats --phase-reads --route host)
  for f in $(cd $D/wh && ls *.fastq *.barcodes filter_reads.log); do cmp -s $D/wd/$f $D/wh/$f && echo "    routed $f: identical ($(stat -c %s $D/wd/$f) bytes)" || echo "    routed $f: DIFFERENT"; done
  echo "    files: device $(ls $D/wd | wc -l), host $(ls $D/wh | wc -l)"; rm -rf $D/wd $D/wh
  t0=$(now); oracle/oracle_classify $ARGS --read $D/r1.fq --read $D/r2.fq -t 32 > $D/out.oracle 2> /dev/null; t1=$(now)
  echo "oracle_classify -t 32: $(python3 -c "print(round($t1-$t0,1))") s rows=$(wc -l < $D/out.oracle) md5=$(md5sum < $D/out.oracle | cut -c1-12)"
  # the real reference binary on the first 2M reads of each file (8M lines), and the product on the same subsample
  head -n 4000000 $D/r1.fq > $D/s1.fq; head -n 4000000 $D/r2.fq > $D/s2.fq
  if [ -x oracle/_ref/classify_O2 ]; then
    t0=$(now); oracle/_ref/classify_O2 $ARGS --read $D/s1.fq --read $D/s2.fq -t 32 > $D/out.ref 2> /dev/null; t1=$(now)
    echo "reference binary (-O2, -t 32) on 2M reads: $(python3 -c "print(round($t1-$t0,1))") s rows=$(wc -l < $D/out.ref) md5=$(md5sum < $D/out.ref | cut -c1-12)"
  fi
  run subsample hast_amd/classify $ARGS --read $D/s1.fq --read $D/s2.fq -t 32 --stats
  rm -f $D/*
done
rm -rf $D

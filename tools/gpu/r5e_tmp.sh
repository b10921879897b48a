cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
bash tools/gpu/s00_pmc.sh > $O/r5k_s00_pmc.txt 2>&1; grep "k_kc_apply\|k_kc_count<6, true>\|k_kc_part<1>" $O/r5k_s00_pmc.txt | cut -c1-1200
timeout -k 10 600 python -m pytest tests/test_fq_gpu.py tests/test_cli_gpu.py -x -q -k "not heavy and not golden" > $O/r5k_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/r5k_pytest.log)"
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_phases__ $D/err.$name | grep -o "gpu_context_s=[0-9.]*\|load_kmers_s=[0-9.]*\|scrub_sizes_clone_s=[0-9.]*\|read_phase_s=[0-9.]*\|total_s=[0-9.]*" | tr '\n' ' ') $(grep -h __stats_setup__ $D/err.$name | cut -d' ' -f2) $(grep -h __stats_read_phase__ $D/err.$name | cut -d' ' -f2-4)"; }
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
cat $D/r1.fq.gz $D/r2.fq.gz > /dev/null
for rep in 1 2 3 4 5; do run gz_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
for fc in 16 64 96 32; do for rep in 1 2; do HAST_GZ_FREE_CUS=$fc run gz_free${fc}_$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done; done
for rep in 1 2 3; do t0=$(now); hast_amd/classify --help > /dev/null 2>&1; t1=$(now); echo "classify --help (process start + exit, no GPU call): $(python3 -c "print(round($t1-$t0,3))") s"; done
rm -rf $D

#!/bin/bash
# Collects a tag's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#     bash profiles/collect.sh <tag> [bench flags, e.g. --clustered]
#   1. kernel-trace + stats of the bench command (the DEFAULT command when no flags are given)
#   2. PMC passes, each in its own run with --kernel-trace only (the pool refuses --pmc with other trace domains):
#        rdsize   TCC_EA0_RDREQ{,_32B,_64B,_128B}_sum   read requests L2 -> fabric by size: bytes = 32 a + 64 b + 128 c
#        rddram   TCC_EA0_RDREQ_DRAM_32B_sum ...         the same bytes in 32-B units as the DRAM side counts them
#        fetch    FETCH_SIZE                             (tallies a gfx950 128-B request at 64 B: MI355X_MICROARCH.md, HBM)
#        write    WRITE_SIZE + write/atomic requests
#        l2       TCC_HIT/MISS/REQ/READ
#        sq       instruction counts, sq2: cycle breakdown
#   3. the rdsize/rddram/fetch passes over tools/hbm_randread with KNOWN byte counts (random 64-B lines and random 128-B
#      blocks, same 16-B-per-lane access shape) = the calibration of those counters on this access pattern.
# profiles/summarize.py <tag> turns gpurun_out/prof_<tag>/ into the tracked summaries under profiles/.
set -u
TAG=${1:-r02}
shift || true
FLAGS="$*"
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
echo "$FLAGS" > $OUT/flags.txt
python3 -c "import bench; print(bench.kernel_source_id())" > $OUT/kernel_source_id.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $FLAGS > $OUT/stats_bench.json 2> $OUT/stats_bench.err
declare -A PASS
PASS[rdsize]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
PASS[rddram]="TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum TCC_READ_SECTORS_sum"
PASS[fetch]="FETCH_SIZE"
PASS[write]="WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum"
PASS[l2]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
PASS[sq]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
PASS[sq2]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT"
for name in rdsize rddram fetch write l2 sq sq2; do
  rocprofv3 --pmc ${PASS[$name]} --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --batches-per-step 1 --no-secondary --cpu-seconds 0 $FLAGS > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
if [ -z "$FLAGS" ]; then
  for name in rdsize rddram fetch; do
    for line in 64 128; do
      rocprofv3 --pmc ${PASS[$name]} --kernel-trace --output-format csv -d $OUT/cal${line}_$name -- tools/hbm_randread 6.4 $line 4 256 2048 > $OUT/cal${line}_$name.json 2> $OUT/cal${line}_$name.err
    done
  done
fi
find $OUT -name "*counter_collection.csv" | wc -l

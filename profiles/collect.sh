#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. kernel-trace + stats of the default bench command (shortened step count)
#   2. PMC passes (FETCH_SIZE / WRITE_SIZE / TCC hit+miss+EA read requests), each in its own run with
#      --kernel-trace only, as the pool requires
#   3. the same PMC pass over tools/hbm_randread (known byte count, same access shape) to calibrate
#      FETCH_SIZE for random 64-B line reads (MI355X_MICROARCH.md, HBM section)
# Outputs land in gpurun_out/prof_*; summaries are copied into profiles/ by hand afterwards.
set -u
TAG=${1:-r01}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats_bench.json 2> $OUT/stats_bench.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/cal_$name -- tools/hbm_randread 6.4 64 4 256 2048 > $OUT/cal_$name.json 2> $OUT/cal_$name.err
done
find $OUT -name "*.csv" | head -50

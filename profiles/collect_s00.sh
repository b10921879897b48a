#!/bin/bash
# rocprofv3 evidence for stage 00's counting path on the GPU box (run through gpurun from the repo root):
#     bash profiles/collect_s00.sh <tag> [atomic]
#   1. kernel trace + stats of `bench.py --workload s00` (the default command of that workload)
#   2. PMC passes, each in its own run with --kernel-trace only, of one step of the same workload:
#        rdsize  TCC_EA0_RDREQ{,_32B,_64B,_128B}_sum   read requests L2 -> fabric by size: bytes = 32 a + 64 b + 128 c  (every request
#                                                        of this part is a 128-B fetch: profiles/pmc_calibration.json)
#        write   WRITE_SIZE + write / atomic requests
#        sq      instruction counts
# "atomic" collects the direct kernel (HAST_KC_COUNT=atomic) instead of the default path of the table's size.
# profiles/summarize_s00.py <tag> turns gpurun_out/prof_<tag>/ into profiles/<tag>_s00_pmc.json + pmc_traffic_s00[_atomic].json.
set -u
TAG=${1:?tag}
MODE=${2:-default}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
if [ "$MODE" = atomic ]; then export HAST_KC_COUNT=atomic; fi
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
echo "$MODE" > $OUT/mode.txt
python3 -c "import bench; print(bench.kc_source_id())" > $OUT/kernel_source_id.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --workload s00 --cpu-seconds 0 > $OUT/stats_bench.json 2> $OUT/stats_bench.err
declare -A PASS
PASS[rdsize]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
PASS[write]="WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_64B_sum"
PASS[sq]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for name in rdsize write sq; do
  rocprofv3 --pmc ${PASS[$name]} --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 bench.py --workload s00 --cpu-seconds 0 --steps 1 --warmup 0 > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
find $OUT -name "*counter_collection.csv" | wc -l

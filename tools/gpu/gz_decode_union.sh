# round 6: how much of the decode kernels' summed duration is the same wall time twice -- two .gz files are open at once, their passes run on
# two streams, and rocprofv3's kernel statistics add the launches' durations up.  One traced run of the 20M-read pair (two gzip -6 members):
# per k_gz_* kernel the sum of the launches' durations, the length of the UNION of their intervals, and the busiest overlap.
# usage: gpurun -- 'bash tools/gpu/gz_decode_union.sh > gpurun_out/gz_decode_union.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$(mktemp -d /dev/shm/hast_dun.XXXXXX); trap 'rm -rf $D' EXIT
for q in ${QUALS:-const noisy}; do
  [ $q = noisy ] && export GEN_FASTQ_QUAL=noisy
  tools/gen_fastq $D 10000000 5000000 100000 21 150 32 0 || exit 1
  tools/pgzip1 $D/r1.fq $D/r1.fq.gz 6 16 32; tools/pgzip1 $D/r2.fq $D/r2.fq.gz 6 16 32
  ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04 -t 32 --stats --read $D/r1.fq.gz --read $D/r2.fq.gz"
  rocprofv3 --kernel-trace --output-format csv -d $D/prof -- hast_amd/classify $ARGS > $D/out 2> $D/err
  echo "== quality lines: $q; md5=$(md5sum < $D/out | cut -c1-12) $(grep -o "read_phase_s=[0-9.]*" $D/err)"
  python3 - $(ls $D/prof/*/*kernel_trace.csv | head -1) <<'PYEOF'
import csv, sys, collections
iv = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].split("::")[-1].replace("void ", "")
    iv[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
def union(v):
    v = sorted(v); tot = 0; cs, ce = v[0]
    for s, e in v[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
allv = [x for v in iv.values() for x in v]
print("   all kernels: %d launches, sum %.1f ms, union %.1f ms, first start to last end %.1f ms" % (len(allv), sum(e - s for s, e in allv) / 1e6, union(allv) / 1e6, (max(e for s, e in allv) - min(s for s, e in allv)) / 1e6))
for n, v in sorted(iv.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    if not (n.startswith("k_gz_") or n.startswith("k_classify")): continue
    print("   %-18s launches %5d  sum %8.1f ms  union %8.1f ms  (sum / union %.2f)" % (n, len(v), sum(e - s for s, e in v) / 1e6, union(v) / 1e6, sum(e - s for s, e in v) / union(v)))
PYEOF
  rm -rf $D/prof
done

#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (profiles/collect_s00.sh) -> profiles/<tag>_kernel_stats.csv, <tag>_pmc.json and the per-read HBM traffic of
stage 00's counting kernels that `bench.py --workload s00` reports as roofline.traffic / roofline.frac: pmc_traffic_s00.json (the
default path of the bench's table: partitioned counting) or pmc_traffic_s00_atomic.json (HAST_KC_COUNT=atomic).

Read bytes = 32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B (TCC_EA0_RDREQ_*_sum; on gfx950 every L2->fabric read request is a
128-B fetch, profiles/pmc_calibration.json -- FETCH_SIZE prices a request at 64 B and reads half of what moved, which is what the
round-1 profile of this kernel did); write bytes = WRITE_SIZE x 1024.  Summed over every launch of the counting kernels in ONE step
of the bench (both parents), divided by the reads of that step."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
mode = open(os.path.join(src, "mode.txt")).read().strip()
COUNTING = ("k_kc_count", "k_kc_emit4", "k_kc_part", "k_kc_apply", "k_kc_spill")


def short(name):
    return name.split("(")[0].replace("void ", "").replace("hast::", "")


per_kernel = {}
for d in sorted(os.listdir(src)):
    if not (d.startswith("pmc_") and os.path.isdir(os.path.join(src, d))):
        continue
    f = sorted(glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)
    if not f:
        continue
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        if not k.startswith(COUNTING):
            continue
        e = per_kernel.setdefault(k, {})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    # launches of a kernel in the step
    seen = {}
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        if k.startswith(COUNTING):
            seen.setdefault(k, set()).add(r["Dispatch_Id"])
    for k, v in seen.items():
        per_kernel[k]["launches"] = len(v)
stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(max(stats, key=os.path.getsize), os.path.join(dst, tag + "_kernel_stats.csv"))
bench = json.loads(open(os.path.join(src, "stats_bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w"), indent=1)
one = json.loads(open(os.path.join(src, "pmc_rdsize.json")).read().strip().splitlines()[-1])
reads = 2 * one["config"]["reads_per_parent"]
L, K = 150, 21
tot = {}
for k, c in per_kernel.items():
    c["read_bytes"] = 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0)
    c["write_bytes"] = 1024 * c.get("WRITE_SIZE", 0)
    for name in ("read_bytes", "write_bytes", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_ATOMIC_sum", "SQ_INSTS_VALU"):
        tot[name] = tot.get(name, 0.0) + c.get(name, 0.0)
summary = {
    "tag": tag, "mode": mode, "partitioned": bool(one["counting"]["partitioned"]),
    "kernel_source_id": open(os.path.join(src, "kernel_source_id.txt")).read().strip(),
    "command": "python3 bench.py --workload s00 --cpu-seconds 0 --steps 1 --warmup 0 (one rocprofv3 --pmc pass per counter group, --kernel-trace only)",
    "reads_in_the_step": reads, "windows_in_the_step": reads * (L - K + 1),
    "per_kernel_sums_over_the_step": per_kernel, "totals": tot,
    "hbm_bytes_per_read": (tot["read_bytes"] + tot["write_bytes"]) / reads,
    "hbm_read_requests_per_read": tot["TCC_EA0_RDREQ_sum"] / reads,
    "hbm_write_requests_per_read": tot["TCC_EA0_WRREQ_sum"] / reads,
    "atomics_per_read": tot["TCC_EA0_ATOMIC_sum"] / reads,
    "valu_lane_instructions_per_window": tot["SQ_INSTS_VALU"] * 64 / (reads * (L - K + 1)),
}
json.dump(summary, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)
json.dump({"workload": "s00", "partitioned": summary["partitioned"], "kernel_source_id": summary["kernel_source_id"],
           "hbm_bytes_per_read": summary["hbm_bytes_per_read"], "hbm_read_requests_per_read": summary["hbm_read_requests_per_read"],
           "hbm_write_requests_per_read": summary["hbm_write_requests_per_read"], "atomics_per_read": summary["atomics_per_read"],
           "valu_lane_instructions_per_window": summary["valu_lane_instructions_per_window"],
           "valu_lane_instructions_per_window_by_kernel": {k: c.get("SQ_INSTS_VALU", 0.0) * 64 / summary["windows_in_the_step"] for k, c in per_kernel.items()},
           "source": "profiles/%s_pmc.json: rocprofv3 --pmc, separate passes, summed over the counting kernels of one bench step; reads = requests by "
                     "size (all 128 B on gfx950), writes = WRITE_SIZE" % tag},
          open(os.path.join(dst, "pmc_traffic_s00%s.json" % ("" if summary["partitioned"] else "_atomic")), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "per_kernel_sums_over_the_step"}, indent=1))
for k, c in per_kernel.items():
    print(k, {n: "%.4g" % v for n, v in sorted(c.items())})

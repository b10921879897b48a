#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by profiles/collect.sh on the GPU box) into the small tracked summaries under
profiles/:  <tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json, <tag>_pmc.json, and the per-launch HBM traffic of
k_classify that bench.py reports as roofline.traffic / roofline.frac: pmc_traffic.json (default command) or
pmc_traffic_<flags>.json (e.g. --clustered).

HBM read bytes of a launch, from counters only (no constants from elsewhere):
    by_size  = 32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B          (TCC_EA0_RDREQ_*_sum, pass "rdsize")
    dram_32B = 32 * TCC_EA0_RDREQ_DRAM_32B_sum                              (pass "rddram")
    fetch    = FETCH_SIZE * 1024                                            (tallies 128-B requests at 64 B on gfx950)
Each is checked on tools/hbm_randread runs of KNOWN byte count: random 128-B blocks (every byte of a request is asked
for) and random 64-B lines.  What the counters show on gfx950 (profiles/pmc_calibration.json): EVERY L2->fabric read
request is a 128-B request (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ, RDREQ_DRAM_32B == 4 per request) -- a random 64-B line
costs a 128-B fetch, so for the 64-B run the counters read exactly twice the bytes the kernel asked for, and FETCH_SIZE
(which prices a request at 64 B) reads half of what moved.  HBM traffic = bytes MOVED = the first method whose factor on the
128-B run is within 3 % of 1 and which agrees with the DRAM-side count; the 64-B run's factor is recorded as the evidence
for the fetch granule.
"""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")


def kernel_source_id():
    h = hashlib.sha256()
    for f in ("hast_kernels.hip", "hast_filter.hip", "hast_common.h", "hast_devutil.h", "hast_device.h"):
        h.update(open(os.path.join(ROOT, "hast_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def counters(d, want):
    """last dispatch's value of every counter of the first kernel whose name contains `want`"""
    # (gpurun MERGES a call's files into gpurun_out/: a tag collected twice leaves the older run's files next to the newer ones)
    f = sorted(glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)
    out = {}
    if not f:
        return out
    for r in csv.DictReader(open(f[0])):
        if want in r["Kernel_Name"]:
            out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: v[-1] for k, v in out.items()}


def read_bytes(c):
    out = {}
    if "TCC_EA0_RDREQ_64B_sum" in c:
        out["by_size"] = 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c["TCC_EA0_RDREQ_64B_sum"] + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0)
    if "TCC_EA0_RDREQ_DRAM_32B_sum" in c:
        out["dram_32B"] = 32 * c["TCC_EA0_RDREQ_DRAM_32B_sum"]
    if "FETCH_SIZE" in c:
        out["fetch"] = 1024 * c["FETCH_SIZE"]
    return out


flags = open(os.path.join(src, "flags.txt")).read().strip() if os.path.exists(os.path.join(src, "flags.txt")) else ""
suffix = "".join(ch for ch in flags.replace("--", "_").replace(" ", "") if ch.isalnum() or ch == "_")
stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
if stats:
    # rocprofv3 also traces the child processes bench.py starts for its cpu_baseline leg (the product CLI on the reference's
    # files): the bench process is the one with the most kernel time
    def total(f):
        return sum(float(r["TotalDurationNs"]) for r in csv.DictReader(open(f)))
    shutil.copy(max(stats, key=total), os.path.join(dst, tag + "_kernel_stats.csv"))
bench = json.loads(open(os.path.join(src, "stats_bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w"), indent=1)

kc = {}
for d in sorted(os.listdir(src)):
    if d.startswith("pmc_") and os.path.isdir(os.path.join(src, d)):
        kc.update(counters(d, "k_classify"))
cal = {}
for line in (64, 128):
    c = {}
    known = None
    for d in sorted(os.listdir(src)):
        if d.startswith("cal%d_" % line) and os.path.isdir(os.path.join(src, d)):
            c.update(counters(d, "k_rand"))
            js = os.path.join(src, d + ".json")
            if os.path.exists(js) and open(js).read().strip():
                known = json.loads(open(js).read().strip().splitlines()[-1])["bytes"]
    if c and known:
        cal[str(line)] = {"known_bytes": known, "counters": c,
                          "factor_known_over_counter": {m: known / v for m, v in read_bytes(c).items() if v}}
if not cal:            # flag runs (--clustered ...) reuse the default run's calibration of the same tag family
    base = os.path.join(dst, tag.rstrip("abcdefghijklmnopqrstuvwxyz_") + "_pmc.json")
    for cand in (os.path.join(dst, "pmc_calibration.json"), base):
        if os.path.exists(cand):
            cal = json.load(open(cand)).get("calibration", {})
            if cal:
                break
rb = read_bytes(kc)
method = None
for m in ("by_size", "dram_32B", "fetch"):
    f128 = cal.get("128", {}).get("factor_known_over_counter", {}).get(m)
    if m in rb and rb[m] and f128 and abs(f128 - 1) < 0.03:
        if m == "by_size" and "dram_32B" in rb and abs(rb["by_size"] / rb["dram_32B"] - 1) > 0.01:
            continue
        method = m
        break
write = 1024 * kc.get("WRITE_SIZE", 0)
summary = {
    "tag": tag, "bench_flags": flags,
    # identifies the device code the box ran (written there by collect.sh); the local tree's id when that file is missing
    "kernel_source_id": (open(os.path.join(src, "kernel_source_id.txt")).read().strip()
                         if os.path.exists(os.path.join(src, "kernel_source_id.txt")) else kernel_source_id()),
    "command": "python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 %s (one PMC pass per counter group, --kernel-trace only)" % flags,
    "k_classify_per_launch": kc,
    "read_bytes_by_method": rb, "method_used": method, "calibration": cal,
    "write_bytes": write,
    "batch_reads": bench["config"]["batch_reads"],
}
if method:
    fetch = rb[method]
    req = kc.get("TCC_EA0_RDREQ_sum")
    summary.update({
        "hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch,
        "hbm_read_requests_per_launch": req, "bytes_per_read_request": fetch / req if req else None,
        "requests_per_read": req / bench["config"]["batch_reads"] if req else None,
        "algorithmic_bytes_per_launch": bench["roofline"].get("algorithmic_bytes_per_launch") or bench["roofline"].get("algorithmic_equiv", {}).get("bytes_per_launch"),
    })
    if "TCC_HIT_sum" in kc:
        summary["l2_hit_rate"] = kc["TCC_HIT_sum"] / (kc["TCC_HIT_sum"] + kc["TCC_MISS_sum"])
    if "SQ_INSTS_VALU" in kc:
        L, K = bench["config"]["read_len"], bench["config"]["k"]
        summary["valu_lane_instructions_per_window"] = kc["SQ_INSTS_VALU"] * 64 / (bench["config"]["batch_reads"] * (L - K + 1))
    wl = flags.split("--workload")[1].split()[0] if "--workload" in flags else "c3"
    json.dump({"workload": wl, "bench_flags": flags, "batch_reads": bench["config"]["batch_reads"],
               "kernel_source_id": summary["kernel_source_id"],
               "hbm_bytes_per_launch": fetch + write, "hbm_read_requests_per_launch": req,
               "bytes_per_read_request": summary["bytes_per_read_request"],
               "source": "profiles/%s_pmc.json: rocprofv3 --pmc, separate passes; read bytes by method '%s' (calibrated on tools/hbm_randread: "
                         "known/counted = %s; every read request on gfx950 is a 128-B fetch), + WRITE_SIZE" % (tag, method, {l + "-B run": round(cal[l]["factor_known_over_counter"][method], 4) for l in cal})},
              open(os.path.join(dst, "pmc_traffic%s.json" % (("_" + suffix.strip("_")) if suffix else "")), "w"), indent=1)
json.dump(summary, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)
if cal and not flags:
    json.dump({"tag": tag, "calibration": cal}, open(os.path.join(dst, "pmc_calibration.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k not in ("calibration",)}, indent=1))

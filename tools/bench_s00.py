#!/usr/bin/env python3
"""Stage-00 measurement (not the repo's headline bench, which is bench.py / stage 01): counting the k-mers of a synthetic
trio's reads into the HBM count table with the reads resident in HBM, then the histogram / selection passes.

    python tools/bench_s00.py [--genome 200e6] [--coverage 30] [--read-len 150] [--k 21] [--batch-reads 8e6]
                              [--table-gb 0=auto] [--cpu-seconds 15]

Prints ONE JSON line.  value = bases of both parents' reads counted per second (whole job: every read once, table
empty at the start).  roofline: the count kernel is bound by random 128-B read-modify-write lines in HBM; algorithmic
bytes per window = 256 (one line read + written back), per read = L + 1 + (L-K+1)*256.
cpu_baseline: the oracle's counter (oracle/s00_oracle.c, 1 thread) on a sample of the same stream.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hast_amd  # noqa: E402
from hast_amd import KcSynth, KmerCounter  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=float, default=200e6)
    ap.add_argument("--coverage", type=float, default=30)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--batch-reads", type=float, default=8e6)
    ap.add_argument("--table-gb", type=float, default=0)
    ap.add_argument("--snp-per-1024", type=int, default=1)
    ap.add_argument("--err-per-4096", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=15)
    ap.add_argument("--passes", type=int, default=1, help="count the whole input this many times (table cleared in between); the last is reported")
    a = ap.parse_args()
    hast_amd.build()
    L, k = a.read_len, a.k
    g = KcSynth(0, int(a.genome), L, a.snp_per_1024, a.err_per_4096, 20)
    n_reads = int(a.genome * a.coverage / L)
    batch = int(a.batch_reads)
    n_batches = (n_reads + batch - 1) // batch
    rec = L + 1
    with hast_amd.Context(k) as ctx:
        # all reads of both parents resident in HBM (generated on the device), the count table in the rest
        bufs = []
        gen = KmerCounter(k, table_bytes=1 << 20)
        t0 = time.time()
        for p in (1, 0):
            for b in range(n_batches):
                n = min(batch, n_reads - b * batch)
                d = ctx.alloc(n * rec)
                gen.synth_device(g, p, b * batch, n, d)
                bufs.append((p, d, n))
        gen.sync()
        t_gen = time.time() - t0
        sample = ctx.to_host(bufs[0][1], (min(bufs[0][2], 400_000) * rec,), np.uint8)
        gen.close()
        kc = KmerCounter(k, table_bytes=int(a.table_gb * (1 << 30)))
        for _ in range(a.passes):
            kc.set_slice(0, 1)
            kc.sync()
            t0 = time.time()
            for p, d, n in bufs:
                kc.count_device(p, d, n * rec)
            kc.sync()
            t_count = time.time() - t0
        st = kc.stats()
        t0 = time.time()
        h = [kc.histo(p) for p in (0, 1)]
        t_histo = time.time() - t0
        bounds = [hast_amd.kc_find_bounds(x) for x in h]
        t0 = time.time()
        n_sel = [kc.select(p, max(1, bounds[p][2]), max(1, bounds[p][3])) for p in (0, 1)]
        t_select = time.time() - t0
        kc.release_table()
        t0 = time.time()
        n_sorted = [kc.selection_sort(p) for p in (0, 1)]
        text0 = kc.selection_text(0, 0, min(n_sorted[0], 1 << 20))
        t_sort = time.time() - t0
        kc.close()
    bases = 2 * n_reads * L
    windows = 2 * n_reads * (L - k + 1)
    alg_bytes = 2 * n_reads * (rec + (L - k + 1) * 256)
    out = {
        "metric": "parental read-bp/sec counted into the k-mer table at k=%d, %dbp reads" % (k, L),
        "value": bases / t_count, "unit": "bp/s", "n_gpus": 1, "higher_is_better": True, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "S00 synthetic trio: %.0f Mbp genome, %gx coverage per parent, %d-bp reads, K=%d" % (a.genome / 1e6, a.coverage, L, k),
                   "reads_per_parent": n_reads, "batch_reads": batch, "snp_per_1024": a.snp_per_1024, "err_per_4096": a.err_per_4096,
                   "table_slots": st["capacity"], "table_gb": st["capacity"] * 16 / 2**30, "load_factor": st["keys"] / st["capacity"]},
        "seconds": {"generate": t_gen, "count": t_count, "histo_x2": t_histo, "select_x2": t_select, "sort_format": t_sort},
        "kmers": {"counted": list(st["total"]), "distinct": list(st["distinct"]), "union": st["keys"], "bounds": [list(b) for b in bounds],
                  "selected": n_sel, "first_row": text0[:k].decode()},
        "roofline": {"bound": "hbm", "kernel": "k_kc_count", "achieved": alg_bytes / t_count / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg_bytes / t_count / 1e9 / 8000.0, "traffic": None,
                     "algorithmic_bytes": alg_bytes, "windows_per_s": windows / t_count},
    }
    tf = os.path.join(ROOT, "profiles", "pmc_traffic_s00.json")
    if os.path.exists(tf):                       # measured HBM traffic of the count kernel (rocprofv3 PMC), scaled per read
        t = json.load(open(tf))
        rf = out["roofline"]
        rf["traffic"] = t["hbm_bytes_per_read"] * 2 * n_reads
        rf["traffic_source"] = t["source"]
        tx = (t["hbm_read_requests_per_read"] + t["hbm_write_requests_per_read"]) * 2 * n_reads / t_count
        rf["hbm_transactions_per_s"] = tx
        rf["hbm_line_rate_ceiling"] = 48e9      # tools/hbm_randread: random 64-B lines per second this part sustains
        rf["hbm_line_rate_frac"] = tx / 48e9
    assert st["total"][0] + st["total"][1] <= windows
    # CPU baseline: the oracle's counter on a sample of the maternal stream (1 thread)
    if a.cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from tests import oracle_binding
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], stdout=subprocess.DEVNULL, check=True)
        o = oracle_binding.load(os.path.join(ROOT, "oracle", "liboracle.so"))
        c = o.ho_s00_new(k)
        done, t0 = 0, time.time()
        step = 20_000 * rec
        while done < sample.size and time.time() - t0 < a.cpu_seconds:
            part = sample[done:done + step]
            o.ho_s00_add_stream(c, 1, part.ctypes.data, part.size)
            done += part.size
        dt = time.time() - t0
        out["cpu_baseline"] = {"value": done / rec * L / dt, "unit": "bp/s", "cores": 1, "kind": "port",
                               "sample": "first %d reads of the maternal stream (%.1f s), oracle/s00_oracle.c -O2" % (done // rec, dt)}
        o.ho_s00_free(c)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

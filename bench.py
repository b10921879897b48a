#!/usr/bin/env python3
"""bench.py -- classified read-bp/s of the HAST stage-01 hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path (hast_classify_device -> k_classify) over one HBM-resident
batch of synthetic 150-bp reads; every step uses a different batch (counter-based generator,
SURVEY 8(d)).  Default workload = the configuration the metric is quoted on: K=21, 200M+200M
synthetic parental 21-mers (merged 6.4 GB table), 10M barcodes.  Tables are replicated per GPU, reads
are sharded by index (weak scaling: every rank classifies steps*batch reads of its own), and the
per-barcode counters are combined with ONE RCCL all-reduce + one D2H inside the timed region.

One JSON line on stdout (rank 0).  Extra objects:
  roofline     k_classify against the HBM roofline: algorithmic bytes (150 + 130*64 per read,
               SURVEY 8(d)) per launch / average launch duration from HIP events on the launch stream.
  cpu_baseline the oracle's CPU restatement of the reference algorithm ("port": two hash sets, two
               probes per k-mer, t worker threads) on this box's host cores, on a bounded sample of the
               same reads against the same full-size sets; a reported baseline, not the target.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n_keys_per_hap, n_barcodes, description)
    "c3": (200_000_000, 10_000_000, "C3 human-trio scale: 200M+200M synthetic 21-mers, 10M barcodes, 150bp reads"),
    "c2": (50_000_000, 1_000_000, "C2: 50M+50M synthetic 21-mers, 1M barcodes, 150bp reads"),
    "c1": (1_000_000, 10_000, "C1 plumbing: 1M+1M synthetic 21-mers, 10k barcodes, 150bp reads"),
    # config 5: K=31, 400M+400M keys, PacBio-style 20 kb reads, barcode-free per-read (hits0, hits1) output
    # (string semantics of the stage-03 per-read classifier); pass --k 31 --read-len 20000 --batch-reads 120000
    "c5": (400_000_000, 1, "C5 HAST4TGS-style: 400M+400M synthetic 31-mers, 20kb reads, per-read assignment"),
}
ADAPTOR_F = b"CTGTCTCTTATACACATCTTAGGAAGACAAGCACTGACGACATGA"   # classify.cpp:312
ADAPTOR_R = b"TCTGCTGAGTCGAGAACGTCTCTGTGAGCCAAGGAGTTGCTCTGG"   # classify.cpp:313
HBM_PEAK_GBS = 8000.0                                           # MI355X_MICROARCH.md: 8.0 TB/s spec


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 20 (s00: 3)")
    ap.add_argument("--warmup", type=int, default=None, help="default 3 (s00: 1)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["s00"], default="c3",
                    help="c1/c2/c3/c5: stage-01 classification (the BASELINE metric); s00: stage-00 k-mer counting (SURVEY 8(f) #4)")
    ap.add_argument("--genome", type=float, default=200e6, help="s00: genome length of the synthetic trio")
    ap.add_argument("--coverage", type=float, default=30, help="s00: coverage per parent")
    ap.add_argument("--table-gb", type=float, default=60, help="s00: size of the count table (0 = 85 %% of the free HBM)")
    ap.add_argument("--batch-reads", type=int, default=16_000_000)
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--load-factor", type=float, default=0.2)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline work (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores of this process")
    ap.add_argument("--minimizer", type=int, default=0, help="minimizer length for bucket placement (0 = library default)")
    ap.add_argument("--clustered", action="store_true", help="keys in runs of K around variant sites (real-data structure) instead of independent random keys")
    ap.add_argument("--no-plants", action="store_true", help="reads without planted parental k-mers (isolates table effects)")
    ap.add_argument("--barcodes", type=int, default=0, help="override the workload's barcode count")
    ap.add_argument("--keys-per-hap", type=int, default=0, help="override the workload's key count per haplotype")
    ap.add_argument("--max-resident-gb", type=float, default=96.0, help="HBM budget for resident read batches")
    args = ap.parse_args()
    s00 = args.workload == "s00"
    args.steps = args.steps if args.steps is not None else (3 if s00 else 20)
    args.warmup = args.warmup if args.warmup is not None else (1 if s00 else 3)

    # The contract is ONE JSON line on stdout.  Libraries chat on fd 1 (RCCL prints its version banner there under
    # NCCL_DEBUG=VERSION), so fd 1 is pointed at stderr for the whole run and the JSON line goes to the saved stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    if s00:
        return bench_s00(args, real_stdout)
    import numpy as np
    import torch                      # first: libhast then binds to the HIP runtime torch already loaded
    import torch.distributed as dist
    import hast_amd
    from hast_amd.binding import make_params, B_ALG_PER_READ
    from hast_amd.sharding import shard_first_read

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (there is no CPU path)")
    # test switches for boxes with ONE GPU: HAST_BENCH_SHARE_GPU=1 puts every rank on device 0 and
    # HAST_BENCH_BACKEND=gloo moves the collectives to the CPU, so the N>1 control flow (sharding by rank, barrier,
    # max-over-ranks timing, rank-0 output) can be run end to end with 2 processes on a 1-GPU box
    if os.environ.get("HAST_BENCH_SHARE_GPU"):
        local_rank = 0
    backend = os.environ.get("HAST_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # HAST_BENCH_FORCE_DIST=1 runs the RCCL init / barrier / all-reduce path even at world size 1 (a check of the
    # N>1 code on a 1-GPU box; launch through torch.distributed.run so RANK/MASTER_* exist)
    use_dist = world > 1 or bool(os.environ.get("HAST_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n_keys, n_bc, wl_desc = WORKLOADS[args.workload]
    if args.barcodes or args.keys_per_hap:
        n_bc, n_keys = args.barcodes or n_bc, args.keys_per_hap or n_keys
        wl_desc += " (overridden: %d keys/hap, %d barcodes)" % (n_keys, n_bc)
    perread = args.workload == "c5"
    if perread:                       # config 5 geometry unless overridden on the command line
        if args.k == 21: args.k = 31
        if args.read_len == 150: args.read_len = 20000
        if args.batch_reads == 16_000_000: args.batch_reads = 120_000
        args.cpu_seconds = 0          # the port's CPU leg restates stage 01 only
    K, L, R = args.k, args.read_len, args.batch_reads
    p = make_params(K, L, n_keys, n_bc, clustered=args.clustered, no_plants=args.no_plants)
    ctx = hast_amd.Context(K, local_rank, minimizer=args.minimizer or None)
    stream = torch.cuda.Stream(device=dev)
    hs = C.c_void_p(stream.cuda_stream)

    # ---- table (replicated per GPU): generate + insert on the device, scrub adaptors ------------
    t0 = time.time()
    ctx.table_reserve(2 * n_keys, args.load_factor)
    ctx.synth_table_build(p)
    akeys = []
    for ad in (ADAPTOR_F, ADAPTOR_R):
        for kmer in hast_amd.chop_read(ad, K):
            if kmer not in akeys:
                akeys.append(kmer)
    ctx.table_erase(np.array(akeys, dtype=np.uint64))
    set_sizes = ctx.table_sizes()
    n_buckets, table_bytes = ctx.table_info()
    t_table = time.time() - t0
    if rank == 0:
        log("table: %d+%d keys -> sets %s, %d buckets, %.2f GB, built in %.3f s" %
            (n_keys, n_keys, set_sizes, n_buckets, table_bytes / 1e9, t_table))

    # ---- counters: caller-owned torch tensor so torch.distributed can all-reduce it -------------
    counts = torch.zeros((n_bc, 4), dtype=torch.int32, device=dev)
    ctx.counts_bind(counts.data_ptr(), n_bc)

    # ---- HBM-resident read batches, all distinct (rank r, step s -> reads [(s*world+r)*R, +R)) ----
    n_total = args.steps + args.warmup
    per_batch = R * L + R * 4
    n_res = max(1, min(n_total, int(args.max_resident_gb * 1e9 // per_batch)))
    t0 = time.time()
    batches = []
    for j in range(n_res):
        b = torch.empty(R * L + 64, dtype=torch.uint8, device=dev)
        ids = torch.empty(R, dtype=torch.int32, device=dev)
        ctx.synth_reads_device(p, shard_first_read(j, world, rank, R), R, b.data_ptr(), ids.data_ptr(), hs)
        batches.append((b, ids))
    stream.synchronize()
    if rank == 0:
        log("%d resident batches x %d reads (%.2f GB) generated in %.1f s" % (n_res, R, n_res * per_batch / 1e9, time.time() - t0))

    votes = offsets = None
    if perread:
        votes = torch.zeros((R, 2), dtype=torch.int32, device=dev)
        offsets = (torch.arange(R + 1, dtype=torch.int64, device=dev) * L)

    def step(j):
        b, ids = batches[j % n_res]
        if perread:
            ctx.classify_perread_device(b.data_ptr(), R * L, offsets.data_ptr(), R, votes.data_ptr(), stream=hs)
        else:
            ctx.classify_device(b.data_ptr(), R * L, R, L, d_barcode_ids=ids.data_ptr(), stream=hs)

    def barrier():
        if use_dist:
            if backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    host_out = torch.empty((R, 2) if perread else (n_bc, 4), dtype=torch.int32, pin_memory=True)

    # ---- warmup -----------------------------------------------------------------------------------
    for j in range(args.warmup):
        step(j)
    stream.synchronize()
    counts.zero_()
    torch.cuda.synchronize()

    # ---- timed region: K steps + ONE all-reduce + D2H of the counters -------------------------------
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ctx.classify_timing(args.steps)          # HIP events around k_classify and k_commit_votes, on the launch stream, inside the library
    barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    with torch.cuda.stream(stream):
        for s in range(args.steps):
            ev0[s].record(stream)
            step(args.warmup + s)
            ev1[s].record(stream)
        if use_dist and not perread:
            if backend == "nccl":
                dist.all_reduce(counts, op=dist.ReduceOp.SUM)      # RCCL, uint32-as-int32 sums
            else:                                                  # test backend: reduce on the host
                stream.synchronize()
                tmp = counts.cpu()
                dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
                counts.copy_(tmp)
        # per-read mode needs no reduction: every rank returns its own reads' (hits0, hits1)
        counts_host = None
        if rank == 0 or perread:
            counts_host = host_out.copy_(votes if perread else counts, non_blocking=True)   # pinned D2H
    stream.synchronize()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    call_ms = [a.elapsed_time(b) for a, b in zip(ev0, ev1)]           # the whole hast_classify_device call
    kern_ms, commit_ms = ctx.classify_times(args.steps)                # k_classify alone / k_commit_votes alone
    ctx.classify_timing(0)
    if len(kern_ms) != args.steps:
        kern_ms, commit_ms = call_ms, [0.0] * len(call_ms)
    kern_avg_ms = sum(kern_ms) / len(kern_ms)

    total_bp = world * args.steps * R * L
    value = total_bp / elapsed
    b_alg = B_ALG_PER_READ(L, K) * R                                # algorithmic bytes per launch
    achieved = b_alg / (kern_avg_ms * 1e-3) / 1e9

    result = None
    if rank == 0:
        ch = counts_host.numpy().view(np.uint32)
        if perread:
            ch = np.concatenate([ch, np.zeros((ch.shape[0], 1), np.uint32)], axis=1)
        result = {
            "metric": "classified read-bp/sec at k=%d, %dbp reads, %dM unique-mers/hap" % (K, L, n_keys // 1_000_000),
            "value": value, "unit": "bp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": wl_desc, "k": K, "read_len": L, "keys_per_hap": n_keys, "barcodes": n_bc,
                       "batch_reads": R, "reads_total": world * args.steps * R, "table_gb": round(table_bytes / 1e9, 3),
                       "load_factor": args.load_factor, "minimizer": ctx.minimizer, "keys": "clustered" if args.clustered else "random", "set_sizes": list(set_sizes), "sharding": "reads by index, tables replicated",
                       "collective": "1x all_reduce(sum,u32[%d]) + D2H in timed region" % (n_bc * 4) if use_dist else "none (D2H of counters in timed region)",
                       "resident_batches": n_res},
            "mode": "per-read votes (stage-03 semantics)" if perread else "per-barcode counts (stage-01 semantics)",
            "roofline": {"bound": "hbm", "kernel": "k_classify", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": b_alg, "kernel_ms_avg": kern_avg_ms,
                         "kernel_ms_min": min(kern_ms), "kernel_ms_max": max(kern_ms),
                         "commit_kernel_ms_avg": sum(commit_ms) / len(commit_ms), "call_ms_avg": sum(call_ms) / len(call_ms)},
            "hits": {"c0": int(ch[:, 0].sum(dtype=np.uint64)), "c1": int(ch[:, 1].sum(dtype=np.uint64)),
                     "neg_reads": int(ch[:, 2].sum(dtype=np.uint64))},
        }
        traffic = _committed_traffic(args, R)
        if traffic:
            rf = result["roofline"]
            rf.update(traffic)
            # what the kernel really moves, next to the algorithmic figure: bytes/s of HBM traffic, and 64-B
            # requests/s against the measured random-line ceiling of this chip (tools/hbm_randread: ~48 G lines/s)
            rf["traffic_GBps"] = traffic["traffic"] / (kern_avg_ms * 1e-3) / 1e9
            if traffic.get("hbm_read_requests_per_launch"):
                rf["hbm_lines_per_s"] = traffic["hbm_read_requests_per_launch"] / (kern_avg_ms * 1e-3)
                rf["hbm_line_rate_ceiling"] = 48e9
                rf["hbm_line_rate_frac"] = rf["hbm_lines_per_s"] / 48e9

    # ---- CPU baseline (rank 0, N=1 only): the oracle's port on this box's host cores -----------------
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        try:
            result["cpu_baseline"] = cpu_baseline(args, ctx, p, batches[0], set_sizes, hs, stream)
        except Exception as e:                                       # the baseline never gates the GPU number
            result["cpu_baseline"] = {"value": None, "unit": "bp/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0:
        real_stdout.write(json.dumps(result) + "\n")
        real_stdout.flush()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()



def bench_s00(args, real_stdout):
    """Stage 00 (not the BASELINE metric): a "step" is one whole counting job -- every read of both parents
    of a synthetic trio, resident in HBM, counted once into the emptied table (k_kc_count); value = bases counted per
    second.  Roofline: HBM transactions; algorithmic bytes per window = 256 (one 128-B bucket line read + written back).
    cpu_baseline = the oracle's counter (oracle/s00_oracle.c, one thread) on a sample of the same stream."""
    import numpy as np
    import hast_amd
    from hast_amd import KcSynth, KmerCounter
    if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
        sys.exit("bench.py --workload s00 measures one GPU (several GPUs split the key space: DESIGN.md section 9)")
    hast_amd.build()
    L, k = args.read_len, args.k
    g = KcSynth(0, int(args.genome), L, 1, 20, 20)
    n_reads = int(args.genome * args.coverage / L)
    batch = min(args.batch_reads, 8_000_000)
    n_batches = (n_reads + batch - 1) // batch
    rec = L + 1
    with hast_amd.Context(k) as ctx:
        bufs = []
        gen = KmerCounter(k, table_bytes=1 << 20)
        for p in (1, 0):                                   # maternal first, as the reference script does
            for b in range(n_batches):
                n = min(batch, n_reads - b * batch)
                d = ctx.alloc(n * rec)
                gen.synth_device(g, p, b * batch, n, d)
                bufs.append((p, d, n))
        gen.sync()
        sample = ctx.to_host(bufs[0][1], (min(bufs[0][2], 400_000) * rec,), np.uint8)
        gen.close()
        kc = KmerCounter(k, table_bytes=int(args.table_gb * (1 << 30)))
        times = []
        for it in range(args.warmup + args.steps):
            kc.sync()
            t0 = time.perf_counter()
            kc.set_slice(0, 1)                             # empty table
            for p, d, n in bufs:
                kc.count_device(p, d, n * rec)
            kc.sync()
            if it >= args.warmup:
                times.append(time.perf_counter() - t0)
        t_count = sum(times) / len(times)
        st = kc.stats()
        t0 = time.perf_counter()
        h = [kc.histo(p) for p in (0, 1)]
        t_histo = time.perf_counter() - t0
        bounds = [hast_amd.kc_find_bounds(x) for x in h]
        t0 = time.perf_counter()
        n_sel = [kc.select(p, max(1, bounds[p][2]), max(1, bounds[p][3])) for p in (0, 1)]
        t_select = time.perf_counter() - t0
        kc.release_table()
        t0 = time.perf_counter()
        n_sorted = [kc.selection_sort(p) for p in (0, 1)]
        text0 = kc.selection_text(0, 0, min(n_sorted[0], 1 << 20))
        t_sort = time.perf_counter() - t0
        kc.close()
    bases = 2 * n_reads * L
    windows = 2 * n_reads * (L - k + 1)
    alg_bytes = 2 * n_reads * (rec + (L - k + 1) * 256)
    assert st["total"][0] + st["total"][1] <= windows
    out = {
        "metric": "parental read-bp/sec counted into the k-mer table at k=%d, %dbp reads" % (k, L),
        "value": bases / t_count, "unit": "bp/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_count * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "S00 synthetic trio: %.0f Mbp genome, %gx coverage per parent, %d-bp reads, K=%d" % (args.genome / 1e6, args.coverage, L, k),
                   "reads_per_parent": n_reads, "batch_reads": batch, "table_slots": st["capacity"], "table_gb": st["capacity"] * 16 / 2**30,
                   "load_factor": st["keys"] / st["capacity"]},
        "seconds": {"count": t_count, "count_min": min(times), "count_max": max(times), "histo_x2": t_histo, "select_x2": t_select, "sort_format": t_sort},
        "kmers": {"counted": list(st["total"]), "distinct": list(st["distinct"]), "union": st["keys"], "bounds": [list(b) for b in bounds],
                  "selected": n_sel, "first_row": text0[:k].decode()},
        "roofline": {"bound": "hbm", "kernel": "k_kc_count", "achieved": alg_bytes / t_count / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_bytes / t_count / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": alg_bytes,
                     "windows_per_s": windows / t_count},
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
    if args.cpu_seconds > 0:
        import subprocess
        from tests import oracle_binding
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], stdout=subprocess.DEVNULL, check=True)
        o = oracle_binding.load(os.path.join(ROOT, "oracle", "liboracle.so"))
        c = o.ho_s00_new(k)
        done, t0 = 0, time.perf_counter()
        step = 20_000 * rec
        while done < sample.size and time.perf_counter() - t0 < args.cpu_seconds:
            part = sample[done:done + step]
            o.ho_s00_add_stream(c, 1, part.ctypes.data, part.size)
            done += part.size
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": done / rec * L / dt, "unit": "bp/s", "cores": 1, "kind": "port",
                               "sample": "first %d reads of the maternal stream (%.1f s), oracle/s00_oracle.c -O2" % (done // rec, dt)}
        o.ho_s00_free(c)
    real_stdout.write(json.dumps(out) + "\n")
    real_stdout.flush()


def _committed_traffic(args, R):
    """HBM bytes per launch from the committed rocprofv3 PMC profile of this same command, if one exists
    (profiles/pmc_traffic.json, written by profiles/collect_pmc.sh); None otherwise."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
    except Exception:
        return None
    if d.get("workload") != args.workload or d.get("batch_reads") != R:
        return None
    out = {"traffic": d.get("hbm_bytes_per_launch"), "traffic_source": d.get("source")}
    if d.get("hbm_read_requests_per_launch"):
        out["hbm_read_requests_per_launch"] = d["hbm_read_requests_per_launch"]
    return out


def cpu_baseline(args, ctx, p, batch0, set_sizes, hs, stream):
    """TEST-INFRASTRUCTURE leg: times oracle/liboracle.so (CPU restatement of the reference algorithm)
    on a bounded sample of batch 0, against full-size sets built from the same synthetic keys, and
    checks the GPU's counts for that sample against it."""
    import subprocess
    import numpy as np
    import torch
    import hast_amd
    from tests import oracle_binding as ob
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True)
    o = ob.load(os.path.join(ROOT, "oracle", "liboracle.so"))
    threads = args.cpu_threads or len(os.sched_getaffinity(0))
    K, L, n_keys, n_bc = p.k, p.read_len, p.n_keys_per_hap, p.n_barcodes
    t0 = time.time()
    oc = o.ho_new()
    dk = torch.empty(n_keys, dtype=torch.int64, device=batch0[0].device)
    for h in (0, 1):
        ctx.synth_keys_device(p, h, 0, n_keys, dk.data_ptr(), hs)
        stream.synchronize()
        keys = dk.cpu().numpy()
        assert o.ho_load_keys_mt(oc, keys.ctypes.data, keys.size, h, K, threads) == 0
    del dk
    o.ho_init_adaptor(oc, ADAPTOR_F, ADAPTOR_R, None)
    assert (o.ho_set_size(oc, 0), o.ho_set_size(oc, 1)) == tuple(set_sizes), "CPU/GPU set sizes differ"
    log("cpu_baseline: sets built on %d threads in %.1f s" % (threads, time.time() - t0))
    bases = batch0[0][:args.batch_reads * L].cpu().numpy()
    ids = batch0[1].cpu().numpy().view(np.uint32)

    def run(n):
        off = np.arange(n + 1, dtype=np.uint64) * L
        e = [np.zeros(n_bc, np.uint32) for _ in range(3)]
        t = time.perf_counter()
        o.ho_classify_ids(oc, bases.ctypes.data, off.ctypes.data, ids.ctypes.data, n, e[0].ctypes.data, e[1].ctypes.data,
                          e[2].ctypes.data, None, threads)
        return time.perf_counter() - t, e

    n_probe = min(args.batch_reads, 20000 * threads)
    dt, _ = run(n_probe)
    rate = n_probe / dt
    # wall seconds of the sample so that (threads x wall) is a bounded amount of CPU work
    n = int(min(args.batch_reads, max(n_probe, rate * args.cpu_seconds)))
    dt, e = run(n)
    # parity of the same sample on the GPU
    ctx.counts_resize(n_bc)
    ctx.classify_device(batch0[0].data_ptr(), n * L, n, L, d_barcode_ids=batch0[1].data_ptr())
    g = ctx.counts_read(n_bc)
    parity = all(np.array_equal(a, b) for a, b in zip(g, e))
    o.ho_free(oc)
    return {"value": n * L / dt, "unit": "bp/s", "cores": threads, "kind": "port",
            "sample": "first %d reads of batch 0 (%.1f s wall on %d threads), full %dM+%dM-key sets, oracle/hast_oracle.c -O2"
                      % (n, dt, threads, n_keys // 1_000_000, n_keys // 1_000_000),
            "gpu_counts_match_cpu_on_sample": bool(parity)}


if __name__ == "__main__":
    main()

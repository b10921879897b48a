"""N>1 path on CPU (gloo, world_size 2): the read-shard arithmetic bench.py uses and the counter merge
(one all_reduce(sum) over the int64 view of the counters' three live words as arrays c0[n] | c1[n] | neg[n] -- hast_counts_pack: what
bench.py and hast_counts_allreduce move since round 5, 24 bytes per barcode) give exactly the single-process
counts -- also for a barcode whose counters stand beyond 2^32 on every rank (the hot no-barcode bucket of real stLFR data).
The per-shard classification here is done by the oracle (no GPU on this box); on the GPU box the same
merge runs over RCCL inside bench.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, steps, R, out_dir):
    sys.path.insert(0, ROOT)
    import hast_amd
    from hast_amd.binding import make_params
    from hast_amd.sharding import shard_first_read
    from tests import oracle_binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = ob.load(os.path.join(ROOT, "oracle", "liboracle.so"))
    k, L, n_keys, n_bc = 21, 150, 5000, 211
    p = make_params(k, L, n_keys, n_bc)
    oc = o.ho_new()
    for h in (0, 1):
        keys = hast_amd.synth_keys_host(p, h, 0, n_keys)
        o.ho_load_keys(oc, keys.ctypes.data, keys.size, h, k)
    counts = np.zeros((n_bc, 4), dtype=np.uint64)              # device layout {c0,c1,neg,reserved}, 64-bit
    counts[7, :3] = (1 << 32) + 5 + rank                        # a hot barcode: each rank already holds more than 2^32 of everything
    off = np.arange(R + 1, dtype=np.uint64) * L
    for s in range(steps):
        bases, ids = hast_amd.synth_reads_host(p, shard_first_read(s, world, rank, R), R)
        e = [np.zeros(n_bc, np.uint32) for _ in range(3)]
        o.ho_classify_ids(oc, bases.ctypes.data, off.ctypes.data, ids.ctypes.data, R, e[0].ctypes.data, e[1].ctypes.data,
                          e[2].ctypes.data, None, 1)
        for c in range(3):
            counts[:, c] += e[c].astype(np.uint64)
    o.ho_free(oc)
    packed = np.ascontiguousarray(counts[:, :3].T)              # hast_counts_pack: c0[n] | c1[n] | neg[n], the reserved word stays home
    t = torch.from_numpy(packed.view(np.int64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)                   # what bench.py does over RCCL
    counts[:, :3] = t.numpy().view(np.uint64).T                # hast_counts_unpack
    if rank == 0:
        np.save(os.path.join(out_dir, "merged.npy"), counts)
    dist.destroy_process_group()


def test_two_rank_shard_merge_equals_single_process(oracle_lib, tmp_path):
    import hast_amd
    from hast_amd.binding import make_params
    from hast_amd.sharding import job_reads
    hast_amd.build()
    steps, R, world = 3, 700, 2
    mp.spawn(_worker, args=(world, _free_port(), steps, R, str(tmp_path)), nprocs=world, join=True)
    merged = np.load(tmp_path / "merged.npy")
    # single process over the union of the shards = reads [0, steps*world*R)
    k, L, n_keys, n_bc = 21, 150, 5000, 211
    p = make_params(k, L, n_keys, n_bc)
    n = job_reads(steps, world, R)
    oc = oracle_lib.ho_new()
    for h in (0, 1):
        keys = hast_amd.synth_keys_host(p, h, 0, n_keys)
        oracle_lib.ho_load_keys(oc, keys.ctypes.data, keys.size, h, k)
    bases, ids = hast_amd.synth_reads_host(p, 0, n)
    off = np.arange(n + 1, dtype=np.uint64) * L
    e = [np.zeros(n_bc, np.uint32) for _ in range(3)]
    oracle_lib.ho_classify_ids(oc, bases.ctypes.data, off.ctypes.data, ids.ctypes.data, n, e[0].ctypes.data, e[1].ctypes.data,
                               e[2].ctypes.data, None, 2)
    oracle_lib.ho_free(oc)
    hot = np.zeros(n_bc, np.uint64)
    hot[7] = sum((1 << 32) + 5 + r for r in range(world))          # what the ranks held before the run, summed without a wrap
    for c in range(3):
        assert np.array_equal(merged[:, c], e[c].astype(np.uint64) + hot)
    assert merged[7, 0] > (1 << 33)
    assert not merged[:, 3].any()

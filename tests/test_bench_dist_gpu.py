"""bench.py's N>1 control flow on a 1-GPU box (-m gpu): two ranks share device 0 and reduce over gloo
(HAST_BENCH_SHARE_GPU / HAST_BENCH_BACKEND test switches); the merged per-barcode totals must equal ONE process
classifying the same read batches, and stdout must hold exactly one JSON line.  (On the 8-GPU node the same code
runs with one rank per GPU over RCCL.)"""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(cmd, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = r.stdout.decode().strip().splitlines()
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_two_ranks_equal_one_process():
    common = ["--workload", "c1", "--batch-reads", "300000", "--cpu-seconds", "0"]
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"] + common,
               {"HAST_BENCH_SHARE_GPU": "1", "HAST_BENCH_BACKEND": "gloo"})
    # ranks 0,1 at timed steps j=1..3 own batches (2j + r) = 2..7; one process with warmup 2 owns j = 2..7
    one = _run([sys.executable, "bench.py", "--steps", "6", "--warmup", "2"] + common, {})
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["reads_total"] == one["config"]["reads_total"] == 6 * 300000
    assert two["hits"] == one["hits"] and two["hits"]["c0"] > 0
    assert two["scaling"] == "weak" and "all_reduce" in two["config"]["collective"]


def _n_gpus():
    import ctypes
    try:
        n = ctypes.c_int(0)
        return n.value if ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs: one rank per GPU, all-reduce over RCCL/xGMI")
def test_two_ranks_over_rccl_equal_one_process():
    """the driver's N=2 launch, as is: one rank per GPU, nccl(=RCCL) backend, ONE all_reduce(sum,u32) of the counters"""
    common = ["--workload", "c1", "--batch-reads", "300000", "--cpu-seconds", "0"]
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29534", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"] + common, {})
    one = _run([sys.executable, "bench.py", "--steps", "6", "--warmup", "2"] + common, {})
    assert two["n_gpus"] == 2 and two["hits"] == one["hits"] and two["hits"]["c0"] > 0

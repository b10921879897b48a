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
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = r.stdout.decode().strip().splitlines()
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def _n_gpus():
    import ctypes
    try:
        n = ctypes.c_int(0)
        return n.value if ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0



# --batches-per-step 1: a step is ONE launch over batch j (the default step walks every resident batch), so that the read ranges of
# N ranks can be laid next to one process's
COMMON = ["--workload", "c1", "--batch-reads", "300000", "--cpu-seconds", "0", "--batches-per-step", "1"]
# BASELINE config 5's shape (per-read mode, K = 31, long reads cut into segments on the device), small
C5 = ["--workload", "c5", "--keys-per-hap", "1000000", "--read-len", "6000", "--batch-reads", "3000", "--cpu-seconds", "0", "--batches-per-step", "1"]


@pytest.fixture(scope="module")
def one_process():
    """one plain process over read batches j = 2..7"""
    return _run([sys.executable, "bench.py", "--steps", "6", "--warmup", "2"] + COMMON, {})


def test_one_rank_over_rccl_self_launched(one_process):
    """The N>1 code path of bench.py on the nccl (= RCCL) backend with ONE rank, started the way a user would start N ranks:
    `python bench.py --gpus 1` with HAST_BENCH_FORCE_DIST=1 and no WORLD_SIZE launches torch.distributed.run as a child itself;
    the rank runs RCCL init, barrier(device_ids), the all_reduce of the int64 view of the (u64) counters and the MAX of the elapsed
    times.  The totals must equal the plain run's and the line must say where the time went."""
    env = {"HAST_BENCH_FORCE_DIST": "1"}
    r = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2"] + COMMON, env)
    assert r["n_gpus"] == 1 and r["hits"] == one_process["hits"] and r["hits"]["c0"] > 0
    assert "all_reduce" in r["config"]["collective"] and "nccl" in r["ranks"]["backend"]
    assert r["allreduce_ms"] > 0 and r["ranks"]["allreduce_ms"]["per_rank"] == [pytest.approx(r["allreduce_ms"], abs=1e-3)]
    assert r["ranks"]["kernel_ms_avg"]["min"] > 0 and len(r["ranks"]["kernel_ms_avg"]["per_rank"]) == 1
    assert one_process["allreduce_ms"] == 0.0 and one_process["ranks"]["backend"] is None


@pytest.mark.skipif(_n_gpus() >= 2, reason="a box with two GPUs runs this launch for real (test_two_ranks_over_rccl_equal_one_process)")
def test_more_ranks_than_gpus_fails_in_the_child_with_a_clear_message():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"] + COMMON, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode != 0 and not r.stdout.strip()
    assert b"has no GPU of its own" in r.stderr and b"launching 2 ranks" in r.stderr


@pytest.mark.parametrize("ranks", [2, 4])
def test_ranks_sharing_the_gpu_equal_one_process(one_process, ranks):
    """N ranks on the one GPU of the box (HAST_BENCH_SHARE_GPU) reducing over gloo: the read ranges of the ranks, the merge and
    the rank-0 output for N = 2 and N = 4 (the box allows six processes on its GPU)."""
    common = COMMON
    n = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
              "--master-port", str(29533 + ranks), "bench.py", "--gpus", str(ranks), "--steps", "3", "--warmup", "1"] + common,
             {"HAST_BENCH_SHARE_GPU": "1", "HAST_BENCH_BACKEND": "gloo"})
    # ranks r = 0..N-1 at timed steps j = 1..3 own batches (N j + r) = N .. 4N-1; one process with warmup N owns the same range
    one = one_process if ranks == 2 else _run([sys.executable, "bench.py", "--steps", str(3 * ranks), "--warmup", str(ranks)] + common, {})
    assert n["n_gpus"] == ranks and one["n_gpus"] == 1
    assert len(n["ranks"]["kernel_ms_avg"]["per_rank"]) == ranks and n["allreduce_ms"] >= 0
    assert n["config"]["reads_total"] == one["config"]["reads_total"] == 3 * ranks * 300000
    assert n["hits"] == one["hits"] and n["hits"]["c0"] > 0
    assert n["scaling"] == "weak" and "all_reduce" in n["config"]["collective"]


def test_config5_two_ranks_sharing_the_gpu_equal_one_process():
    """Per-read mode (BASELINE config 5) under N ranks: no reduction of the rows, but the line carries the hit totals of ALL
    ranks over the whole timed region (two scalars per rank, summed outside the timed region), so N ranks can be checked against
    one process over the same read ranges."""
    one = _run([sys.executable, "bench.py", "--steps", "6", "--warmup", "2"] + C5, {})
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29541", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"] + C5,
               {"HAST_BENCH_SHARE_GPU": "1", "HAST_BENCH_BACKEND": "gloo"})
    assert one["mode"].startswith("per-read") and two["mode"].startswith("per-read") and two["n_gpus"] == 2
    assert one["config"]["reads_total"] == two["config"]["reads_total"] == 6 * 3000
    assert one["hits"]["c0"] > 0 and one["hits"]["c1"] > 0
    assert (two["hits"]["c0"], two["hits"]["c1"]) == (one["hits"]["c0"], one["hits"]["c1"])


def test_default_step_walks_every_resident_batch():
    """Without --batches-per-step a step is one pass over all resident batches (long enough for the driver's busy sampler at the
    driver's fixed --steps): reads_total says so and the totals are steps x one pass."""
    base = ["--workload", "c1", "--batch-reads", "200000", "--cpu-seconds", "0", "--max-resident-gb", "0.1"]     # 3 resident batches
    a = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1"] + base, {})
    b = _run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0"] + base, {})
    assert a["config"]["resident_batches"] == a["config"]["batches_per_step"] == 3
    assert a["config"]["reads_total"] == 2 * 3 * 200000 and b["config"]["reads_total"] == 3 * 200000
    assert a["hits"]["c0"] == 2 * b["hits"]["c0"] and a["hits"]["neg_reads"] == 2 * b["hits"]["neg_reads"] and b["hits"]["c0"] > 0


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs: one rank per GPU, all-reduce over RCCL/xGMI")
def test_two_ranks_over_rccl_equal_one_process(one_process):
    """the driver's N=2 launch, as is: one rank per GPU, nccl(=RCCL) backend, ONE all_reduce(sum,u64) of the counters -- and the
    same through bench.py's own launcher (`python bench.py --gpus 2`)"""
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29534", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"] + COMMON, {})
    assert two["n_gpus"] == 2 and two["hits"] == one_process["hits"] and two["hits"]["c0"] > 0
    own = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"] + COMMON, {})
    assert own["n_gpus"] == 2 and own["hits"] == one_process["hits"]

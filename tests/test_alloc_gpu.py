"""Allocation paths on the GPU (-m gpu): memory that closed streams have PARKED (hast_internal.h: a free stops every stream of the device,
so streams keep their buffers until hast_release_parked) must never make a later allocation fail or quietly take a slower path, and a
context that does fall back to the table-only kernel has to say so."""
import os
import subprocess
import sys

import pytest

import hast_amd
from tests.conftest import ROOT, load_case

pytestmark = pytest.mark.gpu

_PARKED_THEN_FILTER = r'''
import ctypes as C, sys
import numpy as np
import hast_amd
from hast_amd.binding import make_params
lib = hast_amd.lib()
k, n_keys = 21, 10_000_000                      # 20M keys: from 16M keys on the filter takes 4^14 blocks = 34 GB (exact entries)
with hast_amd.Context(k) as ctx:
    free0, total, parked = C.c_size_t(), C.c_size_t(), C.c_size_t()
    assert lib.hast_dev_mem_info(ctx._h, C.byref(free0), C.byref(total), C.byref(parked)) == 0
    # streams with large blocks, closed again: their buffers (~3.3 GB per slot of a 256-MB block) are parked, not freed
    n_streams = 0
    while True:
        fq = C.c_void_p()
        assert lib.hast_fq_create_ex(ctx._h, 256 << 20, 16, None, 1, C.byref(fq)) == 0, lib.hast_last_error()
        lib.hast_fq_destroy(fq)
        n_streams += 1
        f = C.c_size_t()
        assert lib.hast_dev_mem_info(ctx._h, C.byref(f), None, C.byref(parked)) == 0
        if free0.value - f.value > 90 << 30 or n_streams >= 4:
            break
    held = free0.value - f.value
    assert held > 60 << 30 and parked.value > 0, (held, parked.value)
    ctx.table_reserve(2 * n_keys)
    ctx.synth_table_build(make_params(k, 150, n_keys, 1))
    # ballast: what is left free is less than the filter needs
    assert lib.hast_dev_mem_info(ctx._h, C.byref(f), None, None) == 0
    ballast = C.c_void_p()
    want = f.value - (12 << 30)
    assert lib.hast_dev_alloc(ctx._h, want, C.byref(ballast)) == 0, lib.hast_last_error()
    assert lib.hast_dev_mem_info(ctx._h, C.byref(f), None, None) == 0
    assert f.value < 30 << 30, f.value
    assert lib.hast_filter_build(ctx._h) == 0, lib.hast_last_error()
    en, m, t, kp, nbytes = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_uint64()
    assert lib.hast_filter_info(ctx._h, C.byref(en), C.byref(m), C.byref(t), C.byref(kp), C.byref(nbytes)) == 0
    sw = C.create_string_buffer(512)
    lib.hast_ctx_options(ctx._h, sw, 512)
    print("filter", en.value, m.value, nbytes.value, "held_before", held, "switches", sw.value.decode() or "none")
    assert lib.hast_dev_mem_info(ctx._h, None, None, C.byref(parked)) == 0
    print("parked_after", parked.value)
    lib.hast_dev_free(ctx._h, ballast)
'''


def test_parked_memory_is_given_back_before_the_filter_falls_back():
    """90 GB of closed streams' buffers parked, the rest of the device taken: the 34-GB filter of a 20M-key table must still be built
    (exact entries), because the allocation frees what is parked and asks again -- until round 5 the context silently went to the
    round-1 kernel"""
    env = dict(os.environ, HAST_PARK_GB="1000", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", _PARKED_THEN_FILTER], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = r.stdout.decode()
    line = [l for l in out.splitlines() if l.startswith("filter ")][0].split()
    assert line[1] == "2" and line[2] == "14" and int(line[3]) == 128 << 28, out          # exact entries, 4^14 blocks of 128 B
    assert "switches none" in out, out
    assert [l for l in out.splitlines() if l.startswith("parked_after")][0].split()[1] == "0", out


def test_a_context_that_falls_back_to_the_table_only_kernel_says_so(golden_workdir):
    """HAST_TEST_FILTER_OOM=1 makes the filter's allocation fail on every context: same stdout as the real reference binary (the table
    alone decides hits), and the run says what happened -- a WARN line per context, the switch line and the filter line of --stats"""
    exe = hast_amd.classify_exe()
    meta = load_case("rand_k21")["runs"]["pair_w104"]
    d = golden_workdir / "rand_k21"
    # (blocks of 50 records: both contexts get blocks to classify -- a context that never classifies never asks for its filter)
    r = subprocess.run([exe] + meta["argv"] + ["--devices", "0,0", "--stats", "--batch-reads", "50"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                       env=dict(os.environ, HAST_TEST_FILTER_OOM="1"))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout == open(d / meta["expected"], "rb").read()
    err = r.stderr.decode()
    assert len([l for l in err.splitlines() if l.startswith(" WARN : GPU 0 (context") and "filter_fallback_table_only_no_room_for_bytes" in l]) == 2, err[-1500:]
    assert "__stats_filter__ mode=off_table_only" in err
    assert [l for l in err.splitlines() if l.startswith("__stats_switches__")][0].startswith("__stats_switches__ filter=0 filter_fallback_table_only_no_room_for_bytes="), err[-1500:]
    # without the switch: no WARN, exact entries or prints
    r = subprocess.run([exe] + meta["argv"] + ["--devices", "0,0", "--stats"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b"WARN : GPU" not in r.stderr and b"mode=off_table_only" not in r.stderr

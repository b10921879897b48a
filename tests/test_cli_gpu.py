"""The drop-in boundary on the GPU (-m gpu): the `classify` executable (HIP path) must print exactly
what the real reference binary printed for the committed golden inputs (tests/golden/, made by
gen_golden.py from oracle/_ref/classify)."""
import os
import subprocess

import pytest

import hast_amd
from tests.conftest import golden_cases, load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(hast_amd.classify_exe()):
        hast_amd.build()
    return hast_amd.classify_exe()


@pytest.mark.parametrize("case,run", [c for c in golden_cases() if c[0] != "rand_k32"])
def test_cli_matches_reference_golden(exe, golden_workdir, case, run):
    meta = load_case(case)["runs"][run]
    d = golden_workdir / case
    res = subprocess.run([exe] + meta["argv"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert res.stdout == open(d / meta["expected"], "rb").read()
    mine = [l for l in res.stderr.decode().splitlines() if l.startswith("Recorded") or "erase a adaptor" in l]
    assert mine == meta["ref_log"]


def test_cli_small_batches_and_counter_growth(exe, golden_workdir):
    """Tiny batches force many launches and the counter-array regrowth path; output unchanged."""
    meta = load_case("rand_k21")["runs"]["pair_w104"]
    d = golden_workdir / "rand_k21"
    res = subprocess.run([exe] + meta["argv"] + ["--batch-reads", "257", "--stats"], cwd=d,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert res.stdout == open(d / meta["expected"], "rb").read()
    assert "__stats__" in res.stderr.decode()


def test_cli_usage_and_errors(exe, golden_workdir, tmp_path):
    d = golden_workdir / "edge_k7"
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 255 and b"Uasge" in r.stderr and r.stdout == b""          # classify.cpp:425-428
    r = subprocess.run([exe, "--hap0", "hap0.mer", "--hap1", "hap1.mer", "--read", "r1.fq", "-t", "0"], cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 255
    r = subprocess.run([exe, "--hap0", "missing.mer", "--hap1", "hap1.mer", "--read", "r1.fq"], cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 2 and r.stdout == b""       # the reference would hang here (classify.cpp:41)
    short = tmp_path / "short.fq"
    short.write_text("@x#1_1_1/1\nACG\n+\nFFF\n")
    r = subprocess.run([exe, "--hap0", "hap0.mer", "--hap1", "hap1.mer", "--read", str(short)], cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 3                           # the reference aborts (kmer.h:171)

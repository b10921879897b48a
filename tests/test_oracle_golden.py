"""Pins the oracle (CPU restatement, oracle/hast_oracle.c) against
  (i)  the reference's own known-answer vectors: TestAll(), classify.cpp:341-367;
  (ii) golden outputs of the real reference binary (tests/golden/, made by gen_golden.py).
CPU only."""
import ctypes as C
import os
import subprocess

import pytest

from tests import oracle_binding as ob
from tests.conftest import golden_cases, load_case


# ---- (i) the reference's KATs, classify.cpp:341-367 ---------------------------------------
def test_kat_parse_name(oracle_lib):
    assert ob.parse_name(oracle_lib, b"VSDSDS#XXX_xxx_s/1") == b"XXX_xxx_s"          # :342


def test_kat_base_codes(oracle_lib):
    assert [oracle_lib.ho_base2int(c) for c in b"AGCTC"] == [0, 3, 1, 2, 1]          # :344-346
    assert [oracle_lib.ho_base2int(c) for c in b"GAGCT"] == [3, 0, 3, 1, 2]          # :347-349


def test_kat_canonical(oracle_lib):
    assert oracle_lib.ho_canon_str(b"AGCTC", 5) == 0xD9                              # :351-352
    assert oracle_lib.ho_canon_str(b"GAGCT", 5) == 0xD9                              # :353-354


def test_kat_chop_and_roundtrip(oracle_lib):
    km = ob.chop(oracle_lib, b"GAGCTA", 5)                                            # :355-362
    assert km == [0xD9, 0xD8]
    buf = C.create_string_buffer(8)
    oracle_lib.ho_kmer_to_str(km[0], 5, buf)
    assert buf.value == b"AGCTC"                                                      # :363-365
    oracle_lib.ho_kmer_to_str(km[1], 5, buf)
    assert buf.value == b"AGCTA"                                                      # :366


# ---- extra self-consistency (rolling == recompute, K up to 32) -----------------------------
@pytest.mark.parametrize("k", [1, 5, 11, 21, 31, 32])
def test_rolling_equals_recompute(oracle_lib, k):
    import random
    rng = random.Random(k)
    seq = "".join(rng.choice("ACGTacgtNnRY") for _ in range(200)).encode()
    km = ob.chop(oracle_lib, seq, k)
    assert len(km) == len(seq) - k + 1
    for i, v in enumerate(km):
        assert v == oracle_lib.ho_canon_str(seq[i:i + k], k)
    assert ob.chop(oracle_lib, b"ACG", 5) == []


def test_parse_name_edges(oracle_lib):
    p = lambda h: ob.parse_name(oracle_lib, h)
    assert p(b"@V3#2_2_2/1\tx/y\t1") == b"2_2_2/1\tx"      # last '#', last '/'
    assert p(b"@noBarcode/1") == b"@noBarcode"             # no '#': from 0
    assert p(b"@V13#9_9_9") == b"9_9_9"                    # no '/': to end
    assert p(b"@a/b#10_10_10") == b"10_10_10"              # '/' before '#': to end
    assert p(b"") == b""


def test_get_hap(oracle_lib):
    g = lambda bc, c0, c1, n0=100, n1=100, w0=1.0, w1=1.0: oracle_lib.ho_get_hap(bc, len(bc), c0, c1, n0, n1, w0, w1)
    assert g(b"0_0_0", 9, 1) == -1 and g(b"0_0", 9, 1) == -1 and g(b"0", 1, 9) == -1
    assert g(b"1_2_3", 5, 5) == -1 and g(b"1_2_3", 5, 5, w0=1.04) == 0
    assert g(b"1_2_3", 5, 5, n0=101) == 1
    assert g(b"1_2_3", 3, 0) == 0 and g(b"1_2_3", 0, 2) == 1 and g(b"1_2_3", 0, 0) == -1


# ---- (ii) golden outputs of the real reference binary --------------------------------------
@pytest.mark.parametrize("case,run", golden_cases("s01") + golden_cases("s03"))
def test_oracle_matches_reference_golden(oracle_dir, golden_workdir, case, run):
    meta = load_case(case)["runs"][run]
    d = golden_workdir / case
    s03 = meta.get("program") == "s03"
    res = subprocess.run([os.path.join(oracle_dir, "oracle_classify_s03" if s03 else "oracle_classify")] + meta["argv"], cwd=d,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert res.returncode == 0, res.stderr.decode()[-1000:]
    expected = open(d / meta["expected"], "rb").read()
    assert res.stdout == expected
    if s03:
        return
    # set-size / adaptor-erase log lines agree with the reference's (classify.cpp:45,321-336)
    mine = [l for l in res.stderr.decode().splitlines() if l.startswith("Recorded") or "erase a adaptor" in l]
    assert mine == meta["ref_log"]


def test_golden_reference_binary_still_agrees(golden_workdir):
    """When the real reference binary is present (build container), re-run it: the committed
    expected files must be exactly what it prints today."""
    from tests.conftest import ROOT
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "classify")):
        pytest.skip("oracle/_ref/classify not built here")
    for case, run in golden_cases("s01") + golden_cases("s03"):
        meta = load_case(case)["runs"][run]
        d = golden_workdir / case
        ref = os.path.join(ROOT, "oracle", "_ref", "classify_s03" if meta.get("program") == "s03" else "classify")
        res = subprocess.run([ref] + meta["argv"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert res.returncode == 0
        assert res.stdout == open(d / meta["expected"], "rb").read(), (case, run)


# ---- stage 00: the reference script (+ its vendored jellyfish) run on our inputs ---------------------------
@pytest.mark.parametrize("case,run", golden_cases("s00"))
def test_s00_oracle_matches_reference_script_golden(oracle_dir, golden_workdir, tmp_path, case, run):
    from tests.conftest import run_s00_case
    run_s00_case(os.path.join(oracle_dir, "oracle_unshared"), golden_workdir, tmp_path, case, run)


def test_s00_known_answers(oracle_lib):
    """canonical form and counting rules probed on jellyfish 2.3.0 (count -m 5 -C on hand-made records)"""
    o = oracle_lib
    assert o.ho_s00_canon_str(b"TACGT", 5) == o.ho_s00_canon_str(b"ACGTA", 5)        # a k-mer and its reverse complement
    assert o.ho_s00_canon_str(b"acgta", 5) == o.ho_s00_canon_str(b"ACGTA", 5)        # case-insensitive
    c = o.ho_s00_new(5)
    for rec in (b"ACGTACGTAC", b"acgtNACGTAcgtt", b"AC", b"TTTTTRAAAAACCCC"):
        o.ho_s00_add_seq(c, 0, rec, len(rec))
    exp = {b"AAAAA": 2, b"AAAAC": 1, b"AAACC": 1, b"AACCC": 1, b"AACGT": 1, b"ACCCC": 1, b"ACGTA": 5, b"CGTAC": 5}
    for kmer, n in exp.items():
        assert o.ho_s00_count(c, 0, o.ho_s00_canon_str(kmer, 5)) == n
    assert o.ho_s00_distinct(c, 0) == len(exp) and o.ho_s00_distinct(c, 1) == 0
    o.ho_s00_free(c)

"""CPU-only checks of the product library: it loads, exports every symbol include/hast.h declares,
its host-side pieces (no device work) agree with the oracle, and it refuses to run without a GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

import hast_amd
from hast_amd.binding import make_params
from tests import oracle_binding as ob
from tests.conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    hast_amd.build()
    return hast_amd.lib()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "hast.h")).read()
    declared = set(re.findall(r"\b(hast_[a-z0-9_]+)\s*\(", hdr)) - {"hast_status"}
    assert declared == set(hast_amd.ABI_SYMBOLS), declared ^ set(hast_amd.ABI_SYMBOLS)
    out = subprocess.run(["nm", "-D", "--defined-only", hast_amd.lib_path()], stdout=subprocess.PIPE, check=True).stdout.decode()
    exported = set(re.findall(r" T (hast_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hast_amd.HastError) as ei:
        hast_amd.Context(21)
    assert ei.value.status == 2 and "no CPU path" in str(ei.value)
    r = subprocess.run([hast_amd.classify_exe(), "--hap0", "hap0.mer", "--hap1", "hap1.mer", "--read", "r1.fq"],
                       cwd=os.path.join(ROOT, "tests", "golden", "edge_k7"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 4 and r.stdout == b""


def test_product_does_not_link_oracle(lib):
    for f in (hast_amd.lib_path(), hast_amd.classify_exe()):
        out = subprocess.run(["ldd", f], stdout=subprocess.PIPE).stdout.decode()
        assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "hast_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(root, fn), errors="ignore").read()
                assert "oracle" not in txt.lower() or fn == "__init__.py", (fn, "product code must not reference oracle/")


def test_host_kmer_primitives_match_oracle(lib, oracle_lib):
    import random
    rng = random.Random(7)
    assert hast_amd.canon_kmer(b"AGCTC") == 0xD9 and hast_amd.canon_kmer(b"GAGCT") == 0xD9     # classify.cpp:351-354
    assert hast_amd.chop_read(b"GAGCTA", 5) == [0xD9, 0xD8]                                     # classify.cpp:355-362
    for k in (1, 2, 5, 11, 16, 21, 27, 31):
        seq = "".join(rng.choice("ACGTacgtNnRYK-") for _ in range(300)).encode()
        assert hast_amd.chop_read(seq, k) == ob.chop(oracle_lib, seq, k)
        for i in range(0, 250, 17):
            assert hast_amd.canon_kmer(seq[i:i + k]) == oracle_lib.ho_canon_str(seq[i:i + k], k)


def test_host_parse_barcode_and_get_hap_match_oracle(lib, oracle_lib):
    import random
    rng = random.Random(3)
    heads = [b"VSDSDS#XXX_xxx_s/1", b"@V3#2_2_2/1\tx/y\t1", b"@noBarcode/1", b"@V13#9_9_9", b"@a/b#10_10_10", b"", b"#", b"/",
             b"#/", b"/#", b"a#b#c/d/e"]
    for _ in range(300):
        heads.append(bytes(rng.choice(b"ab#/_1\t") for _ in range(rng.randint(0, 20))))
    for h in heads:
        assert hast_amd.parse_barcode(h) == ob.parse_name(oracle_lib, h), h
    for bc in (b"0_0_0", b"0_0", b"0", b"1_2_3", b"00", b"0_0_0_0"):
        for _ in range(200):
            c0, c1 = rng.choice([0, 0, 1, 5, 1000, 2 ** 31 - 1]), rng.choice([0, 0, 1, 5, 999, 2 ** 31 - 1])
            n0, n1 = rng.randint(1, 10 ** 9), rng.randint(1, 10 ** 9)
            w0, w1 = rng.choice([1.0, 1.04, 0.5]), rng.choice([1.0, 1.04, 2.5])
            assert hast_amd.get_hap(bc, c0, c1, n0, n1, w0, w1) == oracle_lib.ho_get_hap(bc, len(bc), c0, c1, n0, n1, w0, w1)


def test_synth_host_generator_properties(lib, oracle_lib):
    """The synthetic reads are deterministic, plant real parental k-mers, and some carry N."""
    k, L, n_keys = 21, 150, 4000
    p = make_params(k, L, n_keys, 128)
    a, ids = hast_amd.synth_reads_host(p, 100, 2000)
    b, ids2 = hast_amd.synth_reads_host(p, 100, 2000)
    assert np.array_equal(a, b) and np.array_equal(ids, ids2)
    c, _ = hast_amd.synth_reads_host(p, 1100, 1000)
    assert np.array_equal(a[1000 * L:], c)                        # counter-based: independent of batch split
    assert ids.max() < 128 and set(np.unique(a)) <= set(b"ACGTN")
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    oc = oracle_lib.ho_new()
    for h in (0, 1):
        assert oracle_lib.ho_load_keys(oc, keys[h].ctypes.data, n_keys, h, k) == 0
    e = [np.zeros(128, np.uint32) for _ in range(3)]
    off = np.arange(2001, dtype=np.uint64) * L
    oracle_lib.ho_classify_ids(oc, a.ctypes.data, off.ctypes.data, ids.ctypes.data, 2000, e[0].ctypes.data, e[1].ctypes.data,
                               e[2].ctypes.data, None, 2)
    oracle_lib.ho_free(oc)
    hits = int(e[0].sum() + e[1].sum())
    assert 1500 < hits < 4000                                     # ~1.5 planted k-mers per read
    assert 0 < (a == ord("N")).sum() < 40


def test_find_bounds_matches_oracle_and_the_awk_program(lib, oracle_lib, tmp_path):
    """hast_kc_find_bounds (host arithmetic of find_bounds.awk) against the oracle's restatement on random histograms and, in the
    build container, against the reference's own awk program run on the printed rows"""
    import ctypes as C
    import random
    rng = random.Random(5)
    awk = "/root/reference/00.build_unshare_kmers_by_jellyfish/find_bounds.awk"
    for it in range(60):
        h = np.zeros(hast_amd.KC_HISTO_HIGH + 2, dtype=np.uint64)
        shape = rng.choice(["typical", "sparse", "flat", "empty", "rising", "tail"])
        if shape == "typical":                      # error peak at 1, valley, coverage peak
            peak = rng.randint(8, 60)
            for c in range(1, 4 * peak):
                h[c] = int(1e6 / c ** 3 + 5e4 * np.exp(-((c - peak) ** 2) / (2.0 * peak))) + rng.randint(0, 3)
        elif shape == "sparse":
            for c in rng.sample(range(1, 10002), rng.randint(1, 12)):
                h[c] = rng.randint(1, 1000)
        elif shape == "flat":
            h[1:rng.randint(2, 50)] = 7
        elif shape == "rising":
            for c in range(1, 30):
                h[c] = c * 10
        elif shape == "tail":
            h[1], h[2], h[10001] = 100, 50, rng.randint(1, 500)
        got = hast_amd.kc_find_bounds(h)
        out = [C.c_long() for _ in range(4)]
        oracle_lib.ho_s00_find_bounds(h.ctypes.data_as(C.POINTER(C.c_uint64)), *[C.byref(x) for x in out])
        assert got == tuple(x.value for x in out), (shape, got)
        if os.path.exists(awk):
            rows = "".join("%d %d\n" % (c, h[c]) for c in range(1, 10002) if h[c])
            (tmp_path / "h.histo").write_text(rows)
            r = subprocess.run(["awk", "-f", awk, str(tmp_path / "h.histo")], stdout=subprocess.PIPE, check=True).stdout.decode()
            ref = tuple(int(l.split("=")[1]) for l in r.splitlines())
            assert got == ref, (shape, got, ref)


def test_committed_pmc_profiles_belong_to_these_device_sources(monkeypatch):
    """bench.py prices the HBM traffic of k_classify_f from profiles/pmc_traffic*.json and refuses a profile taken on other
    device code (roofline.frac would then be null in the driver's bench line): the committed profiles must carry the hash of
    the device sources in this tree, for every workload bench.py can be asked for.  Runs under measurement switches have no
    profile of their own and get no fraction."""
    import json
    import types
    import bench
    for v in bench.MEASUREMENT_SWITCHES:
        monkeypatch.delenv(v, raising=False)
    sid = bench.kernel_source_id()
    for name, wl, clustered in (("pmc_traffic.json", "c3", False), ("pmc_traffic_clustered.json", "c3", True),
                                ("pmc_traffic_workloadc2.json", "c2", False), ("pmc_traffic_workloadc5.json", "c5", False)):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert d["kernel_source_id"] == sid, "%s was taken on other device code: run tools/gpu/collect_all.sh + profiles/summarize.py" % name
        assert d["workload"] == wl
        k, L = (31, 20000) if wl == "c5" else (21, 150)
        args = types.SimpleNamespace(workload=wl, clustered=clustered, no_plants=False, k=k, read_len=L)
        R = d["batch_reads"]
        ms = d["hbm_read_requests_per_launch"] / 47e9 * 1e3           # a launch at 47 G requests/s, about what the part serves
        rf = bench.roofline(args, R, 1.0, [ms, ms * 1.02], [2.0, 2.0], [ms + 2, ms + 3], None, (15, 23) if wl == "c5" else (14, 21))
        assert rf["frac"] is not None and 0.3 < rf["frac"] < 1.0 and rf["request_rate"]["frac"] <= 1.0, (name, rf)
        # the efficiency figures next to the utilisation: bytes and requests per read against the useful-bytes floor and the
        # sampling scheme's lower bound (K = 21: 1190 B and 17.7 requests per 150-bp read)
        us = rf["useful"]
        assert us["floor_bytes_per_read"] == L + (L - k + 1) * 8 and 0 < us["frac_of_floor"] < 1 and 0 < us["frac_of_scheme_bound"] < 1, us
        if wl != "c5":
            assert us["scheme_bound_requests_per_read"] == pytest.approx(130 * 3 / 22)
    monkeypatch.setenv("HAST_FILTER_EXACT", "0")
    rf = bench.roofline(types.SimpleNamespace(workload="c3", clustered=False, no_plants=False, k=21, read_len=150), 48_000_000, 1.0, [25.0], [2.0], [27.0])
    assert rf["frac"] is None and "HAST_FILTER_EXACT" in rf["traffic_source"]

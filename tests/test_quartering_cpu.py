"""CPU test of hast_amd/quartering_fastq: byte-identical outputs to the reference's awk program
(01.classify_stlfr_reads/quartering_fastq.awk under mawk, fixtures made by tests/golden/gen_golden.py)."""
import gzip
import hashlib
import json
import os
import shutil
import subprocess

import pytest

import hast_amd
from tests.conftest import GOLDEN, ROOT

EXE = os.environ.get("HAST_QUARTERING_EXE") or os.path.join(ROOT, "hast_amd", "quartering_fastq")


@pytest.fixture(scope="module")
def exe():
    hast_amd.build()
    return EXE


def test_edge_case_all_branches(exe, tmp_path):
    exp = json.load(open(os.path.join(GOLDEN, "quartering", "expected.json")))["edge"]
    for fn, txt in exp["inputs"].items():
        (tmp_path / fn).write_text(txt)
    for threads in (1, 5):
        for fn in exp["outputs"]:
            (tmp_path / fn).unlink(missing_ok=True)
        r = subprocess.run([exe, "-t", str(threads), "--prefix", "e.fq", "p.bc", "m.bc", "h.bc", "e.fq"], cwd=tmp_path,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and r.stdout == b""
        assert r.stderr.decode() == exp["stderr"]
        got = {fn: (tmp_path / fn).read_text() for fn in exp["outputs"]}
        assert got == exp["outputs"]
        assert sorted(os.listdir(tmp_path)) == sorted(list(exp["inputs"]) + list(exp["outputs"]))


def _bgzf(data, tail=b""):
    """blocked gzip as bgzip writes it (members of <= 64 KB with a BC extra field + the empty end-of-file member),
    optionally followed by an ordinary gzip member"""
    import struct
    import zlib
    def block(d):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = c.compress(d) + c.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(raw) + 25) + raw +
                struct.pack("<II", zlib.crc32(d) & 0xFFFFFFFF, len(d)))
    out = b"".join(block(data[i:i + 65280]) for i in range(0, len(data), 65280))
    if tail:
        c = zlib.compressobj(6, zlib.DEFLATED, 31)
        return out + c.compress(tail) + c.flush()
    return out + block(b"")


@pytest.mark.parametrize("threads,block_mb,via", [(1, 64, "file"), (8, 1, "file"), (3, 1, "gz"), (4, 64, "stdin"), (2, 1, "bgzf"), (5, 64, "bgzf+gz"),
                                                  (2, 1, "gz-zlib")])
def test_rand_k21_matches_reference_awk(exe, tmp_path, threads, block_mb, via):
    exp = json.load(open(os.path.join(GOLDEN, "quartering", "expected.json")))
    for name in ("paternal", "maternal", "homozygous"):
        shutil.copy(os.path.join(GOLDEN, "quartering", name + ".unique.barcodes"), tmp_path)
    lists = ["paternal.unique.barcodes", "maternal.unique.barcodes", "homozygous.unique.barcodes"]
    for fq in ("r1.fq", "r2.fq"):
        data = gzip.open(os.path.join(GOLDEN, "rand_k21", fq + ".gz")).read()
        if fq == "r2.fq":
            data = data[:-1] + b"\n" + exp["r2_tail"].encode()
        cmd = [exe, "-t", str(threads), "--block-mb", str(block_mb), "--prefix", fq] + lists
        if via in ("gz", "gz-zlib"):
            with gzip.open(tmp_path / (fq + ".gz"), "wb") as f:
                f.write(data)
            env = dict(os.environ, HAST_INFLATE="zlib") if via == "gz-zlib" else None
            r = subprocess.run(cmd + [fq + ".gz"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        elif via.startswith("bgzf"):                 # blocked gzip: inflated by several threads; "+gz": an ordinary member appended
            cut = len(data) * 2 // 3 if via == "bgzf+gz" else len(data)
            (tmp_path / (fq + ".gz")).write_bytes(_bgzf(data[:cut], data[cut:]))
            r = subprocess.run(cmd + [fq + ".gz"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env=dict(os.environ, HAST_BGZF_THREADS="3"))
        elif via == "stdin":
            r = subprocess.run(cmd + ["-"], cwd=tmp_path, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        else:
            (tmp_path / fq).write_bytes(data)
            r = subprocess.run(cmd + [fq], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-500:]
        e = exp["files"][fq]
        assert hashlib.md5(r.stderr).hexdigest() == e["stderr_md5"]
        for cls in ("paternal", "maternal", "homozygous", "nobarcode"):
            p = tmp_path / ("%s.%s.fastq" % (fq, cls))
            if cls in e:
                b = p.read_bytes()
                assert (len(b), hashlib.md5(b).hexdigest()) == (e[cls]["bytes"], e[cls]["md5"]), (fq, cls)
            else:
                assert not p.exists()
    log = (tmp_path / "filter_reads.log").read_text()
    if via == "file":
        assert log == exp["filter_reads_log"]
    else:   # the FILENAME line is whatever was given (awk prints "-" behind gzip -dc); the counters are the same
        strip = lambda t: [l for l in t.splitlines() if l.startswith("#")]
        assert strip(log) == strip(exp["filter_reads_log"])


def test_merge_counts_equals_single_run(oracle_dir, golden_workdir, tmp_path):
    """8(f)#3: merging the TSVs of two shards (r1 alone, r2 alone) reproduces the reference's output for both files
    together (rand_k21 golden, weight0 1.04).  Shard TSVs come from the oracle (CPU); set sizes from its log."""
    hast_amd.build()
    d = golden_workdir / "rand_k21"
    base = ["--hap0", "hap0.mer", "--hap1", "hap1.mer", "--weight0", "1.04"]
    sizes = None
    for i, fq in enumerate(("r1.fq.gz", "r2.fq.gz")):
        r = subprocess.run([os.path.join(oracle_dir, "oracle_classify")] + base + ["--read", fq], cwd=d, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, check=True)
        (tmp_path / ("shard%d.tsv" % i)).write_bytes(r.stdout)
    # |S_0|, |S_1| after adaptor scrub: unique canonical keys, from the oracle library
    from tests import oracle_binding as ob
    o = ob.load(os.path.join(oracle_dir, "liboracle.so"))
    oc = o.ho_new()
    for h in (0, 1):
        assert o.ho_load_kmers_file(oc, str(d / ("hap%d.mer" % h)).encode(), h) == 0
    o.ho_init_adaptor(oc, b"CTGTCTCTTATACACATCTTAGGAAGACAAGCACTGACGACATGA", b"TCTGCTGAGTCGAGAACGTCTCTGTGAGCCAAGGAGTTGCTCTGG", None)
    sizes = (o.ho_set_size(oc, 0), o.ho_set_size(oc, 1))
    o.ho_free(oc)
    r = subprocess.run([os.path.join(ROOT, "hast_amd", "merge_counts"), "--set0", str(sizes[0]), "--set1", str(sizes[1]),
                        "--weight0", "1.04", "shard0.tsv", "shard1.tsv"], cwd=tmp_path, stdout=subprocess.PIPE, check=True)
    assert r.stdout == open(d / "expected.pair_w104.tsv", "rb").read()


def test_block_reads_by_several_readers_equal_a_single_reader(exe, tmp_path):
    """blocks >= 32 MB of a regular file are filled by four pread() workers; smaller ones by one fread: same outputs, also
    when the file ends inside a share and when a block ends inside a record"""
    import numpy as np
    rng = np.random.default_rng(3)
    n = 330_000
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 100))]
    ids = rng.integers(0, 40, size=n)
    recs = []
    for i in range(n):
        recs.append(b"@r%d#%d_%d_%d/1\n" % (i, ids[i], ids[i], ids[i] % 3 + 1) + seqs[i].tobytes() + b"\n+\n" + b"F" * 100 + b"\n")
    (tmp_path / "big.fq").write_bytes(b"".join(recs))                      # ~73 MB
    (tmp_path / "p.bc").write_text("".join("%d_%d_%d\n" % (i, i, i % 3 + 1) for i in range(0, 13)))
    (tmp_path / "m.bc").write_text("".join("%d_%d_%d\n" % (i, i, i % 3 + 1) for i in range(13, 27)))
    (tmp_path / "h.bc").write_text("".join("%d_%d_%d\n" % (i, i, i % 3 + 1) for i in range(27, 33)))
    sums = []
    for block_mb in (64, 40, 8):
        r = subprocess.run([exe, "-t", "4", "--block-mb", str(block_mb), "--prefix", "b%d" % block_mb, "p.bc", "m.bc", "h.bc", "big.fq"],
                           cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr[-300:]
        outs = [tmp_path / ("b%d.%s.fastq" % (block_mb, c)) for c in ("paternal", "maternal", "homozygous", "nobarcode")]
        sums.append([hashlib.md5(o.read_bytes()).hexdigest() if o.exists() else None for o in outs] + [hashlib.md5(r.stderr).hexdigest()])
        assert sums[-1][0] is not None and outs[0].stat().st_size > 10_000_000
    assert sums[0] == sums[1] == sums[2]


def test_awk_model_of_the_gpu_tests_equals_the_awk_program():
    """tests/test_fq_gpu.py checks the device router against a ten-line Python restatement of quartering_fastq.awk:21-49 (_awk_route):
    that restatement against the REAL awk program's outputs on the hand-made edge case (every branch) and on the rand_k21 files"""
    import re
    from tests.test_fq_gpu import _awk_route
    exp = json.load(open(os.path.join(GOLDEN, "quartering", "expected.json")))
    e = exp["edge"]
    cls_of = {}
    for name, c in (("p.bc", 1), ("m.bc", 2), ("h.bc", 3)):
        for line in e["inputs"][name].encode().splitlines():
            cls_of.setdefault(re.split(rb"[#/]", line)[0], c)
    out, dropped = _awk_route(e["inputs"]["e.fq"].encode(), cls_of)
    names = {0: "e.fq.nobarcode.fastq", 1: "e.fq.paternal.fastq", 2: "e.fq.maternal.fastq", 3: "e.fq.homozygous.fastq"}
    for c in range(4):
        assert out[c].decode() == e["outputs"].get(names[c], ""), names[c]
    assert "".join("ERROR : unclassify barcode : %s\n" % d.decode() for d in dropped) == e["stderr"]
    cls_of = {}
    for name, c in (("paternal", 1), ("maternal", 2), ("homozygous", 3)):
        for line in open(os.path.join(GOLDEN, "quartering", name + ".unique.barcodes"), "rb").read().splitlines():
            cls_of.setdefault(re.split(rb"[#/]", line)[0], c)
    for fq in ("r1.fq", "r2.fq"):
        data = gzip.open(os.path.join(GOLDEN, "rand_k21", fq + ".gz")).read()
        if fq == "r2.fq":
            data = data[:-1] + b"\n" + exp["r2_tail"].encode()
        out, dropped = _awk_route(data, cls_of)
        want = exp["files"][fq]
        for c, cls in enumerate(("nobarcode", "paternal", "maternal", "homozygous")):
            if cls in want:
                assert (len(out[c]), hashlib.md5(out[c]).hexdigest()) == (want[cls]["bytes"], want[cls]["md5"]), (fq, cls)
            else:
                assert out[c] == b""
        err = "".join("ERROR : unclassify barcode : %s\n" % d.decode() for d in dropped).encode()
        assert hashlib.md5(err).hexdigest() == want["stderr_md5"]

"""Host-only test of the stage-00 record reader (hast_amd/csrc/seqstream.h): the base stream it makes of a FASTA/FASTQ file
must hold exactly the k-mers the pinned restatement of the reference's reader finds in that file."""
import ctypes as C
import glob
import gzip
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = tmp_path_factory.mktemp("seqstream") / "test_seqstream"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-o", str(exe),
                    os.path.join(ROOT, "tests", "native", "test_seqstream.cpp")], check=True)
    return str(exe)


def _table(o, k, add):
    c = o.ho_s00_new(k)
    rc = add(c)
    keys = np.zeros(max(1, o.ho_s00_select(c, 0, 1, 1 << 40, None)), dtype=np.uint64)
    n = o.ho_s00_select(c, 0, 1, 1 << 40, keys.ctypes.data_as(C.POINTER(C.c_uint64)))
    out = {int(x): o.ho_s00_count(c, 0, int(x)) for x in keys[:n]}
    total = o.ho_s00_total(c, 0)
    o.ho_s00_free(c)
    return rc, out, total


def _inputs(tmp_path):
    files = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "s00_*", "*"))):
        base = os.path.basename(p)
        if base.startswith(("expected", "case")):
            continue
        if p.endswith(".gz"):
            q = tmp_path / (os.path.basename(os.path.dirname(p)) + "_" + base[:-3])
            q.write_bytes(gzip.open(p).read())
            files.append(str(q))
        else:
            files.append(p)
    return files


@pytest.mark.parametrize("piece", [1 << 16, 7, 1])
def test_stream_holds_the_same_kmers_as_the_file(driver, oracle_lib, tmp_path, piece):
    o = oracle_lib
    for path in _inputs(tmp_path):
        if piece == 1 and os.path.getsize(path) > 60000:
            continue
        r = subprocess.run([driver, "-p", str(piece), path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, (path, r.stderr.decode()[-500:])
        assert b"runtime error" not in r.stderr
        stream = np.frombuffer(r.stdout, dtype=np.uint8)
        for k in (11, 21, 32):
            arr = (C.c_char_p * 1)(path.encode())
            rc, want, total = _table(o, k, lambda c: o.ho_s00_add_files(c, 0, arr, 1, 0))
            assert rc == 0
            _, got, total2 = _table(o, k, lambda c: o.ho_s00_add_stream(c, 0, stream.ctypes.data, stream.size))
            assert got == want and total == total2, (path, k)


def test_concatenated_inputs_and_errors(driver, oracle_lib, tmp_path):
    o = oracle_lib
    a, b = tmp_path / "a.fq", tmp_path / "b.fq"
    a.write_bytes(b"@r1\nACGTACGTACGTTTGACCA\n+\nIIIIIIIIIIIIIIIIIII\n@r2\nGGGTTTAAACCCGGGTTTAA")       # ends inside a record
    b.write_bytes(b"ACGTT\n+\n" + b"I" * 25 + b"\n@r3\nTTTTTTTTTTTTTTTTT\n+\nIIIIIIIIIIIIIIIII")
    r = subprocess.run([driver, "-c", "-p", "5", str(a), str(b)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    # `zcat a b | reader`: r2's sequence continues in the second file
    assert r.stdout == b"ACGTACGTACGTTTGACCA\nGGGTTTAAACCCGGGTTTAAACGTT\nTTTTTTTTTTTTTTTTT\n"
    arr = (C.c_char_p * 2)(str(a).encode(), str(b).encode())
    # the restatement reads a gz list as one stream; these are plain, so emulate with one joined file
    j = tmp_path / "j.fq"
    j.write_bytes(a.read_bytes() + b.read_bytes())
    arr1 = (C.c_char_p * 1)(str(j).encode())
    stream = np.frombuffer(r.stdout, dtype=np.uint8)
    rc, want, _ = _table(o, 11, lambda c: o.ho_s00_add_files(c, 0, arr1, 1, 0))
    _, got, _ = _table(o, 11, lambda c: o.ho_s00_add_stream(c, 0, stream.ctypes.data, stream.size))
    assert rc == 0 and got == want
    bad = {
        "short_qual.fq": b"@r1\nACGTACGTAC\n+\nIIII\n@r2\nGGGGGGG\n+\nIIIIIII\n",
        "long_qual.fq": b"@r1\nACGTACGTAC\n+\nIIIIIIIIIIIIII\n@r2\nGGGGGGG\n+\nIIIIIII\n",
        "no_header.fq": b"@r1\nACGTACGTAC\n+\nIIIIIIIIII\nACGTACGTAC\n",
        "no_qual.fq": b"@r1\nACGTACGTAC\n+\n",
        "not_seq.txt": b"XACGT\n",
    }
    for name, data in bad.items():
        p = tmp_path / name
        p.write_bytes(data)
        r = subprocess.run([driver, str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 3, name
        arrb = (C.c_char_p * 1)(str(p).encode())
        c = o.ho_s00_new(5)
        assert o.ho_s00_add_files(c, 0, arrb, 1, 0) < 0, name            # the restatement refuses the same inputs
        o.ho_s00_free(c)
    ok = {"seq_only.fq": (b"@r1\nACGTACGTAC\n", b"ACGTACGTAC\n"), "empty.fq": (b"", b""), "hdr.fq": (b"@r1\n", b"\n"),
          "crlf.fa": (b">x\r\nACG\r\nTTT\r\n\r\n>y\r\nAA\rCC\r\n", b"ACGTTT\nAA\rCC\n")}
    for name, (data, want) in ok.items():
        p = tmp_path / name
        p.write_bytes(data)
        r = subprocess.run([driver, "-p", "3", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and r.stdout == want, (name, r.stdout)

"""Host-only test of the DEVICE inflate's per-lane code and host logic (hast_amd/csrc/gz_core.h, gz_chain.h), driven by
tests/native/test_gz_core.cpp with plain loops standing in for the kernels: the same corpus as the host decoders' test
(every block type, member layout, header field, flush point, level and window size), chunks from 64 bytes to 1 MB (chunks
in which no block starts, blocks larger than a chunk, stored / fixed blocks at chunk borders, members that end inside a
chunk), symbol buffers too small for a block, candidates arriving segment by segment -- same bytes as zlib, CRC-32 and ISIZE
of every member checked by slices + GF(2) operators as the CRC kernel does it; truncated and damaged input are errors.
Built with ASAN + UBSAN.  (tests/test_gz_gpu.py runs the same corpus through the kernels themselves.)"""
import gzip
import os
import random
import subprocess
import zlib

import pytest

from tests.conftest import ROOT
from tests.test_inflate_cpu import CASES, FQ, fastq, member


@pytest.fixture(scope="module")
def driver_exe(tmp_path_factory):
    exe = tmp_path_factory.mktemp("gzcore") / "test_gz_core"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", str(exe),
                    os.path.join(ROOT, "tests", "native", "test_gz_core.cpp")], check=True)
    return str(exe)


class _Driver(str):
    """the driver's path; subprocess.run([driver, ...]) below starts it with this decoder's flag in front of the other arguments"""
    flag = None


@pytest.fixture(params=["lane", "wave"])
def driver(driver_exe, request, tmp_path):
    """lane: gz_core.h's decode_chunk (a lane per chunk, one symbol at a time); wave: the symbol loop of k_gz_decode restated with plain
    loops (-w: a token parsed at each of 64 bit offsets, the chain walked, rounds of at most 64 symbols, the ring of recent symbols)"""
    if request.param == "lane":
        return driver_exe
    sh = tmp_path / "driver_wave.sh"
    sh.write_text('#!/bin/sh\nexec "%s" -w "$@"\n' % driver_exe)
    sh.chmod(0o755)
    return str(sh)


def inflate_all(blob):
    """zlib's view of a gzip file: members one after the other, trailing garbage ignored (as gzread does)"""
    out, rest = b"", blob
    while rest[:2] == b"\x1f\x8b":
        d = zlib.decompressobj(31)
        out += d.decompress(rest)
        rest = d.unused_data
    return out


@pytest.mark.parametrize("name", sorted(n for n in CASES if n not in ("not_gzip",)))
def test_same_bytes_as_zlib(driver, tmp_path, name):
    p = tmp_path / (name + ".gz")
    p.write_bytes(CASES[name])
    want = inflate_all(CASES[name])
    for chunk, seg, room in ((32768, 64, 12), (1 << 20, 2, 12), (4096, 3, 12), (700, 50, 40), (64, 1000, 100), (32768, 1, 0.5), (5000, 7, 1.0)):
        if chunk < 1000 and len(CASES[name]) > 200_000:
            continue
        got = subprocess.run([driver, "-c", str(chunk), "-s", str(seg), "-r", str(room), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert got.returncode == 0, (name, chunk, seg, room, got.stderr[-300:])
        assert got.stdout == want, (name, chunk, seg, room)


@pytest.mark.parametrize("name", sorted(n for n in CASES if n not in ("not_gzip",)))
def test_the_chain_keeps_up_with_a_ring(driver, tmp_path, name):
    """What a file that goes round a ring on the device (gz_api.cpp "THE RING") needs of the chain: its end never falls further behind
    the segment being walked than a deflate block is long -- with slots that hold a block and with slots too small for ANY block of
    the stream (every block decoded by a follow-up job whose target is a candidate that holds no data), through members of stored
    blocks in which no search finds a start (the eager walk decodes on through them) -- while the end of the input is not in sight.
    The driver (-e) checks the distance after every segment; the bytes are zlib's."""
    p = tmp_path / (name + ".gz")
    p.write_bytes(CASES[name])
    want = inflate_all(CASES[name])
    for chunk, seg, room in ((4096, 3, 12), (4096, 4, 0.5), (1024, 16, 200), (32768, 2, 12), (700, 50, 1.0)):
        if chunk < 1000 and len(CASES[name]) > 200_000:
            continue
        got = subprocess.run([driver, "-e", str(262144 + chunk * seg), "-c", str(chunk), "-s", str(seg), "-r", str(room), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert got.returncode == 0, (name, chunk, seg, room, got.stderr[-300:])
        assert got.stdout == want, (name, chunk, seg, room)


def test_a_job_that_reaches_the_end_of_its_view_of_a_ring_is_followed_up(driver, tmp_path):
    """On the device a job sees a ring from the lap its first bit lies in to a piece behind that lap's end (ChunkJob.limit_bits).  A job
    that runs on from its first boundary into a block no search can find -- a member's FINAL block -- can reach that end with all of the
    input on the device: the chain keeps what the job committed and sends a follow-up job (whose view starts where IT starts) instead of
    reporting "input ends inside a compressed block" (round 6: tests/test_gz_gpu.py met it when the ring's size changed).  The driver's
    -v RING:PIECE gives every job that view; the rings the library gives the GPU test's geometries, over two level-9 members and members
    with stored blocks in between (16 KB x 5 chunks a pass under a ring of 704 KB is the case the GPU test met)."""
    rng = random.Random(77)
    fq = fastq(rng, 30_000)
    noise = bytes(rng.getrandbits(8) for _ in range(300_000))
    blobs = {"l9_members": member(fq[:1_000_000], 9) + member(fq[1_000_000:], 9),
             "stored_between": member(fq[:700_000], 6) + member(noise, 6) + member(fq[700_000:], 4)}
    met = 0
    for name, blob in blobs.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(blob)
        want = inflate_all(blob)
        for chunk, seg, room in ((16384, 5, 12), (32768, 2, 12)):
            # the ring gz_api.cpp gives such a stream under HAST_GZ_RING_BYTES=131072, pieces of 64 KB (two passes side by side, and one at a time), and its neighbours
            for passes in ((4, 3) if chunk == 16384 else (4,)):
                need = passes * chunk * seg + chunk * seg + 4 * 65536
                ring = (max(131072, need) + 65535) // 65536 * 65536
                for r in (ring, ring + 65536):
                    got = subprocess.run([driver, "-e", str(1 << 30), "-v", "%d:65536" % r, "-c", str(chunk), "-s", str(seg), "-r", str(room), str(p)],
                                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                    assert got.returncode == 0, (name, r, chunk, got.stderr[-300:])
                    assert got.stdout == want, (name, r, chunk)
                    met += b"view ended" in got.stderr
    assert met > 0                                     # (the case was there: the driver says when a job's view ended before the input did)


def test_what_the_waves_steps_are_made_of(driver_exe, tmp_path):
    """-w -S: the counters the measurements quote (profiles/round6_gz_decode_latency_ab.txt: on a FASTQ's lines nearly every round of
    k_gz_decode copies from further back than the ring of recent symbols) -- they add up, and a gzip -6 FASTQ has that property"""
    import re
    p = tmp_path / "fq.gz"
    p.write_bytes(CASES["fastq_l6"])
    r = subprocess.run([driver_exe, "-w", "-S", "-c", "16384", "-r", "20", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == inflate_all(CASES["fastq_l6"])
    m = re.search(rb"wave steps (\d+) \(with second-level tables (\d+)\) rounds (\d+) with_match (\d+) with_far_copy (\d+) symbols_in_rounds (\d+) tokens (\d+) matches (\d+) long_matches (\d+) long_far (\d+)", r.stderr)
    assert m, r.stderr[-300:]
    steps, full, rounds, with_match, with_far, syms, toks, matches, longs, longs_far = (int(x) for x in m.groups())
    assert full <= steps and with_far <= with_match <= rounds and matches <= toks and longs_far <= longs
    assert toks <= syms <= rounds * 64                  # (symbols of everything decoded: a chunk whose result the chain did not accept is decoded again)
    assert with_far * 2 > rounds                       # (most rounds reach further back than kRing = 512 symbols)


def test_search_kernels_strict_parse_equals_the_full_one(driver_exe):
    """header_parses8 (what a lane of k_gz_search runs: counts in packed words, a 128-byte table) == header_parses (gz_core.h's
    read_dynamic with zlib's completeness rules) on 400 000 random bit strings, half of them behind a valid code-length code"""
    r = subprocess.run([driver_exe, "-f", "400000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr[-300:]
    assert int(r.stdout.split()[0]) >= 0


def test_not_gzip_is_refused(driver, tmp_path):
    p = tmp_path / "plain.gz"
    p.write_bytes(CASES["not_gzip"])
    got = subprocess.run([driver, str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert got.returncode == 3 and b"not a gzip file" in got.stderr


def test_big_stream_long_runs_and_many_members(driver, tmp_path):
    rng = random.Random(9)
    data = fastq(rng, 40_000) + b"\x00" * 3_000_000 + fastq(rng, 20_000)
    p = tmp_path / "big.gz"
    with gzip.open(p, "wb", compresslevel=4) as f:
        f.write(data)
    for chunk, room in ((32768, 12), (8192, 3)):
        got = subprocess.run([driver, "-c", str(chunk), "-r", str(room), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert got.returncode == 0 and got.stdout == data, got.stderr[-300:]
    data, blob = b"", b""
    for _ in range(2000):
        d = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 400)))
        data += d
        blob += member(d, rng.choice([0, 1, 6, 9]))
    p = tmp_path / "many.gz"
    p.write_bytes(blob)
    for chunk in (32768, 300):
        got = subprocess.run([driver, "-c", str(chunk), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert got.returncode == 0 and got.stdout == data, got.stderr[-300:]


def test_random_streams_with_flush_points(driver, tmp_path):
    """sync / full flushes at random places (the empty stored blocks pigz and bgzip write between their pieces), random levels
    and strategies: a chunk's decode runs on through stored / fixed / final blocks behind its stop, which no search can find"""
    rng = random.Random(11)
    for it in range(12):
        data, blob = b"", b""
        for _ in range(rng.randint(1, 6)):
            c = zlib.compressobj(rng.choice([1, 4, 6, 9]), zlib.DEFLATED, 31, 9, rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_FIXED]))
            for _ in range(rng.randint(1, 30)):
                kind = rng.random()
                d = (bytes(rng.choice(b"ACGTN\n") for _ in range(rng.randint(0, 3000))) if kind < 0.6 else
                     bytes(rng.getrandbits(8) for _ in range(rng.randint(0, 500))) if kind < 0.8 else b"A" * rng.randint(0, 70000))
                data += d
                blob += c.compress(d)
                if rng.random() < 0.5:
                    blob += c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
            blob += c.flush()
        p = tmp_path / ("r%d.gz" % it)
        p.write_bytes(blob)
        for chunk, seg in ((32768, 64), (2048, 5), (300, 11)):
            r = subprocess.run([driver, "-c", str(chunk), "-s", str(seg), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0 and r.stdout == data, (it, chunk, seg, r.stderr[-300:])


def test_truncated_and_damaged_input_is_an_error_never_other_data(driver, tmp_path):
    blob = member(FQ[:400_000])
    rng = random.Random(3)
    want = FQ[:400_000]
    for cut in (len(blob) - 1, len(blob) - 5, len(blob) - 9, len(blob) // 2, 11, 3):
        p = tmp_path / "cut.gz"
        p.write_bytes(blob[:cut])
        r = subprocess.run([driver, "-c", "8192", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 3, cut
        assert want.startswith(r.stdout)                     # what was delivered before the error is a prefix of the real data
    for _ in range(25):
        b = bytearray(blob)
        at = rng.randrange(10, len(b) - 8)
        b[at] ^= 1 << rng.randrange(8)
        p = tmp_path / "flip.gz"
        p.write_bytes(bytes(b))
        r = subprocess.run([driver, "-c", "8192", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 3 or r.stdout == want, at    # (a flip in a stored block's padding bits changes nothing)

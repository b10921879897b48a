"""The device inflate (include/hast.h hast_gz_*, hast_amd/csrc/gz_kernels.hip) against zlib's bytes, through the C ABI: the
whole corpus of the host decoders' test (every block type, member layout, header field, flush point, level, window size),
with chunks of 64 bytes to 1 MB (chunks in which no block starts, blocks larger than a chunk, stored / fixed blocks at chunk
borders, members that end inside a chunk), passes of a few chunks (the symbol arenas taking turns, the next pass launched before the current one is walked), too little room per chunk
(follow-up jobs), reads of 1 byte to 4 MB; truncated and bit-flipped files are errors -- CRC-32 and ISIZE of every member are
checked on the way -- never other data."""
import gzip
import os
import random
import zlib

import numpy as np
import pytest

import hast_amd
from tests.test_gz_core_cpu import inflate_all
from tests.test_inflate_cpu import CASES, FQ, fastq, member

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    if not os.path.exists(hast_amd.lib_path()):
        hast_amd.build()
    with hast_amd.Context(21) as c:
        yield c


CONFIGS = ((0, 0, 0.0, 1 << 22), (1 << 20, 2, 12, 1 << 20), (4096, 3, 12, 65537), (700, 50, 40, 1 << 22), (64, 1000, 100, 1 << 22), (32768, 1, 0.5, 99_999), (5000, 7, 1.0, 1 << 22))


@pytest.mark.parametrize("name", sorted(n for n in CASES if n != "not_gzip"))
def test_same_bytes_as_zlib(ctx, tmp_path, name):
    p = tmp_path / (name + ".gz")
    p.write_bytes(CASES[name])
    want = inflate_all(CASES[name])
    for chunk, seg, room, piece in CONFIGS:
        if chunk and chunk < 1000 and len(CASES[name]) > 200_000:
            continue
        with hast_amd.GzReader(ctx, str(p), chunk, seg, room) as z:
            got = z.read_all(piece)
            st = z.stats()
        assert got == want, (name, chunk, seg, room, len(got), len(want))
        assert st["out_bytes"] == len(want) and st["compressed_bytes"] == len(CASES[name])


@pytest.mark.parametrize("ahead", [0, 1, 2])
def test_passes_side_by_side(ctx, tmp_path, monkeypatch, ahead):
    """HAST_GZ_AHEAD: how many further passes of a stream are on the GPU beside the one in front (gz_api.cpp hast_gz::ahead; default 1:
    two decode streams, three symbol arenas, three job arrays; 0 = one pass at a time, as until round 6; 2 = three side by side): a FASTQ
    of many passes, members back to back, stored blocks in between, with slots that hold a block and slots too small for any -- the bytes
    are zlib's in every mode, also through a ring of the compressed bytes and with damage in a late pass reported as an error."""
    monkeypatch.setenv("HAST_GZ_AHEAD", str(ahead))
    rng = random.Random(5 + ahead)
    fq = fastq(rng, 20_000)
    noise = bytes(rng.getrandbits(8) for _ in range(100_000))
    data = fq[:900_000] + noise + fq[900_000:]
    blob = member(fq[:900_000], 6) + member(noise, 6) + member(fq[900_000:], 9)
    p = tmp_path / "many_passes.gz"
    p.write_bytes(blob)
    for chunk, seg, room in ((4096, 4, 50), (16384, 5, 12), (1024, 16, 200), (4096, 4, 0)):
        with hast_amd.GzReader(ctx, str(p), chunk, seg, room) as z:
            got = z.read_all(1 << 20)
            st = z.stats()
        assert got == data, (ahead, chunk, seg, room, len(got), len(data))
        assert st["members"] == 3
    monkeypatch.setenv("HAST_GZ_PIECE_BYTES", "65536")
    monkeypatch.setenv("HAST_GZ_RING_BYTES", "131072")
    with hast_amd.GzReader(ctx, str(p), 4096, 4, 50) as z:
        got = z.read_all(99_999)
        st = z.stats()
    assert got == data and st["ring_bytes"] > 0 and st["ring_laps"] >= 1, (ahead, st)
    bad = bytearray(blob)
    bad[len(blob) * 3 // 4] ^= 0x10
    q = tmp_path / "damaged_late.gz"
    q.write_bytes(bytes(bad))
    with pytest.raises(hast_amd.HastError):
        with hast_amd.GzReader(ctx, str(q), 4096, 4, 50) as z:
            z.read_all(1 << 20)


def test_tiny_reads_and_reads_across_batches(ctx, tmp_path):
    p = tmp_path / "fq.gz"
    p.write_bytes(CASES["fastq_l6"])
    want = inflate_all(CASES["fastq_l6"])
    with hast_amd.GzReader(ctx, str(p), 4096, 5, 12) as z:
        got = bytearray()
        rng = random.Random(2)
        while True:
            a = z.read(rng.choice([1, 2, 7, 100, 4097, 300_000]))
            if a.size == 0:
                break
            got += a.tobytes()
        assert bytes(got) == want
        assert z.read(100).size == 0                     # the end stays the end


def test_not_gzip_and_missing_files(ctx, tmp_path):
    p = tmp_path / "plain.gz"
    p.write_bytes(CASES["not_gzip"])
    with pytest.raises(hast_amd.HastError) as ei:
        hast_amd.GzReader(ctx, str(p))
    assert ei.value.status == 9                         # HAST_ERR_UNSUPPORTED: the caller inflates (passes through) on the host
    with pytest.raises(hast_amd.HastError) as ei:
        hast_amd.GzReader(ctx, str(tmp_path / "nope.gz"))
    assert ei.value.status == 8
    e = tmp_path / "empty.gz"
    e.write_bytes(b"")
    with hast_amd.GzReader(ctx, str(e)) as z:
        assert z.read_all() == b""


def test_big_stream_long_runs_and_many_members(ctx, tmp_path):
    rng = random.Random(9)
    data = fastq(rng, 40_000) + b"\x00" * 3_000_000 + fastq(rng, 20_000)
    p = tmp_path / "big.gz"
    with gzip.open(p, "wb", compresslevel=4) as f:
        f.write(data)
    for chunk, seg, room in ((0, 0, 0), (8192, 40, 3)):
        with hast_amd.GzReader(ctx, str(p), chunk, seg, room) as z:
            assert z.read_all() == data
            st = z.stats()
        assert st["members"] == 1 and st["accepted"] >= 1
    data, blob = b"", b""
    for _ in range(2000):
        d = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 400)))
        data += d
        blob += member(d, rng.choice([0, 1, 6, 9]))
    p = tmp_path / "many.gz"
    p.write_bytes(blob)
    for chunk in (0, 300):
        with hast_amd.GzReader(ctx, str(p), chunk) as z:
            assert z.read_all() == data
            assert z.stats()["members"] == 2000


def test_a_file_larger_than_its_ring_on_the_device(ctx, tmp_path, monkeypatch):
    """the compressed bytes go round a ring on the device (gz_api.cpp "THE RING"; by default for files beyond 2 GB): pieces of 64 KB,
    a ring of 192-320 KB, files of 1-3 MB -- a FASTQ at levels 1, 6 and 9, members back to back, stored blocks (random bytes) in
    between; chunks of 1 to 32 KB, passes of 2 to 16 chunks, reads of 977 bytes to 4 MB: the same bytes as zlib, the ring in
    use, the uploader made to wait for the chain"""
    rng = random.Random(77)
    monkeypatch.setenv("HAST_GZ_PIECE_BYTES", "65536")
    monkeypatch.setenv("HAST_GZ_RING_BYTES", "131072")
    fq = fastq(rng, 30_000)
    noise = bytes(rng.getrandbits(8) for _ in range(300_000))
    blobs = {
        "l6": (fq, member(fq, 6)),
        "l1": (fq, member(fq, 1)),
        "l9_members": (fq[:1_000_000] + fq[1_000_000:], member(fq[:1_000_000], 9) + member(fq[1_000_000:], 9)),
        "stored_between": (fq[:700_000] + noise + fq[700_000:], member(fq[:700_000], 6) + member(noise, 6) + member(fq[700_000:], 4)),
    }
    for name, (data, blob) in blobs.items():
        assert len(blob) > 900_000, (name, len(blob))
        p = tmp_path / (name + ".gz")
        p.write_bytes(blob)
        # (the last geometries: a block of these streams is 100-250 k symbols of output and does NOT fit a chunk's slot -- every block is
        # decoded by a follow-up job with more room, whose target is the next candidate's start although that candidate holds no data)
        for chunk, seg, room, piece in ((4096, 4, 50, 1 << 22), (1024, 16, 200, 977), (32768, 2, 0, 65537), (0, 3, 0, 1 << 20), (16384, 5, 12, 4099), (4096, 4, 0, 1 << 22)):
            with hast_amd.GzReader(ctx, str(p), chunk, seg, room) as z:
                got = z.read_all(piece)
                st = z.stats()
            assert got == data, (name, chunk, seg, len(got), len(data))
            assert st["ring_bytes"] >= 131072 and st["upload_waited_for_ring"] > 0, (name, chunk, seg, st)


def test_random_streams_with_flush_points(ctx, tmp_path):
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
        for chunk, seg in ((0, 0), (2048, 5), (300, 11)):
            with hast_amd.GzReader(ctx, str(p), chunk, seg) as z:
                assert z.read_all(rng.choice([1 << 22, 5000, 977])) == data, (it, chunk, seg)


def test_truncated_and_damaged_input_is_an_error_never_other_data(ctx, tmp_path):
    blob = member(FQ[:400_000])
    want = FQ[:400_000]
    rng = random.Random(3)

    def run(data):
        p = tmp_path / "x.gz"
        p.write_bytes(data)
        got = bytearray()
        try:
            with hast_amd.GzReader(ctx, str(p), 8192) as z:
                while True:
                    a = z.read(1 << 20)
                    if a.size == 0:
                        return bytes(got), None
                    got += a.tobytes()
        except hast_amd.HastError as e:
            return bytes(got), e

    for cut in (len(blob) - 1, len(blob) - 5, len(blob) - 9, len(blob) // 2, 11, 3):
        got, err = run(blob[:cut])
        assert err is not None and err.status in (8, 9), cut
        assert want.startswith(got)                          # what was delivered in front of the damage is real data
    for _ in range(25):
        b = bytearray(blob)
        at = rng.randrange(10, len(b) - 8)
        b[at] ^= 1 << rng.randrange(8)
        got, err = run(bytes(b))
        assert err is not None or got == want, at


def test_a_level6_fastq_of_100_mb_with_default_geometry(ctx, tmp_path):
    """default geometry (16-KB chunks, 6144 per pass, 20 symbols of room per compressed byte) on a stream of ~1900 chunks: every candidate the search
    finds is a real boundary or is skipped by the chain, few follow-up jobs, output == zlib's"""
    rng = np.random.default_rng(4)
    n = 300_000
    bases = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n, 150))
    qual = rng.choice(np.frombuffer(b"FFFFF:F,F#", np.uint8), (n, 150))
    recs = []
    for i in range(n):
        recs.append(b"@V300R%09d#%d_%d_%d/1\n" % (i, i % 1536 + 1, (i * 7) % 1536 + 1, (i * 13) % 1536 + 1) + bases[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n")
    data = b"".join(recs)
    p = tmp_path / "fq100.gz"
    with gzip.open(p, "wb", compresslevel=6) as f:
        f.write(data)
    with hast_amd.GzReader(ctx, str(p)) as z:
        got = z.read_all(16 << 20)
        st = z.stats()
    assert got == data
    assert st["followup_jobs"] <= st["chunks"] // 20 + 2, st


def test_a_pass_with_fewer_slots_than_chunks_with_a_block_start(ctx, tmp_path, monkeypatch):
    """the symbol arena of a pass holds slots for 70 % of its chunks (4 in 10 chunks of a FASTQ hold no block start), handed out by the
    decode kernel itself; with slots for 20 % most chunks find the pool empty, report "found, no room" and are decoded by follow-up
    jobs: same bytes, many follow-up jobs"""
    rng = np.random.default_rng(9)
    n = 60_000
    bases = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n, 150))
    qual = rng.choice(np.frombuffer(b"FFFFF:F,F#", np.uint8), (n, 150))
    data = b"".join(b"@V300R%09d#%d_%d_%d/1\n" % (i, i % 1536 + 1, (i * 7) % 1536 + 1, (i * 13) % 1536 + 1) + bases[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n" for i in range(n))
    p = tmp_path / "fq20.gz"
    with gzip.open(p, "wb", compresslevel=6) as f:
        f.write(data)
    jobs = {}
    for frac in ("0.2", "1.0"):
        monkeypatch.setenv("HAST_GZ_SLOT_FRACTION", frac)
        with hast_amd.GzReader(ctx, str(p), 8192, 64, 24) as z:
            got = z.read_all(1 << 20)
            jobs[frac] = z.stats()["followup_jobs"]
        assert got == data, frac
    assert jobs["0.2"] > jobs["1.0"] + 20, jobs


def test_a_stream_closed_early_with_passes_in_flight(ctx, tmp_path):
    """the reader gives up after a few bytes (or after none) while the producer has the next segment's nominal pass on the GPU: close
    must drain the streams, free the arenas and return -- with small passes (several segments in flight) and with the default ones.
    (Found by this test: a CU-masked stream created right after another had been destroyed hung inside the runtime every other time;
    such streams now come out of a pool and go back to it.)"""
    rng = np.random.default_rng(5)
    data = rng.choice(np.frombuffer(b"ACGT\n", np.uint8), 6_000_000).tobytes()
    p = tmp_path / "early.gz"
    p.write_bytes(gzip.compress(data, 6))
    for chunk, seg, take in ((4096, 3, 1000), (4096, 3, 0), (0, 0, 70_000), (2048, 1, 5_000_000)):
        with hast_amd.GzReader(ctx, str(p), chunk, seg) as z:
            if take:
                got = bytearray()
                while len(got) < take:
                    a = z.read(min(65536, take - len(got)))
                    assert a.size
                    got += a.tobytes()
                assert bytes(got) == data[:take]
    # and the context is still good for a whole stream
    with hast_amd.GzReader(ctx, str(p)) as z:
        assert z.read_all(1 << 20) == data

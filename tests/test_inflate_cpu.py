"""Host-only test of the ingest's gzip decoder (hast_amd/csrc/fast_inflate.h) against zlib: same bytes out for every kind of
deflate block, member layout, buffer size and call size; errors for truncated / damaged input.  Built with ASAN + UBSAN."""
import gzip
import os
import random
import struct
import subprocess
import zlib

import pytest

from tests.conftest import ROOT


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = tmp_path_factory.mktemp("inflate") / "test_inflate"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", str(exe),
                    os.path.join(ROOT, "tests", "native", "test_inflate.cpp"), "-lz", "-pthread"], check=True)
    return str(exe)


def member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=31, flags=0, extra=b""):
    if flags == 0:
        c = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
        return c.compress(data) + c.flush()
    # hand-made header with optional fields (RFC 1952): FEXTRA 4, FNAME 8, FCOMMENT 16, FHCRC 2
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    raw = c.compress(data) + c.flush()
    head = b"\x1f\x8b\x08" + bytes([flags]) + b"\x00\x00\x00\x00\x00\x03"
    if flags & 4:
        head += struct.pack("<H", len(extra)) + extra
    if flags & 8:
        head += b"file name.fq\x00"
    if flags & 16:
        head += b"a comment\x00"
    if flags & 2:
        head += struct.pack("<H", zlib.crc32(head) & 0xFFFF)
    return head + raw + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def fastq(rng, n):
    out = []
    for i in range(n):
        L = rng.choice([100, 150, 150, 37])
        out.append("@V300R%09d#%d_%d_%d/1\n%s\n+\n%s\n" % (i, i % 1536, i % 977, i % 3, "".join(rng.choice("ACGT") for _ in range(L)),
                                                           "".join(rng.choice("FFFFFF:,F#") for _ in range(L))))
    return "".join(out).encode()


def corpus():
    rng = random.Random(5)
    fq = fastq(rng, 6000)
    rnd = bytes(rng.getrandbits(8) for _ in range(300_000))
    rep = (b"ACGT" * 50 + b"\n") * 4000 + b"A" * 100_000 + bytes(range(256)) * 300
    cases = {
        "fastq_l6": member(fq), "fastq_l1": member(fq, 1), "fastq_l9": member(fq, 9),
        "fastq_fixed": member(fq, 6, zlib.Z_FIXED), "fastq_huffman_only": member(fq, 6, zlib.Z_HUFFMAN_ONLY), "fastq_rle": member(fq, 6, zlib.Z_RLE),
        "stored": member(fq[:200_000], 0), "random_l6": member(rnd), "random_l0": member(rnd, 0), "repeats": member(rep, 9),
        "empty_member": member(b""), "one_byte": member(b"A"), "no_input": b"",
        "members": member(fq[:70_000]) + member(b"") + member(fq[70_000:140_000], 1) + member(rnd[:5000], 0) + member(fq[140_000:], 9),
        "header_fields": member(fq[:50_000], flags=4 | 8 | 16 | 2, extra=b"xx\x03\x00abc") + member(fq[50_000:90_000], flags=8),
        "small_window": member(fq[:100_000], 6, wbits=16 + 9),
        "trailing_garbage": member(fq[:30_000]) + b"\x00\x00\x00garbage that is not a member",
        "not_gzip": fq[:40_000],
    }
    return cases, fq


CASES, FQ = corpus()


@pytest.mark.parametrize("name", sorted(CASES))
def test_same_bytes_as_zlib(driver, tmp_path, name):
    p = tmp_path / (name + ".gz")
    p.write_bytes(CASES[name])
    want = subprocess.run([driver, "-z", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert want.returncode == 0
    for piece, inbuf in ((1 << 20, 1 << 20), (65536, 4096), (7, 64), (1, 333), (1000, 100)):
        if piece < 100 and len(want.stdout) > 400_000:
            continue
        got = subprocess.run([driver, "-p", str(piece), "-i", str(inbuf), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert got.returncode == 0, (name, piece, inbuf, got.stderr[-300:])
        assert got.stdout == want.stdout, (name, piece, inbuf)


def test_big_stream_and_python_gzip_writer(driver, tmp_path):
    rng = random.Random(9)
    data = fastq(rng, 40_000) + b"\x00" * 3_000_000 + fastq(rng, 20_000)
    p = tmp_path / "big.gz"
    with gzip.open(p, "wb", compresslevel=4) as f:
        f.write(data)
    got = subprocess.run([driver, "-p", "16777216", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert got.returncode == 0 and got.stdout == data


def test_thousands_of_tiny_members_with_tiny_input_buffers(driver, tmp_path):
    """BGZF-like input: every member ends with a byte-aligned trailer right after the last bits of its data, so the decoder
    keeps handing whole bytes back from its bit buffer -- also across refills of the input buffer (this once stepped
    in front of the buffer)"""
    rng = random.Random(1)
    data, blob = b"", b""
    for _ in range(3000):
        d = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 400)))
        data += d
        blob += member(d, rng.choice([0, 1, 6, 9]))
    p = tmp_path / "many.gz"
    p.write_bytes(blob)
    for inbuf in (40, 64, 77, 128, 1000, 1 << 20):
        for piece in (1 << 20, 4096, 13):
            r = subprocess.run([driver, "-i", str(inbuf), "-p", str(piece), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0 and r.stdout == data, (inbuf, piece, r.stderr[-300:])


def test_random_streams_with_flush_points(driver, tmp_path):
    """members written with sync / full flushes at random places (empty stored blocks in mid-stream, as pigz and bgzip
    emit them), random levels and strategies, decoded with random buffer and call sizes"""
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
        for _ in range(3):
            inbuf, piece = rng.choice([33, 64, 200, 4096, 1 << 20]), rng.choice([1 << 20, 5000, 97])
            r = subprocess.run([driver, "-i", str(inbuf), "-p", str(piece), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0 and r.stdout == data, (it, inbuf, piece, r.stderr[-300:])


def bgzf(data, tail=b""):
    def block(d):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = c.compress(d) + c.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(raw) + 25) + raw +
                struct.pack("<II", zlib.crc32(d) & 0xFFFFFFFF, len(d)))
    out = b"".join(block(data[i:i + 65280]) for i in range(0, len(data), 65280))
    return out + (member(tail) if tail else block(b""))


def test_blocked_gzip_is_inflated_in_parallel_and_checked(driver, tmp_path):
    rng = random.Random(21)
    data = fastq(rng, 9000) + bytes(rng.getrandbits(8) for _ in range(100_000))
    p = tmp_path / "b.gz"
    p.write_bytes(bgzf(data))
    want = subprocess.run([driver, "-z", str(p)], stdout=subprocess.PIPE).stdout
    assert want == data                                            # zlib reads the same file as ordinary multi-member gzip
    for piece, threads in ((1 << 24, 8), (200_000, 3), (70_000, 1), (65_536, 2), (9000, 4), (1, 2)):
        r = subprocess.run([driver, "-b", "-p", str(piece), "-t", str(threads), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and r.stdout == data, (piece, threads, r.stderr[-300:])
    # an ordinary gzip member behind the blocks: the serial decoder takes over
    p.write_bytes(bgzf(data[:500_000], tail=data[500_000:]))
    r = subprocess.run([driver, "-b", "-p", "300000", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == data, r.stderr[-300:]
    whole = bgzf(data)
    # padding after the last member: ignored whatever its length, as gzread and the serial decoder do (a few bytes used to be
    # "the file ends inside a block" while 18 or more were ignored)
    for pad in (b"\0", b"\0" * 5, b"\n" * 17, b"\0" * 18, b"junk" * 20):
        p.write_bytes(whole + pad)
        assert subprocess.run([driver, "-z", str(p)], stdout=subprocess.PIPE).stdout == data          # zlib: ignored
        r = subprocess.run([driver, "-b", "-p", "300000", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and r.stdout == data, (pad, r.stderr[-300:])
    # ... but the beginning of another member is a truncated file
    p.write_bytes(whole + whole[:9])
    r = subprocess.run([driver, "-b", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 3
    for damage in ("flip", "cut", "cut_in_header"):
        b = bytearray(whole)
        if damage == "flip":
            b[len(b) // 2] ^= 0x20
        elif damage == "cut":
            b = b[: len(b) // 2]
        else:
            b = b[: len(b) - 28 - 20 + 7] if False else b[:7]
        p.write_bytes(bytes(b))
        r = subprocess.run([driver, "-b", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode in (3, 4), (damage, r.returncode)      # 4: too short to be recognised as BGZF at all
        assert b"AddressSanitizer" not in r.stderr and b"runtime error" not in r.stderr


def test_damaged_input_is_an_error(driver, tmp_path):
    good = member(FQ[:120_000]) + member(FQ[120_000:200_000], 1)
    rng = random.Random(3)
    for cut in (5, 11, 100, len(good) // 3, len(good) - 9, len(good) - 1):
        p = tmp_path / ("cut%d.gz" % cut)
        p.write_bytes(good[:cut])
        r = subprocess.run([driver, "-q", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 3, (cut, r.stderr)
    flipped = 0
    for _ in range(40):
        b = bytearray(good)
        i = rng.randrange(12, len(b))
        b[i] ^= 1 << rng.randrange(8)
        p = tmp_path / "flip.gz"
        p.write_bytes(bytes(b))
        r = subprocess.run([driver, str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode in (0, 3)
        assert b"AddressSanitizer" not in r.stderr and b"runtime error" not in r.stderr
        if r.returncode == 0:                      # a flip can only go unnoticed if it did not change the data (header bytes)
            assert r.stdout == FQ[:200_000]
        else:
            flipped += 1
    assert flipped > 30


# ---- several threads on ONE ordinary gzip stream (hast_amd/csrc/par_inflate.h) --------------------------------------------
# chunk sizes far below a deflate block's size force every path: chunks in which no block starts, chunks that starve,
# boundaries in front of stored / fixed blocks (the search only knows dynamic ones), members that end inside a chunk.
PAR_SHAPES = ((1 << 20, 4, 1 << 20), (40_000, 3, 100_000), (5_000, 8, 65_536), (700, 2, 1 << 20), (64, 5, 7777), (3_000, 1, 1))


@pytest.mark.parametrize("name", sorted(n for n in CASES if n not in ("no_input", "not_gzip")))
def test_parallel_inflate_same_bytes_as_zlib(driver, tmp_path, name):
    p = tmp_path / (name + ".gz")
    p.write_bytes(CASES[name])
    want = subprocess.run([driver, "-z", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert want.returncode == 0
    for chunk, threads, piece in PAR_SHAPES:
        if piece < 100 and len(want.stdout) > 400_000:
            continue
        got = subprocess.run([driver, "-P", "-c", str(chunk), "-t", str(threads), "-p", str(piece), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert got.returncode == 0, (name, chunk, threads, piece, got.stderr[-300:])
        assert got.stdout == want.stdout, (name, chunk, threads, piece)


def test_parallel_inflate_leaves_other_inputs_to_the_serial_decoder(driver, tmp_path):
    for name in ("no_input", "not_gzip"):
        p = tmp_path / (name + ".gz")
        p.write_bytes(CASES[name])
        assert subprocess.run([driver, "-P", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode == 4


def test_parallel_inflate_random_streams_with_flush_points(driver, tmp_path):
    rng = random.Random(12)
    for it in range(10):
        data, blob = b"", b""
        for _ in range(rng.randint(1, 5)):
            c = zlib.compressobj(rng.choice([1, 4, 6, 9]), zlib.DEFLATED, 31, 9, rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_FIXED]))
            for _ in range(rng.randint(1, 30)):
                kind = rng.random()
                d = (bytes(rng.choice(b"ACGTN\n") for _ in range(rng.randint(0, 30000))) if kind < 0.6 else
                     bytes(rng.getrandbits(8) for _ in range(rng.randint(0, 5000))) if kind < 0.8 else b"A" * rng.randint(0, 700000))
                data += d
                blob += c.compress(d)
                if rng.random() < 0.3:
                    blob += c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
            blob += c.flush()
        p = tmp_path / ("r%d.gz" % it)
        p.write_bytes(blob)
        for _ in range(3):
            chunk, threads, piece = rng.choice([64, 300, 2000, 20000, 1 << 20]), rng.randint(1, 6), rng.choice([1 << 20, 5000, 97])
            r = subprocess.run([driver, "-P", "-c", str(chunk), "-t", str(threads), "-p", str(piece), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0 and r.stdout == data, (it, chunk, threads, piece, r.stderr[-300:])


def test_parallel_inflate_big_fastq_every_level(driver, tmp_path):
    rng = random.Random(13)
    data = fastq(rng, 60_000)
    for level in (1, 6, 9):
        p = tmp_path / ("big%d.gz" % level)
        p.write_bytes(member(data[:7_000_000], level) + member(data[7_000_000:], level))
        for chunk, threads in ((1 << 20, 4), (100_000, 7), (30_000, 3)):
            r = subprocess.run([driver, "-P", "-c", str(chunk), "-t", str(threads), "-p", str(1 << 24), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0 and r.stdout == data, (level, chunk, threads, r.stderr[-300:])


def test_parallel_inflate_damaged_input_is_an_error(driver, tmp_path):
    good = member(FQ[:400_000]) + member(FQ[400_000:700_000], 1)
    rng = random.Random(4)
    for cut in (5, 11, 100, len(good) // 3, len(good) // 2, len(good) - 9, len(good) - 1):
        p = tmp_path / ("cut%d.gz" % cut)
        p.write_bytes(good[:cut])
        for chunk in (1 << 20, 9_000):
            r = subprocess.run([driver, "-q", "-P", "-c", str(chunk), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 3, (cut, chunk, r.stderr)
    flipped = 0
    for _ in range(60):
        b = bytearray(good)
        i = rng.randrange(12, len(b))
        b[i] ^= 1 << rng.randrange(8)
        p = tmp_path / "flip.gz"
        p.write_bytes(bytes(b))
        r = subprocess.run([driver, "-P", "-c", str(rng.choice([1 << 20, 20_000, 3_000])), "-t", "3", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode in (0, 3)
        assert b"AddressSanitizer" not in r.stderr and b"runtime error" not in r.stderr
        if r.returncode == 0:                      # a flip can only go unnoticed if it did not change the data (header bytes)
            assert r.stdout == FQ[:700_000]
        else:
            flipped += 1
    assert flipped > 45


def test_parallel_inflate_has_no_data_races(tmp_path):
    """the same decoder under ThreadSanitizer: the producer thread, its decode pool, the caller's resolve pool"""
    exe = tmp_path / "test_inflate_tsan"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-o", str(exe), os.path.join(ROOT, "tests", "native", "test_inflate.cpp"),
                    "-lz", "-pthread"], check=True)
    p = tmp_path / "m.gz"
    p.write_bytes(CASES["members"] + CASES["fastq_l1"])
    want = subprocess.run([str(exe), "-z", str(p)], stdout=subprocess.PIPE).stdout
    for chunk, threads, piece in ((3000, 6, 65536), (100_000, 3, 1 << 20), (1 << 20, 4, 9999)):
        r = subprocess.run([str(exe), "-P", "-c", str(chunk), "-t", str(threads), "-p", str(piece), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and r.stdout == want
        assert b"ThreadSanitizer" not in r.stderr, r.stderr[-2000:]


def test_parallel_inflate_bounds_a_chunks_output(driver, tmp_path):
    """highly compressible data (1 MB of deflate can be 1 GB of output): a chunk stops at the first block boundary behind its
    output budget and the next batch goes on from there"""
    rng = random.Random(14)
    data = b"\0" * 6_000_000 + fastq(rng, 3000) + b"ACGT" * 1_000_000 + fastq(rng, 3000) + b"N" * 3_000_000
    p = tmp_path / "z.gz"
    p.write_bytes(member(data, 6))
    for chunk, threads, max_out in ((1 << 20, 4, 100_000), (2_000, 3, 50_000), (1 << 20, 2, 1), (500, 5, 1 << 30)):
        r = subprocess.run([driver, "-P", "-c", str(chunk), "-t", str(threads), "-m", str(max_out), "-p", str(1 << 22), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and r.stdout == data, (chunk, threads, max_out, r.stderr[-300:])


# ---- the byte source of every CLI (hast_amd/csrc/ingest.h: BlockSource) picks the decoder by what the file is ---------------------
def test_block_source_routes_every_kind_of_input(tmp_path):
    """plain file, ordinary gzip (parallel inflate, serial decoder, zlib), blocked gzip with and without an ordinary member
    behind it, a .gz name on plain bytes, an empty file -- through the background reader and through read_into(), alone and
    several files open at once (they share the inflate-thread budget); a damaged .gz is an error on every route"""
    exe = tmp_path / "test_blocksource"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread", "-o", str(exe),
                    os.path.join(ROOT, "tests", "native", "test_blocksource.cpp"), "-lz"], check=True)
    rng = random.Random(31)
    data = fastq(rng, 12000)
    files = {"plain.fq": data, "ord.fq.gz": member(data, 6), "two.fq.gz": member(data[:1_000_000], 1) + member(data[1_000_000:], 9),
             "blocked.fq.gz": bgzf(data), "blocked_tail.fq.gz": bgzf(data[:900_000], tail=data[900_000:]), "notgz.fq.gz": data, "empty.fq.gz": b"",
             "empty.fq": b""}
    for name, blob in files.items():
        (tmp_path / name).write_bytes(blob)
    want = {name: (b"" if name.startswith("empty") else data) for name in files}
    envs = [{}, {"HAST_GZ_THREADS": "1"}, {"HAST_GZ_THREADS": "5", "HAST_GZ_BUDGET": "5"}, {"HAST_INFLATE": "zlib"}, {"HAST_BGZF_THREADS": "3"}]
    for env in envs:
        for name in files:
            for extra in ([], ["-u"], ["-b", "70000"], ["-u", "-b", "4194304"]):
                r = subprocess.run([str(exe)] + extra + [str(tmp_path / name)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
                assert r.returncode == 0 and r.stdout == want[name], (env, name, extra, r.stderr[-300:])
    # all of them open at once
    names = [n for n in files]
    for env in envs[:3]:
        r = subprocess.run([str(exe)] + [str(tmp_path / n) for n in names], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert r.returncode == 0 and r.stdout == b"".join(want[n] for n in names), (env, r.stderr[-300:])
    # damage
    whole = files["ord.fq.gz"]
    (tmp_path / "cut.fq.gz").write_bytes(whole[:len(whole) // 2])
    flipped = bytearray(whole)
    flipped[len(flipped) // 2] ^= 0x10
    (tmp_path / "flip.fq.gz").write_bytes(bytes(flipped))
    for env in envs[:4]:
        for name in ("cut.fq.gz", "flip.fq.gz"):
            for extra in ([], ["-u"]):
                r = subprocess.run([str(exe)] + extra + [str(tmp_path / name)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
                assert r.returncode == 3, (env, name, extra, r.returncode, r.stderr[-300:])
                assert b"AddressSanitizer" not in r.stderr and b"runtime error" not in r.stderr
    # a read that FAILS is not the end of the input (here: the "file" is a directory, every read gives EISDIR): an error on
    # every route -- it used to pass as an empty input with exit code 0
    for d in ("adir.fq", "adir.fq.gz"):
        (tmp_path / d).mkdir()
        for env in envs[:4]:
            for extra in ([], ["-u"], ["-u", "-b", "16777216"], ["-b", "16777216"]):
                r = subprocess.run([str(exe)] + extra + [str(tmp_path / d)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
                assert r.returncode == 3 and r.stdout == b"", (d, env, extra, r.returncode, r.stderr[-300:])
                assert b"AddressSanitizer" not in r.stderr and b"runtime error" not in r.stderr

"""GPU parity of stage 00 (parent-unique k-mer sets): hast_kc_* and the unshared_kmers program against the oracle
(oracle/s00_oracle.c, pinned on the reference script's own outputs) and against those committed outputs."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

import hast_amd
from hast_amd import KcSynth, KmerCounter
from tests.conftest import golden_cases, run_s00_case

pytestmark = pytest.mark.gpu
U64P = C.POINTER(C.c_uint64)


@pytest.fixture(scope="module")
def built():
    hast_amd.build()
    return hast_amd.lib()


def random_stream(rng, n, k):
    """bases with everything the counting rules distinguish: both cases, N/n, IUPAC, separators, CR, repeats"""
    out = bytearray()
    motifs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(k, 3 * k))) for _ in range(40)]
    while len(out) < n:
        r = rng.random()
        if r < 0.35:
            s = rng.choice(motifs)
            if rng.random() < 0.5:
                s = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
        elif r < 0.40:
            s = rng.choice("ACGT") * rng.randint(1, 4 * k)
        else:
            s = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 200)))
        if rng.random() < 0.2:
            s = s.lower()
        out += s.encode()
        r = rng.random()
        if r < 0.25:
            out += b"\n"
        elif r < 0.35:
            out += rng.choice([b"N", b"n", b"R", b"\r", b"NNNN", b"-", b"\x00", b"\xff", b"@", b">"])
    return np.frombuffer(bytes(out[:n]), dtype=np.uint8).copy()


def oracle_table(o, k, streams):
    c = o.ho_s00_new(k)
    for parent, data in streams:
        o.ho_s00_add_stream(c, parent, data.ctypes.data, data.size)
    return c


def oracle_select(o, c, parent, lo, hi):
    n = o.ho_s00_select(c, parent, lo, hi, None)
    keys = np.zeros(max(n, 1), dtype=np.uint64)
    o.ho_s00_select(c, parent, lo, hi, keys.ctypes.data_as(U64P))
    return keys[:n]


def oracle_histo(o, c, parent):
    h = np.zeros(hast_amd.KC_HISTO_HIGH + 2, dtype=np.uint64)
    o.ho_s00_histo(c, parent, h.ctypes.data_as(U64P))
    return h


def key_text(o, keys, k):
    buf = C.create_string_buffer(k + 1)
    out = []
    for x in keys:
        o.ho_s00_key_to_str(int(x), k, buf)
        out.append(buf.value + b"\n")
    return b"".join(out)


def check_against_oracle(o, kc, c, k, bounds):
    st = kc.stats()
    assert st["distinct"] == (o.ho_s00_distinct(c, 0), o.ho_s00_distinct(c, 1))
    assert st["total"] == (o.ho_s00_total(c, 0), o.ho_s00_total(c, 1))
    for p in (0, 1):
        assert np.array_equal(kc.histo(p), oracle_histo(o, c, p)), p


@pytest.mark.parametrize("k", [21, 31, 32, 11, 5, 1, 16])
def test_counts_histograms_selections_vs_oracle(built, oracle_lib, k):
    o = oracle_lib
    rng = random.Random(100 + k)
    pat, mat = random_stream(rng, 300_000, k), random_stream(rng, 260_000, k)
    shared = random_stream(rng, 150_000, k)                      # k-mers both parents have
    c = oracle_table(o, k, [(0, pat), (0, shared), (1, mat), (1, shared)])
    bounds = [(1, 1 << 30), (2, 5), (1, 1), (3, 3), (7, 2)]
    with KmerCounter(k, table_bytes=64 << 20) as kc:
        kc.count(0, pat)
        kc.count(1, mat)
        for p in (0, 1):
            kc.count(p, shared)
        kc.sync()
        check_against_oracle(o, kc, c, k, bounds)
        for lo, hi in bounds:
            got = [kc.select(p, lo, hi) for p in (0, 1)]
            want = [oracle_select(o, c, p, lo, hi) for p in (0, 1)]
            assert got == [w.size for w in want], (lo, hi)
        # the selections accumulate: what has been appended so far, sorted, is the sorted union with multiplicity
        kc.release_table()
        for p in (0, 1):
            allw = np.sort(np.concatenate([oracle_select(o, c, p, lo, hi) for lo, hi in bounds]))
            n = kc.selection_sort(p)
            assert n == allw.size
            assert kc.selection_text(p, 0, n) == key_text(o, allw, k)
            if n > 10:
                assert kc.selection_text(p, 3, 5) == key_text(o, allw[3:8], k)
            # the same rows as stage-01 table keys
            tk = kc.selection_keys(p, 0, n)
            txt = key_text(o, allw, k).split(b"\n")[:-1]
            assert [int(x) for x in tk[:200]] == [hast_amd.canon_kmer(t) for t in txt[:200]]
    o.ho_s00_free(c)


@pytest.mark.parametrize("k,table_mb,record_mb,switches", [
    (21, 64, 0, {}), (21, 256, 2, {}), (11, 64, 1, {}), (29, 64, 0, {}), (5, 64, 0, {}), (27, 1024, 0, {}), (31, 64, 0, {}),
    # K = 17 .. 21 (m = 16, W = 2 .. 6) go through k_kc_emit4 (four windows per lane); the other K and HAST_KC_EMIT=lanes through k_kc_count<W, EMIT>
    (17, 64, 0, {}), (18, 64, 1, {}), (19, 64, 0, {}), (20, 256, 2, {}),
    (21, 64, 0, {"HAST_KC_EMIT": "lanes"}), (21, 256, 2, {"HAST_KC_EMIT": "lanes", "HAST_KC_FRESH": "0"}),
    # a table cleared before it is counted into (default: declared empty, the first flush writes every slice), with records that find no room
    (21, 256, 2, {"HAST_KC_FRESH": "0"}), (19, 64, 1, {"HAST_KC_FRESH": "0"}),
    # one region per level-1 bin, and a number that does not divide anything
    (21, 64, 0, {"HAST_KC_L1_SPLIT": "1"}), (21, 256, 2, {"HAST_KC_L1_SPLIT": "3"}),
    # tiles of 1024 and 8192 window starts
    (21, 64, 0, {"HAST_KC_TILE": "1024"}), (20, 64, 1, {"HAST_KC_TILE": "8192"}),
])
def test_partitioned_counting_equals_oracle(built, oracle_lib, monkeypatch, k, table_mb, record_mb, switches):
    """The counting path of large tables (kc_kernels.hip "partitioned counting": windows written out as records of a minimizer run,
    partitioned by bucket range in two levels, counted per slice in LDS), forced on small ones: counts, histograms and selections ==
    oracle -- with an ample record buffer (one flush at sync), with a tiny one (HAST_KC_RECORD_MB: many flushes, records that find
    no room are counted on the spot, bins that overflow go through the spill list), for K up to 27 (one window per record: 2 K bits of bases + 4 bits for the place of its minimizer +
    6 bits of run length and parent); K = 29 and 31 have no room in a 64-bit record and stay with the atomic kernel.  Streams with every byte class, homopolymers (thousands of
    windows of one bucket: the spill path), both parents, a second round of counting after a read of the table."""
    o = oracle_lib
    rng = random.Random(900 + k + table_mb)
    pat, mat = random_stream(rng, 400_000, k), random_stream(rng, 300_000, k)
    shared = random_stream(rng, 150_000, k)
    more = random_stream(rng, 100_000, k)
    monkeypatch.setenv("HAST_KC_COUNT", "partition")
    monkeypatch.setenv("HAST_KC_FLUSH", "sweep")          # (few records for the table's size would take the atomic path where they lie)
    if record_mb:
        monkeypatch.setenv("HAST_KC_RECORD_MB", str(record_mb))
    for name, value in switches.items():
        monkeypatch.setenv(name, value)
    with KmerCounter(k, table_bytes=table_mb << 20) as kc:
        info = kc.partition_info()
        assert info["partitioned"] == (k <= 27), info
        kc.count(0, pat)
        kc.count(1, mat)
        for p in (0, 1):
            kc.count(p, shared)
        kc.sync()
        c = oracle_table(o, k, [(0, pat), (0, shared), (1, mat), (1, shared)])
        check_against_oracle(o, kc, c, k, None)
        for lo, hi in [(1, 1 << 30), (2, 5), (1, 1)]:
            assert [kc.select(p, lo, hi) for p in (0, 1)] == [oracle_select(o, c, p, lo, hi).size for p in (0, 1)], (lo, hi)
        # counting goes on after the table has been read (a flush happens before every read)
        kc.count(1, more)
        o.ho_s00_add_stream(c, 1, more.ctypes.data, more.size)
        check_against_oracle(o, kc, c, k, None)
        info = kc.partition_info()
        if k <= 27:
            assert info["flushes"] >= 2 and info["records"] > 0, info
            if record_mb:
                assert info["flushes"] >= 3, info
    o.ho_s00_free(c)


def test_partitioned_and_atomic_counting_agree_on_a_trio(built, monkeypatch):
    """the synthetic trio of the bench (reads with errors over two parental genomes) through both paths: same histograms, same
    selections, same distinct / total counts -- the partitioned path with its flushes swept through LDS, with every flush taken
    through the direct path where the records lie (what few records for the table's size do by themselves), and as a table of this
    size counts when nothing is forced"""
    g = hast_amd.KcSynth(0, 3_000_000, 150, 2, 20, 20)
    res = {}
    for mode in ("atomic", "partition", "partition/atomic-flush", "default"):
        monkeypatch.delenv("HAST_KC_COUNT", raising=False)
        monkeypatch.delenv("HAST_KC_FLUSH", raising=False)
        if mode != "default":
            monkeypatch.setenv("HAST_KC_COUNT", mode.split("/")[0])
        if mode == "partition":
            monkeypatch.setenv("HAST_KC_FLUSH", "sweep")
        if mode == "partition/atomic-flush":
            monkeypatch.setenv("HAST_KC_FLUSH", "atomic")
        with KmerCounter(21, table_bytes=1 << 30) as kc:
            assert kc.partition_info()["partitioned"] == (mode != "atomic")
            for parent in (0, 1):
                for first in (0, 150_000):
                    kc.count(parent, hast_amd.kc_synth_host(g, parent, first, 150_000))
            kc.sync()
            st = kc.stats()
            res[mode] = (st["distinct"], st["total"], st["keys"], [kc.histo(p) for p in (0, 1)], [kc.select(p, 5, 60) for p in (0, 1)])
    a = res["atomic"]
    for mode in ("partition", "partition/atomic-flush", "default"):
        b = res[mode]
        assert a[:3] == b[:3] and a[4] == b[4] and a[4][0] > 1000, mode
        for p in (0, 1):
            assert np.array_equal(a[3][p], b[3][p]), mode


def test_slices_partition_the_key_space(built, oracle_lib):
    o, k = oracle_lib, 21
    rng = random.Random(7)
    pat, mat = random_stream(rng, 400_000, k), random_stream(rng, 400_000, k)
    c = oracle_table(o, k, [(0, pat), (1, mat), (1, pat[:100_000])])
    want = [oracle_select(o, c, p, 1, 50) for p in (0, 1)]
    with KmerCounter(k, table_bytes=32 << 20) as kc:
        for n_slices in (3, 1):
            h = [np.zeros(hast_amd.KC_HISTO_HIGH + 2, dtype=np.uint64) for _ in (0, 1)]
            distinct = [0, 0]
            for s in range(n_slices):
                kc.set_slice(s, n_slices)
                kc.count(0, pat)
                kc.count(1, mat)
                kc.count(1, pat[:100_000])
                kc.sync()
                st = kc.stats()
                for p in (0, 1):
                    kc.histo(p, into=h[p])
                    distinct[p] += st["distinct"][p]
                    kc.select(p, 1, 50)
                if n_slices > 1:
                    assert 0 < st["keys"] < o.ho_s00_distinct(c, 0) + o.ho_s00_distinct(c, 1)
            for p in (0, 1):
                assert np.array_equal(h[p], oracle_histo(o, c, p))
                assert distinct[p] == o.ho_s00_distinct(c, p)
        kc.release_table()
        for p in (0, 1):                                           # both rounds appended: every key twice
            n = kc.selection_sort(p)
            assert kc.selection_text(p, 0, n) == key_text(o, np.repeat(want[p], 2), k)
    o.ho_s00_free(c)


def test_full_table_is_reported_and_high_load_is_exact(built, oracle_lib):
    o, k = oracle_lib, 21
    rng = random.Random(11)
    data = np.frombuffer(bytes(rng.choice(b"ACGT") for _ in range(200_000)), dtype=np.uint8).copy()     # ~200k distinct
    with KmerCounter(k, table_bytes=64 * 128) as kc:               # 512 slots
        kc.count(0, data)
        with pytest.raises(hast_amd.HastError) as ei:
            kc.sync()
        assert ei.value.status == 5
    c = oracle_table(o, k, [(0, data), (1, data[50_000:])])
    n_keys = o.ho_s00_distinct(c, 0)
    with KmerCounter(k, table_bytes=int(n_keys / 0.92) // 8 * 128) as kc:   # load factor 0.92: long overflow chains
        kc.count(0, data)
        kc.count(1, data[50_000:])
        kc.sync()
        st = kc.stats()
        assert st["keys"] == n_keys and st["keys"] / st["capacity"] > 0.9
        check_against_oracle(o, kc, c, k, None)
        assert kc.select(0, 1, 10) == oracle_select(o, c, 0, 1, 10).size
    o.ho_s00_free(c)


def test_streams_longer_than_a_staging_buffer_and_device_input(built, oracle_lib):
    """host streams are cut into 64-MB pieces (windows across a cut counted once); device-resident streams; the
    device generator equals the host generator"""
    o, k = oracle_lib, 21
    g = KcSynth(0, 300_000, 100, 12, 25, 30)
    n_reads = 700_000                                              # 70.7 MB per parent: two staging pieces
    host = [hast_amd.kc_synth_host(g, p, 0, n_reads) for p in (0, 1)]
    # make the cut fall inside a run of bases: one long record instead of reads around the 64-MB mark
    cut = (64 << 20) - 64
    host[0][cut - 3000:cut + 3000][host[0][cut - 3000:cut + 3000] == 10] = ord("A")
    c = oracle_table(o, k, [(0, host[0]), (1, host[1])])
    with hast_amd.Context(k) as ctx, KmerCounter(k, table_bytes=1 << 30) as kc:
        kc.count(0, host[0])
        d = ctx.alloc(host[1].size)
        kc.synth_device(g, 1, 0, n_reads, d)
        kc.count_device(1, d, host[1].size)
        kc.sync()
        assert np.array_equal(ctx.to_host(d, (host[1].size,), np.uint8), host[1])
        check_against_oracle(o, kc, c, k, None)
        h = kc.histo(0)
        lo, hi = hast_amd.kc_find_bounds(h)[2:]
        assert kc.select(0, lo, hi) == oracle_select(o, c, 0, lo, hi).size
        assert kc.select(1, 2, 10_000) == oracle_select(o, c, 1, 2, 10_000).size
    o.ho_s00_free(c)


def test_randomized_configurations_vs_oracle(built, oracle_lib):
    """fuzz: K, stream make-up, table size (load factor up to ~0.95), slices, call pattern; HAST_FUZZ_SEED / HAST_FUZZ_ITERS"""
    o = oracle_lib
    rng = random.Random(int(os.environ.get("HAST_FUZZ_SEED", "20261004")))
    for it in range(int(os.environ.get("HAST_FUZZ_ITERS", "24"))):
        k = rng.choice([1, 2, 7, 11, 15, 16, 17, 21, 21, 25, 31, 32])
        n = rng.choice([0, 1, k - 1, k, 1000, 40_000, 200_000])
        streams = [(p, random_stream(rng, max(n, 1), k)[:n]) for p in (0, 1, rng.randrange(2))]
        if rng.random() < 0.3 and n > 100:                          # a popular minimizer + a hot k-mer
            s = streams[0][1]
            s[: n // 2] = np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(it).integers(0, 4, n // 2)]
            s[np.arange(0, n // 2 - 40, 97)[:, None] + np.arange(20, 20 + rng.choice([16, 30]))] = ord("A")
        c = oracle_table(o, k, streams)
        n_keys = max(1, sum(1 for _ in range(1)) * (o.ho_s00_distinct(c, 0) + o.ho_s00_distinct(c, 1)))
        lf = rng.choice([0.05, 0.5, 0.9, 0.95])
        n_slices = rng.choice([1, 1, 2, 5])
        table_bytes = max(64 * 128, int(n_keys / lf / n_slices * 1.3) // 8 * 128)
        lo, hi = rng.choice([(1, 1 << 31), (1, 1), (2, 40), (3, 3)])
        with KmerCounter(k, table_bytes=table_bytes) as kc:
            h = [np.zeros(hast_amd.KC_HISTO_HIGH + 2, dtype=np.uint64) for _ in (0, 1)]
            try:
                for sl in range(n_slices):
                    kc.set_slice(sl, n_slices)
                    for p, data in streams:
                        if rng.random() < 0.5 and data.size > 10:   # one stream in two calls must be cut at a separator...
                            cut = int(np.argmax(data[data.size // 2:] == 10)) + data.size // 2
                            if data[cut] == 10:
                                kc.count(p, data[:cut + 1])
                                kc.count(p, data[cut + 1:])
                                continue
                        kc.count(p, data)
                    kc.sync()
                    for p in (0, 1):
                        kc.histo(p, into=h[p])
                        kc.select(p, lo, hi)
            except hast_amd.HastError as e:                         # uneven slices can overflow a tight table: that must be SAID
                assert e.status == 5, e
                continue
            kc.release_table()
            for p in (0, 1):
                assert np.array_equal(h[p], oracle_histo(o, c, p)), (it, k, p)
                want = oracle_select(o, c, p, lo, hi)
                nsel = kc.selection_sort(p)
                assert nsel == want.size, (it, k, p, lf, n_slices)
                assert kc.selection_text(p, 0, nsel) == key_text(o, want, k), (it, k, p)
        o.ho_s00_free(c)


def test_kmers_piled_on_one_minimizer(built, oracle_lib):
    """poly-A cores with random flanks: hundreds of thousands of distinct k-mers share the minimizer A^16"""
    import time
    o, k = oracle_lib, 21
    rng = np.random.default_rng(5)
    n, L = 150_000, 100
    reads = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L + 1))]
    reads[:, L] = 10
    reads[rng.random(n) < 0.4, 30:70] = ord("A")
    data = reads.reshape(-1).copy()
    c = oracle_table(o, k, [(0, data), (1, data[: data.size // 3])])
    with KmerCounter(k, table_bytes=1 << 30) as kc:
        t0 = time.time()
        kc.count(0, data)
        kc.count(1, data[: data.size // 3])
        kc.sync()
        dt = time.time() - t0
        check_against_oracle(o, kc, c, k, None)
        assert kc.select(0, 1, 3) == oracle_select(o, c, 0, 1, 3).size
    assert dt < 20, "piled-up k-mers: %.1f s" % dt
    o.ho_s00_free(c)


# ---- the program ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case,run", golden_cases("s00"))
def test_unshared_kmers_matches_reference_script_golden(built, golden_workdir, tmp_path, case, run):
    res = run_s00_case(hast_amd.unshared_kmers_exe(), golden_workdir, tmp_path, case, run, extra_args=["--table-gb", "0.25", "--stats"])
    assert b"[stats]" in res.stderr


def test_unshared_kmers_default_table_size(built, golden_workdir, tmp_path):
    """no --table-gb: the table takes 85 % of the free HBM"""
    run_s00_case(hast_amd.unshared_kmers_exe(), golden_workdir, tmp_path, "s00_fasta_k11", "ge2")


def test_unshared_kmers_slices_overflow_retry_and_table_handoff(built, oracle_dir, golden_workdir, tmp_path):
    """a table too small for the input: the program starts over with more slices and still writes the same sets;
    --save-table gives classify the same table as the text files do"""
    exe = hast_amd.unshared_kmers_exe()
    case, run = "s00_trio_k21", "auto"
    run_s00_case(exe, golden_workdir, tmp_path / "a", case, run, extra_args=["--table-gb", "0.0002", "--slices", "3"])
    res = run_s00_case(exe, golden_workdir, tmp_path / "b", case, run, extra_args=["--table-gb", "0.00005", "--save-table", "sets.hastkeys"])
    assert b"starting over" in res.stderr
    work = tmp_path / "b" / ("%s_%s" % (case, run))
    args = ["--read", "m.fq", "--read", "p1.fq"]
    a = subprocess.run([hast_amd.classify_exe(), "--hap0", "paternal.unique.filter.mer", "--hap1", "maternal.unique.filter.mer"] + args,
                       cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    b = subprocess.run([hast_amd.classify_exe(), "--load-table", "sets.hastkeys"] + args, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-500:], b.stderr[-500:])
    assert a.stdout == b.stdout and len(a.stdout) > 0


def _n_gpus():
    import ctypes
    try:
        n = ctypes.c_int(0)
        return n.value if ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


@pytest.mark.parametrize("devices", ["0,0", "0,0,0", pytest.param("0,1", marks=pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs"))])
def test_unshared_kmers_key_space_split_over_several_tables(built, golden_workdir, tmp_path, devices):
    """--devices a,b,..: one table per GPU, each owning a share of the minimizer space (here: several tables on the one
    GPU of the test box); with --slices on top and a table that overflows"""
    exe = hast_amd.unshared_kmers_exe()
    run_s00_case(exe, golden_workdir, tmp_path / "a", "s00_trio_k21", "auto", extra_args=["--table-gb", "0.05", "--devices", devices])
    run_s00_case(exe, golden_workdir, tmp_path / "b", "s00_gz_k25", "gz", extra_args=["--table-gb", "0.01", "--devices", devices, "--slices", "2"])
    res = run_s00_case(exe, golden_workdir, tmp_path / "c", "s00_trio_k21", "k17_bounds", extra_args=["--table-gb", "0.00003", "--devices", devices])
    assert b"starting over" in res.stderr


def test_unshared_kmers_gz_files_cut_inside_a_record(built, golden_workdir, tmp_path):
    """`zcat a b | counter` (build_unshared_kmers.sh:187-188): the gz files of a parent are one stream.  The program reads
    them in parallel and must notice when that is not the same thing: re-cut the golden case's files mid-record."""
    import gzip
    import shutil
    from tests.conftest import load_case
    case, run = "s00_gz_k25", "gz"
    src = golden_workdir / case
    work = tmp_path / "recut"
    shutil.copytree(src, work)
    for parent in "mp":
        # the script puts each new file in FRONT of the list (:105,109): the stream is b then a
        whole = gzip.open(work / ("%s_b.fq.gz" % parent)).read() + gzip.open(work / ("%s_a.fq.gz" % parent)).read()
        cut = len(whole) // 3 + (17 if parent == "m" else 140)                 # inside a sequence / a quality line
        for name, part in (("b", whole[:cut]), ("a", whole[cut:])):
            with gzip.GzipFile(work / ("%s_%s.fq.gz" % (parent, name)), "wb", mtime=0) as f:
                f.write(part)
    meta = load_case(case)["runs"][run]
    res = subprocess.run([hast_amd.unshared_kmers_exe()] + meta["argv"] + ["--table-gb", "0.1"], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert res.returncode == 0, res.stderr[-1000:]
    assert b"in order" in res.stderr
    for prod, rec in meta["products"].items():
        assert open(work / prod, "rb").read() == open(work / rec["expected"], "rb").read(), prod     # our rows are sorted already


def test_unshared_kmers_argument_checks(built, golden_workdir, tmp_path):
    exe = hast_amd.unshared_kmers_exe()
    d = golden_workdir / "s00_gz_k25"
    run = lambda *a: subprocess.run([exe] + list(a), cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run().returncode == 0 and b"Usage" in run().stdout                        # build_unshared_kmers.sh:57-60
    assert run("--help").returncode == 0
    assert run("--bogus").returncode == 0                                                # :113-116 (bare `exit`)
    assert run("--paternal", "p_a.fq.gz").returncode == 1                                # no maternal
    assert run("--paternal", "p_a.fq.gz", "--maternal", "m_a.fq.gz", "--mer", "10").returncode == 1
    assert run("--paternal", "p_a.fq.gz", "--maternal", "m_a.fq.gz", "--p-lower", "0").returncode == 1
    assert run("--paternal", "p_a.fq.gz", "--maternal", "nope.fq.gz").returncode == 1
    (d / "plain.fq").write_bytes(b"@r\nACGT\n+\nIIII\n")
    assert run("--paternal", "p_a.fq.gz", "--paternal", "plain.fq", "--maternal", "m_a.fq.gz").returncode == 1      # :166-185
    (d / "bad.fq").write_bytes(b"@r\nACGTACGTACGTACGTACGTACGTAAA\n+\nIII\n")
    r = run("--paternal", "bad.fq", "--maternal", "plain.fq", "--table-gb", "0.01")
    assert r.returncode == 1 and b"quality" in r.stdout

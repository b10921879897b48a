"""GPU parity tests (-m gpu): the HIP path through the C ABI (libhast.so) against the oracle
(CPU restatement of the reference, pinned by tests/test_oracle_golden.py).  Integer work: every
comparison is bit-exact."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

import hast_amd
from hast_amd.binding import make_params
from tests import oracle_binding as ob
from tests.conftest import golden_cases, load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(hast_amd.lib_path()):
        hast_amd.build()
    return hast_amd.lib()


def oracle_from_keys(o, k, keys0, keys1):
    oc = o.ho_new()
    for h, keys in ((0, keys0), (1, keys1)):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        assert o.ho_load_keys(oc, keys.ctypes.data, keys.size, h, k) == 0
    return oc


def oracle_counts(o, oc, bases, offsets, ids, n_bc, threads=4):
    e0, e1, eneg = (np.zeros(n_bc, np.uint32) for _ in range(3))
    o.ho_classify_ids(oc, bases.ctypes.data, offsets.ctypes.data, ids.ctypes.data, ids.size,
                      e0.ctypes.data, e1.ctypes.data, eneg.ctypes.data, None, threads)
    return e0, e1, eneg


def oracle_votes(o, oc, bases, offsets):
    n = offsets.size - 1
    out = np.zeros((n, 2), np.uint32)
    v0, v1, hn = C.c_uint32(), C.c_uint32(), C.c_int()
    raw = bases.tobytes()
    for i in range(n):
        s = raw[int(offsets[i]):int(offsets[i + 1])]
        o.ho_read_votes(oc, s, len(s), C.byref(v0), C.byref(v1), C.byref(hn))
        out[i] = (v0.value, v1.value)
    return out


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", [21, 32])
def test_table_build_sizes_lookup_erase(built, oracle_lib, k):
    rng = np.random.default_rng(1)
    mask = (1 << (2 * k)) - 1
    raw0 = rng.integers(0, mask, 40000, dtype=np.uint64)
    raw1 = rng.integers(0, mask, 40000, dtype=np.uint64)
    canon = lambda a: np.array([min(int(x), oracle_lib.ho_revcomp(int(x), k)) for x in a], dtype=np.uint64)
    k0, k1 = canon(raw0), canon(raw1)
    k1[:500] = k0[:500]                       # keys in both sets
    k0 = np.concatenate([k0, k0[:1000]])      # duplicates
    oc = oracle_from_keys(oracle_lib, k, k0, k1)
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(k0.size + k1.size)
        ctx.table_insert_keys(0, k0)
        ctx.table_insert_keys(1, k1)
        assert ctx.table_sizes() == (oracle_lib.ho_set_size(oc, 0), oracle_lib.ho_set_size(oc, 1))
        ctx.table_insert_keys(0, k0)          # idempotent
        assert ctx.table_sizes() == (oracle_lib.ho_set_size(oc, 0), oracle_lib.ho_set_size(oc, 1))
        probe = np.concatenate([k0[:3000], k1[:3000], canon(rng.integers(0, mask, 3000, dtype=np.uint64))])
        tags = ctx.table_lookup(probe)
        exp = np.array([oracle_lib.ho_contains(oc, 0, int(x)) | (oracle_lib.ho_contains(oc, 1, int(x)) << 1) for x in probe], np.uint8)
        assert np.array_equal(tags, exp)
        # erase (InitAdaptor semantics): both sets lose the key; report which had it
        victims = np.concatenate([k0[:10], k1[600:610], k0[100:105]])     # k0[:10] are in both sets
        hit = ctx.table_erase(victims)
        assert np.array_equal(hit, exp_hit(oracle_lib, oc, victims))
        assert ctx.table_sizes() == (oracle_lib.ho_set_size(oc, 0), oracle_lib.ho_set_size(oc, 1))
        assert not ctx.table_lookup(victims).any()
    oracle_lib.ho_free(oc)


def exp_hit(o, oc, victims):
    """what the reference's find-then-erase loop reports, replayed on the oracle's sets"""
    import ctypes
    out = []
    for v in victims:
        h = o.ho_contains(oc, 0, int(v)) | (o.ho_contains(oc, 1, int(v)) << 1)
        out.append(h)
        buf = ctypes.create_string_buffer(40)
        o.ho_kmer_to_str(int(v), o.ho_k(oc), buf)
        o.ho_init_adaptor(oc, buf.value, buf.value, None)     # erases exactly this canonical key
    return np.array(out, np.uint8)


def test_table_high_load_overflow_chains(built, oracle_lib):
    """Load factor 0.9 forces bucket-overflow chains through insert, lookup and classify."""
    k = 15
    n = 30000
    p = make_params(k, 100, n, 37)
    keys = [hast_amd.synth_keys_host(p, h, 0, n) for h in (0, 1)]
    oc = oracle_from_keys(oracle_lib, k, keys[0], keys[1])
    bases, ids = hast_amd.synth_reads_host(p, 0, 5000)
    off = np.arange(5001, dtype=np.uint64) * 100
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(int(oracle_lib.ho_set_size(oc, 0) + oracle_lib.ho_set_size(oc, 1)), 0.9)
        ctx.table_insert_keys(0, keys[0])
        ctx.table_insert_keys(1, keys[1])
        assert ctx.table_sizes() == (oracle_lib.ho_set_size(oc, 0), oracle_lib.ho_set_size(oc, 1))
        ctx.counts_resize(37)
        ctx.classify_batch(bases, off, ids, 100)
        got = ctx.counts_read(37)
    exp = oracle_counts(oracle_lib, oc, bases, off, ids, 37)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)
    oracle_lib.ho_free(oc)


def test_table_full_is_reported(built):
    with hast_amd.Context(21) as ctx:
        ctx.table_reserve(100, 0.9)                      # 64 buckets minimum = 512 slots
        keys = np.arange(1, 2000, dtype=np.uint64)       # small values are canonical enough to be distinct
        with pytest.raises(hast_amd.HastError) as ei:
            ctx.table_insert_keys(0, keys)
        assert ei.value.status == 5


def test_insert_text_matches_reference_set_sizes(built, oracle_lib, golden_workdir):
    for case in ("edge_k7", "rand_k21", "rand_k31", "rand_k32", "rand_k11"):
        d = golden_workdir / case
        t0, t1 = open(d / "hap0.mer", "rb").read(), open(d / "hap1.mer", "rb").read()
        k = t0.index(b"\n")
        oc = oracle_lib.ho_new()
        assert oracle_lib.ho_load_kmers_text(oc, t0, len(t0), 0) == 0
        assert oracle_lib.ho_load_kmers_text(oc, t1, len(t1), 1) == 0
        with hast_amd.Context(k) as ctx:
            ctx.table_reserve(len(t0) // (k + 1) + len(t1) // (k + 1) + 2)
            assert ctx.table_insert_text(0, t0) == oracle_lib.ho_lines_loaded(oc, 0)
            assert ctx.table_insert_text(1, t1) == oracle_lib.ho_lines_loaded(oc, 1)
            assert ctx.table_sizes() == (oracle_lib.ho_set_size(oc, 0), oracle_lib.ho_set_size(oc, 1))
        oracle_lib.ho_free(oc)


def test_insert_text_file_streams_the_same_table(built, oracle_lib, golden_workdir, tmp_path):
    """hast_table_insert_text_file (pread workers + pinned double buffer) == hast_table_insert_text on the file's bytes: the
    golden k-mer files, and a file of several pieces (3.3M lines) with an unterminated tail that must be dropped"""
    for case in ("edge_k7", "rand_k21", "rand_k31", "rand_k32"):
        d = golden_workdir / case
        t = [open(d / ("hap%d.mer" % h), "rb").read() for h in (0, 1)]
        k = t[0].index(b"\n")
        sizes, lines = [], []
        for from_file in (False, True):
            with hast_amd.Context(k) as ctx:
                ctx.table_reserve(len(t[0]) // (k + 1) + len(t[1]) // (k + 1) + 2)
                lines.append([ctx.table_insert_text_file(h, d / ("hap%d.mer" % h)) if from_file else ctx.table_insert_text(h, t[h]) for h in (0, 1)])
                sizes.append(ctx.table_sizes())
        assert sizes[0] == sizes[1] and lines[0] == lines[1], case
    k, n = 21, 3_300_000
    p = make_params(k, 150, n, 10)
    keys = hast_amd.synth_keys_host(p, 0, 0, n)
    codes = np.frombuffer(b"ACTG", np.uint8)                       # kmer.h:12 int2base
    shifts = (2 * (k - 1 - np.arange(k))).astype(np.uint64)
    text = np.full((n, k + 1), ord("\n"), np.uint8)
    text[:, :k] = codes[((keys[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.int64)]
    path = tmp_path / "big.mer"
    path.write_bytes(text.tobytes() + b"ACGTACGTAC")               # the tail has no newline: dropped (classify.cpp:41)
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(n + 2)
        assert ctx.table_insert_text_file(0, path) == n
        assert ctx.table_sizes() == (np.unique(keys).size, 0)
        assert np.array_equal(ctx.table_lookup(keys[::1000]), np.ones(keys[::1000].size, np.uint8))
    (tmp_path / "ragged.mer").write_bytes(b"ACGTA\nACG\nACGTACC\n")
    with hast_amd.Context(5) as ctx:
        ctx.table_reserve(100)
        with pytest.raises(hast_amd.HastError) as ei:
            ctx.table_insert_text_file(0, tmp_path / "ragged.mer")
        assert ei.value.status == 6
        with pytest.raises(hast_amd.HastError):
            ctx.table_insert_text_file(0, tmp_path / "no_such_file.mer")


def test_insert_text_acgt_check_is_optional(built, tmp_path):
    """stage 01 codes any byte (kmer.h:11); hast_ctx_set_text_check makes the device refuse lines that are not upper-case
    A/C/G/T (what the stage-03 string classifier needs), from memory and from a file"""
    text = b"ACGTA\nACGTn\nacgta\nRYACG\n"
    (tmp_path / "odd.mer").write_bytes(text)
    with hast_amd.Context(5) as ctx:
        ctx.table_reserve(100)
        assert ctx.table_insert_text(0, text) == 4
        assert ctx.table_insert_text_file(1, tmp_path / "odd.mer") == 4
        ctx.set_text_check(True)
        for call in (lambda: ctx.table_insert_text(0, text), lambda: ctx.table_insert_text_file(0, tmp_path / "odd.mer")):
            with pytest.raises(hast_amd.HastError) as ei:
                call()
            assert ei.value.status == 6
        assert ctx.table_insert_text(0, b"ACGTA\nTTTTT\n") == 2


def test_insert_text_rejects_ragged_lines(built):
    with hast_amd.Context(5) as ctx:
        ctx.table_reserve(100)
        with pytest.raises(hast_amd.HastError) as ei:
            ctx.table_insert_text(0, b"ACGTA\nACG\nACGTACC\n")
        assert ei.value.status == 6


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("clustered", [False, True])
def test_synth_generators_agree(built, clustered):
    p = make_params(21, 150, 5000, 100, clustered=clustered)
    with hast_amd.Context(21) as ctx:
        n = 3000
        d_b, d_i, d_k = ctx.alloc(n * 150), ctx.alloc(n * 4), ctx.alloc(n * 8)
        ctx.synth_reads_device(p, 12345, n, d_b, d_i)
        ctx.synth_keys_device(p, 1, 77, n, d_k)
        ctx.sync()
        hb, hi = hast_amd.synth_reads_host(p, 12345, n)
        assert np.array_equal(ctx.to_host(d_b, (n * 150,), np.uint8), hb)
        assert np.array_equal(ctx.to_host(d_i, (n,), np.uint32), hi)
        assert np.array_equal(ctx.to_host(d_k, (n,), np.uint64), hast_amd.synth_keys_host(p, 1, 77, n))


@pytest.mark.parametrize("k,L", [(21, 150), (31, 150), (32, 150), (11, 100), (5, 64), (27, 151), (1, 40)])
def test_classify_fixed_length_vs_oracle(built, oracle_lib, k, L):
    n_keys, n_reads, n_bc = 20000, 20011, 333
    if k <= 5:
        n_keys = 200
    p = make_params(k, L, n_keys, n_bc)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    oc = oracle_from_keys(oracle_lib, k, keys[0], keys[1])
    bases, ids = hast_amd.synth_reads_host(p, 0, n_reads)
    off = np.arange(n_reads + 1, dtype=np.uint64) * L
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(2 * n_keys)
        ctx.synth_table_build(p)
        assert ctx.table_sizes() == (oracle_lib.ho_set_size(oc, 0), oracle_lib.ho_set_size(oc, 1))
        ctx.counts_resize(n_bc)
        d_b, d_i, d_v = ctx.alloc(n_reads * L), ctx.alloc(n_reads * 4), ctx.alloc(n_reads * 8)
        ctx.synth_reads_device(p, 0, n_reads, d_b, d_i)
        ctx.classify_device(d_b, n_reads * L, n_reads, L, d_barcode_ids=d_i, d_votes=d_v)
        ctx.sync()
        got = ctx.counts_read(n_bc)
        votes = ctx.to_host(d_v, (n_reads, 2), np.uint32)
    exp = oracle_counts(oracle_lib, oc, bases, off, ids, n_bc)
    assert np.array_equal(votes[:2000], oracle_votes(oracle_lib, oc, bases[:2000 * L], off[:2001]))
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)
    assert int(exp[0].sum()) + int(exp[1].sum()) > 0
    oracle_lib.ho_free(oc)


@pytest.mark.parametrize("k,L,fm", [(21, 6000, 0), (31, 9000, 0), (21, 4097, 8), (21, 4096, 8)])
def test_classify_fixed_long_reads_without_offsets_vs_oracle(built, oracle_lib, k, L, fm):
    """Fixed-length reads longer than a kernel row's 4096 positions, handed over WITHOUT offsets (hast.h: d_offsets == NULL):
    they must take the segmented path like reads with offsets do -- the filter's t-mer order keeps a row position in 12 bits, so
    a longer row would sample another m-mer than the one a key was filed under and miss it.  Keys are the reads' own windows
    at positions >= 4000, so that almost all hits lie where positions would wrap; with and without 'N'; votes and counters ==
    oracle, for the filter (exact entries / prints) and for the table probed directly."""
    rng = np.random.default_rng(k * 1000 + L)
    n_reads, n_bc = 400, 37
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n_reads * L)].copy()
    rows = bases.reshape(n_reads, L)
    rows[5, 17] = ord("N")                       # whole-read skip (classify.cpp:190-193)
    rows[6, L - 1] = ord("N")
    rows[7, 4500 % L] = ord("n")                 # lower case is a base (kmer.h:11)
    keys = [[], []]
    for i in range(60):
        km = hast_amd.chop_read(rows[i, 4000:].tobytes(), k)
        keys[i & 1] += km[::2]
        keys[1 - (i & 1)] += km[5::11]           # some keys in both sets
    keys = [np.unique(np.array(x, dtype=np.uint64)) for x in keys]
    rows[100:160] = rows[0:60]                   # the key-bearing reads once more, further down the batch
    ids = rng.integers(0, n_bc, n_reads).astype(np.uint32)
    off = np.arange(n_reads + 1, dtype=np.uint64) * L
    oc = oracle_from_keys(oracle_lib, k, keys[0], keys[1])
    exp_votes = np.zeros((n_reads, 2), np.uint32)
    e = [np.zeros(n_bc, np.uint32) for _ in range(3)]
    oracle_lib.ho_classify_ids_votes(oc, bases.ctypes.data, off.ctypes.data, ids.ctypes.data, n_reads, e[0].ctypes.data,
                                     e[1].ctypes.data, e[2].ctypes.data, None, exp_votes.ctypes.data, 4)
    oracle_lib.ho_free(oc)
    assert int(exp_votes.sum()) > 30 * (L - 4000 - k) and int(e[2].sum()) >= 2
    for enable in (1, 2, 0):
        with hast_amd.Context(k) as ctx:
            ctx.set_filter(enable, fm if enable else 0)
            ctx.table_reserve(keys[0].size + keys[1].size)
            ctx.table_insert_keys(0, keys[0])
            ctx.table_insert_keys(1, keys[1])
            ctx.counts_resize(n_bc)
            d_b, d_i, d_v = ctx.to_device(bases), ctx.to_device(ids), ctx.alloc(n_reads * 8)
            ctx.classify_device(d_b, bases.size, n_reads, L, d_barcode_ids=d_i, d_votes=d_v)
            got = ctx.counts_read(n_bc)
            votes = ctx.to_host(d_v, (n_reads, 2), np.uint32)
        assert np.array_equal(votes, exp_votes), (k, L, enable)
        for a, b in zip(got, e):
            assert np.array_equal(a, b), (k, L, enable)


def ragged_reads(rng, k, keys, n, max_len):
    seqs = []
    for i in range(n):
        r = rng.random()
        if r < 0.02:
            L = 0
        elif r < 0.06:
            L = rng.randint(1, k - 1) if k > 1 else 1
        elif r < 0.10:
            L = k
        else:
            L = rng.randint(k, max_len)
        s = [rng.choice("ACGT") for _ in range(L)]
        if L >= k and rng.random() < 0.6:
            for _ in range(rng.randint(1, 3)):
                key = int(rng.choice(keys))
                km = "".join("ACTG"[(key >> (2 * (k - 1 - j))) & 3] for j in range(k))
                o = rng.randint(0, L - k)
                s[o:o + k] = km
        if L and rng.random() < 0.05:
            s[rng.randrange(L)] = "N"
        if L and rng.random() < 0.05:
            s[rng.randrange(L)] = "n"               # lower-case n is NOT a skip (kmer.h:11 -> G)
        if rng.random() < 0.05:
            s = [c.lower() for c in s]
        if L and rng.random() < 0.03:
            s[rng.randrange(L)] = rng.choice("RYKM*-.")
        seqs.append("".join(s).encode())
    return seqs


@pytest.mark.parametrize("k,max_len", [(21, 180), (31, 97), (32, 140), (7, 300), (13, 2500), (21, 70000), (32, 9000)])
def test_classify_ragged_reads_vs_oracle(built, oracle_lib, k, max_len):
    """Variable-length reads through offsets: empty, shorter than K (the reference aborts; we define
    0 windows), len==K, N / n / lower-case / IUPAC bytes, unaligned starts.  max_len > 4096 goes through the
    segmented path (whole-read N skip by a pre-pass, votes summed over segments)."""
    rng = random.Random(k * 1000 + max_len)
    n_keys, n_bc = 3000, 50
    p = make_params(k, 100, n_keys, n_bc)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    oc = oracle_from_keys(oracle_lib, k, keys[0], keys[1])
    seqs = ragged_reads(rng, k, np.concatenate(keys), 6000 if max_len < 1000 else 700 if max_len < 4096 else 160, max_len)
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    ids = np.array([rng.randrange(n_bc) for _ in seqs], dtype=np.uint32)
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(2 * n_keys)
        ctx.table_insert_keys(0, keys[0])
        ctx.table_insert_keys(1, keys[1])
        ctx.counts_resize(n_bc)
        d_b, d_o, d_i = ctx.to_device(bases), ctx.to_device(off), ctx.to_device(ids)
        d_v = ctx.alloc(len(seqs) * 8)
        ctx.classify_device(d_b, bases.size, len(seqs), int(lens.max()), d_offsets=d_o, d_barcode_ids=d_i, d_votes=d_v)
        ctx.sync()
        got = ctx.counts_read(n_bc)
        votes = ctx.to_host(d_v, (len(seqs), 2), np.uint32)
        # same reads through the host-buffer entry point (what the CLI uses), on top: counts double
        ctx.classify_batch(bases, off, ids, int(lens.max()))
        got2 = ctx.counts_read(n_bc)
    assert np.array_equal(votes, oracle_votes(oracle_lib, oc, bases, off))
    exp = oracle_counts(oracle_lib, oc, bases, off, ids, n_bc)
    for g, g2, e in zip(got, got2, exp):
        assert np.array_equal(g, e)
        assert np.array_equal(g2, 2 * e)                 # linearity
    oracle_lib.ho_free(oc)


def test_classify_votes_only_and_linearity(built, oracle_lib):
    """Per-read mode (no barcodes) and additivity over batches: counts(A)+counts(B) == counts(A u B)."""
    k, L, n_keys, n_bc, n = 21, 150, 10000, 97, 30000
    p = make_params(k, L, n_keys, n_bc)
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(2 * n_keys)
        ctx.synth_table_build(p)
        d_b, d_i, d_v = ctx.alloc(n * L), ctx.alloc(n * 4), ctx.alloc(n * 8)
        ctx.synth_reads_device(p, 5000, n, d_b, d_i)
        ctx.counts_resize(n_bc)
        ctx.classify_device(d_b, n * L, n, L, d_barcode_ids=d_i)
        whole = ctx.counts_read(n_bc)
        ctx.counts_zero()
        h = 12345
        ctx.classify_device(d_b, n * L, h, L, d_barcode_ids=d_i)
        ctx.classify_device(d_b + h * L, (n - h) * L, n - h, L, d_barcode_ids=d_i + 4 * h)
        parts = ctx.counts_read(n_bc)
        ctx.counts_zero()
        ctx.classify_device(d_b, n * L, n, L, d_votes=d_v)          # votes only: counters untouched
        ctx.sync()
        untouched = ctx.counts_read(n_bc)
        votes = ctx.to_host(d_v, (n, 2), np.uint32)
        ids = ctx.to_host(d_i, (n,), np.uint32)
    for w, q, u in zip(whole, parts, untouched):
        assert np.array_equal(w, q)
        assert not u.any()
    assert np.array_equal(np.bincount(ids, weights=votes[:, 0], minlength=n_bc).astype(np.uint32), whole[0])
    assert np.array_equal(np.bincount(ids, weights=votes[:, 1], minlength=n_bc).astype(np.uint32), whole[1])
    assert np.array_equal(np.bincount(ids, weights=(votes.sum(1) == 0), minlength=n_bc).astype(np.uint32), whole[2])


@pytest.mark.parametrize("k,m", [(21, 1), (21, 5), (21, 11), (21, 16), (21, 20), (21, 21), (31, 9), (31, 31), (32, 32), (32, 20), (12, 7)])
def test_minimizer_length_does_not_change_results(built, oracle_lib, k, m):
    """Bucket placement by minimizer (any m in [1,K]) is invisible in the results: table sizes, lookups,
    per-read votes and per-barcode counts stay bit-exact against the oracle; ragged reads included."""
    rng = random.Random(k * 100 + m)
    n_keys, n_bc = 6000, 41
    p = make_params(k, 100, n_keys, n_bc)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    oc = oracle_from_keys(oracle_lib, k, keys[0], keys[1])
    seqs = ragged_reads(rng, k, np.concatenate(keys), 3000, 160)
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    ids = np.array([rng.randrange(n_bc) for _ in seqs], dtype=np.uint32)
    with hast_amd.Context(k, minimizer=m) as ctx:
        assert ctx.minimizer == m
        ctx.table_reserve(2 * n_keys, 0.6)
        ctx.table_insert_keys(0, keys[0])
        ctx.table_insert_keys(1, keys[1])
        assert hast_amd.lib().hast_ctx_set_minimizer(ctx._h, k) != 0       # fixed once the table exists
        assert ctx.table_sizes() == (oracle_lib.ho_set_size(oc, 0), oracle_lib.ho_set_size(oc, 1))
        probe = np.concatenate([keys[0][:500], keys[1][:500]])
        assert (ctx.table_lookup(probe) != 0).all()
        ctx.counts_resize(n_bc)
        d_b, d_o, d_i = ctx.to_device(bases), ctx.to_device(off), ctx.to_device(ids)
        d_v = ctx.alloc(len(seqs) * 8)
        ctx.classify_device(d_b, bases.size, len(seqs), int(lens.max()), d_offsets=d_o, d_barcode_ids=d_i, d_votes=d_v)
        ctx.sync()
        got = ctx.counts_read(n_bc)
        votes = ctx.to_host(d_v, (len(seqs), 2), np.uint32)
    assert np.array_equal(votes, oracle_votes(oracle_lib, oc, bases, off))
    exp = oracle_counts(oracle_lib, oc, bases, off, ids, n_bc)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)
    oracle_lib.ho_free(oc)


# ---- per-read mode (BASELINE config 5; secondary oracle = stage-03 per-read classifier) ----------------
def _s03_oracle(o, lines0, lines1):
    oc = o.ho_s03_new()
    for h, lines in ((0, lines0), (1, lines1)):
        t = ("\n".join(lines) + "\n").encode()
        assert o.ho_s03_load_text(oc, t, len(t), h) == 0
    return oc


def _kmer_str(key, k):
    return "".join("ACTG"[(int(key) >> (2 * (k - 1 - j))) & 3] for j in range(k))


@pytest.mark.parametrize("k,max_len,fm", [(21, 3000, 0), (31, 30000, 0), (32, 2000, 0), (11, 600, 0), (27, 100, 0),
                                          (21, 30000, 14), (15, 3000, 8), (12, 700, 12),        # fm: exact filter entries
                                          (31, 30000, 15)])     # config 5's geometry (m = 15, t = 6, kp = 23, prints filed once): the kernel with that geometry compiled in
def test_perread_strict_mode_vs_s03_oracle(built, oracle_lib, k, max_len, fm):
    """Integer hits per read == the stage-03 reference's string lookups: windows containing N / lower-case /
    IUPAC bytes miss (no whole-read skip), reads far longer than an LDS row are segmented on the device."""
    rng = random.Random(k + max_len)
    n_keys = 2000
    p = make_params(k, 100, n_keys, 1)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    lines = [[_kmer_str(x, k) for x in keys[h]] for h in (0, 1)]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    lines[0][:50] = ["".join(comp[c] for c in reversed(s)) for s in lines[0][:50]]   # files need not be canonical
    oc = _s03_oracle(oracle_lib, lines[0], lines[1])
    allk = lines[0] + lines[1]
    seqs = []
    for i in range(300 if max_len < 5000 else 60):
        L = rng.choice([0, 1, k - 1, k, k + 1, 100, max_len // 3, max_len]) if i > 3 else max_len
        s = [rng.choice("ACGT") for _ in range(L)]
        for _ in range(rng.randint(0, 1 + L // 150)):
            if L >= k:
                km = rng.choice(allk)
                o = rng.randint(0, L - k)
                s[o:o + k] = km if rng.random() < 0.5 else "".join(comp[c] for c in reversed(km))
        for _ in range(rng.randint(0, 1 + L // 400)):
            if L:
                s[rng.randrange(L)] = rng.choice("NnacgtRYK-")
        seqs.append("".join(s).encode())
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    with hast_amd.Context(k) as ctx:
        if fm:
            ctx.set_filter(1, fm)
        ctx.table_reserve(2 * n_keys)
        ctx.table_insert_text(0, ("\n".join(lines[0]) + "\n").encode())
        ctx.table_insert_text(1, ("\n".join(lines[1]) + "\n").encode())
        votes = ctx.classify_perread(bases, off)
        votes2 = ctx.classify_perread(bases, off)                 # idempotent: the output rows are overwritten
        assert not fm or ctx.filter_mode() == (1 if (k, fm) == (31, 15) else 2)
        if (k, fm) == (31, 15):
            assert ctx.filter_info()[1:4] == (15, 6, 23)
    exp = np.zeros_like(votes)
    h0, h1 = C.c_uint32(), C.c_uint32()
    for i, sq in enumerate(seqs):
        oracle_lib.ho_s03_read_hits(oc, sq, len(sq), C.byref(h0), C.byref(h1))
        exp[i] = (h0.value, h1.value)
    oracle_lib.ho_s03_free(oc)
    assert np.array_equal(votes, exp)
    assert np.array_equal(votes2, exp)
    assert exp.sum() > 0


@pytest.mark.parametrize("case", ["s03_k21", "s03_k31"])
def test_perread_rows_match_s03_reference_golden(built, oracle_lib, golden_workdir, case):
    """Hits from the GPU, formatted as the stage-03 program prints them, reproduce the real reference binary's
    stdout for the committed FASTA input (multi-line records, ids, ambiguous/haplotype calls, %0.6f densities)."""
    d = golden_workdir / case
    t0, t1 = open(d / "hap0.mer", "rb").read(), open(d / "hap1.mer", "rb").read()
    k = t0.index(b"\n")
    oc = oracle_lib.ho_s03_new()
    assert oracle_lib.ho_s03_load_text(oc, t0, len(t0), 0) == 0 and oracle_lib.ho_s03_load_text(oc, t1, len(t1), 1) == 0
    names, seqs = [], []
    for line in open(d / "reads.fa", "rb").read().split(b"\n"):
        if not line:
            continue
        if line.startswith(b">"):
            names.append(line[1:])
            seqs.append(b"")
        else:
            seqs[-1] += line
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(len(t0) // (k + 1) + len(t1) // (k + 1) + 2)
        ctx.table_insert_text(0, t0)
        # the library follows load_kmers' framing: the unterminated tail of hap1.mer is dropped
        ctx.table_insert_text(1, t1[:t1.rindex(b"\n") + 1])
        votes = ctx.classify_perread(bases, off)
    out = b""
    buf = C.create_string_buffer(4096)
    for name, v in zip(names, votes):
        n = oracle_lib.ho_s03_format_row(oc, name, len(name), int(v[0]), int(v[1]), buf)
        out += buf.raw[:n]
    oracle_lib.ho_s03_free(oc)
    assert out == open(d / "expected.fasta.tsv", "rb").read()


def test_counts_allreduce_rccl_path_single_rank(built, monkeypatch):
    """hast_counts_allreduce over RCCL with a 1-rank communicator (HAST_FORCE_RCCL=1): checks dlopen of librccl, the
    symbols and the ncclUint32/ncclSum enum values used by the N>1 merge; the sum over one rank is the identity."""
    monkeypatch.setenv("HAST_FORCE_RCCL", "1")
    k, L, n_keys, n_bc, n = 21, 150, 5000, 64, 4000
    p = make_params(k, L, n_keys, n_bc)
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(2 * n_keys)
        ctx.synth_table_build(p)
        ctx.counts_resize(n_bc)
        d_b, d_i = ctx.alloc(n * L), ctx.alloc(n * 4)
        ctx.synth_reads_device(p, 0, n, d_b, d_i)
        ctx.classify_device(d_b, n * L, n, L, d_barcode_ids=d_i)
        before = ctx.counts_read(n_bc)
        arr = (C.c_void_p * 1)(ctx._h)
        st = hast_amd.lib().hast_counts_allreduce(arr, 1)
        assert st == 0, hast_amd.lib().hast_last_error()
        after = ctx.counts_read(n_bc)
    for a, b in zip(before, after):
        assert np.array_equal(a, b)
    assert int(before[0].sum()) > 0


@pytest.mark.parametrize("k,n_keys,L,n_reads", [(21, 5_000_000, 150, 300_000), (31, 5_000_000, 5000, 6_000)])
def test_tables_placement_invariance(built, k, n_keys, L, n_reads):
    """Results must not depend on where keys live.  The same reads are classified against a table placed by the default
    minimizer and against one placed by plain hashing of the key (m = K), in one batch and split in two; set sizes,
    per-read votes and per-barcode counts must be identical, and hits must exist.  (A self-comparison; the BASELINE-size
    tables -- slot indices beyond 2^32 bytes and 2^31 slots -- are compared with the oracle's full-size sets in
    test_baseline_size_configs_vs_oracle / test_baseline_config5_size_vs_oracle.)"""
    n_bc = 100_000
    p = make_params(k, L, n_keys, n_bc)
    results = []
    for m in (None, k):
        with hast_amd.Context(k, minimizer=m) as ctx:
            ctx.table_reserve(2 * n_keys)
            ctx.synth_table_build(p)
            sizes = ctx.table_sizes()
            ctx.counts_resize(n_bc)
            d_b, d_i, d_v = ctx.alloc(n_reads * L), ctx.alloc(n_reads * 4), ctx.alloc(n_reads * 8)
            ctx.synth_reads_device(p, 777, n_reads, d_b, d_i)
            ctx.classify_device(d_b, n_reads * L, n_reads, L, d_barcode_ids=d_i, d_votes=d_v)
            whole = ctx.counts_read(n_bc)
            votes = ctx.to_host(d_v, (n_reads, 2), np.uint32)
            ctx.counts_zero()
            h = n_reads // 3
            ctx.classify_device(d_b, n_reads * L, h, L, d_barcode_ids=d_i)
            ctx.classify_device(d_b + h * L, (n_reads - h) * L, n_reads - h, L, d_barcode_ids=d_i + 4 * h)
            parts = ctx.counts_read(n_bc)
            for w, q in zip(whole, parts):
                assert np.array_equal(w, q)
            ids = ctx.to_host(d_i, (n_reads,), np.uint32)
            for f in (d_b, d_i, d_v):
                ctx.free(f)
        assert np.array_equal(np.bincount(ids, weights=votes[:, 0], minlength=n_bc).astype(np.uint32), whole[0])
        results.append((sizes, votes, whole))
    (s0, v0, w0), (s1, v1, w1) = results
    assert s0 == s1 and abs(s0[0] - n_keys) < n_keys // 1000
    assert np.array_equal(v0, v1)
    for a, b in zip(w0, w1):
        assert np.array_equal(a, b)
    assert int(v0.sum()) > n_reads // 2


@pytest.mark.parametrize("k,fm,ft,fkp,n_keys,lf", [
    (21, 0, 0, 0, 30000, 0.2),        # geometry by key count
    (21, 8, 0, 0, 400000, 0.2),       # 4^8 blocks for 800k prints: most sub-buckets FULL -> everything is verified in the table
    (21, 14, 6, 21, 30000, 0.85),     # the BASELINE geometry (mod-minimizer, W=8, t=6) on a small table with overflow chains
    (21, 13, 4, 0, 30000, 0.2),       # the C2 geometry (W=9, t=4)
    (21, 12, 3, 0, 30000, 0.2),       # t not congruent to m: still exact, only denser
    (31, 14, 0, 0, 30000, 0.2),       # long windows sampled on their first 22 bases
    (31, 14, 14, 31, 30000, 0.2),     # plain forward minimizer over the whole window, W=18
    (32, 10, 5, 27, 30000, 0.2),      # wide keys
    (9, 9, 9, 0, 3000, 0.2), (6, 3, 2, 0, 1500, 0.2), (2, 2, 1, 0, 10, 0.2), (1, 1, 1, 0, 2, 0.2),
    # geometries whose entries are EXACT codes (2(K-m) + log2 W <= 17 bits): a match in the filter is the hit
    (15, 8, 0, 0, 30000, 0.2),        # W = 8, 14 + 3 bits, like the BASELINE geometry, on a 8-MB filter
    (11, 4, 0, 0, 60000, 0.2),        # 256 blocks for 240k strings: every sub-bucket FULL -> matches found, misses verified
    (20, 14, 0, 0, 30000, 0.2),       # W = 7: positions 0..6 in 3 bits
    (17, 12, 0, 14, 30000, 0.2),      # sampling on a prefix (kp < K): the bases behind it are part of the code all the same
    (13, 13, 0, 0, 20000, 0.2),       # W = 1, no bases outside the m-mer: the entry is the tag bits alone
])
def test_filter_geometries_vs_oracle_and_exact_table(built, oracle_lib, k, fm, ft, fkp, n_keys, lf):
    """The fingerprint filter may only cost time: for every geometry (incl. overfull filters, t-mers that do not fit the
    mod-minimizer rule, plain minimizers, W = 1) per-read votes and per-barcode counts must equal the oracle's, and equal
    what the exact table gives when probed directly (hast_ctx_set_filter(enable = 0))."""
    L, n_bc, n_reads = 150 if k <= 21 else 400, 500, 20000
    p = make_params(k, L, n_keys, n_bc)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    bases, ids = hast_amd.synth_reads_host(p, 5, n_reads)
    # ragged lengths incl. reads shorter than K, and a few reads with N / lower case
    rng = np.random.default_rng(k * 131 + fm)
    lens = rng.integers(max(1, k - 2), L + 1, n_reads).astype(np.uint64)
    lens[::7] = L
    off = np.zeros(n_reads + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    rag = np.concatenate([bases[i * L:i * L + int(lens[i])] for i in range(n_reads)])
    for i in range(0, n_reads, 97):
        rag[int(off[i])] |= 0x20
    oc = oracle_from_keys(oracle_lib, k, keys[0], keys[1])
    exp_votes = np.zeros((n_reads, 2), np.uint32)
    e = [np.zeros(n_bc, np.uint32) for _ in range(3)]
    oracle_lib.ho_classify_ids_votes(oc, rag.ctypes.data, off.ctypes.data, ids.ctypes.data, n_reads, e[0].ctypes.data,
                                     e[1].ctypes.data, e[2].ctypes.data, None, exp_votes.ctypes.data, 4)
    oracle_lib.ho_free(oc)
    modes = []
    for enable in (1, 2, 0):          # filter (exact entries where they fit) / filter with prints always / the table directly
        with hast_amd.Context(k) as ctx:
            ctx.set_filter(enable, fm if enable else 0, ft if enable else 0, fkp if enable else 0)
            ctx.table_reserve(2 * n_keys, lf)
            ctx.table_insert_keys(0, keys[0])
            ctx.table_insert_keys(1, keys[1])
            ctx.counts_resize(n_bc)
            d_b, d_o, d_i, d_v = ctx.to_device(rag), ctx.to_device(off), ctx.to_device(ids), ctx.alloc(n_reads * 8)
            ctx.classify_device(d_b, rag.size, n_reads, L, d_offsets=d_o, d_barcode_ids=d_i, d_votes=d_v)
            got = ctx.counts_read(n_bc)
            votes = ctx.to_host(d_v, (n_reads, 2), np.uint32)
            en, m, t, kp, nbytes = ctx.filter_info()
            assert en == bool(enable)
            modes.append(ctx.filter_mode())
            if enable:
                assert nbytes == 128 * 4 ** m and 1 <= t <= m <= min(k, 15) and m <= kp <= k
                if fkp:
                    assert kp == fkp
                if fm:
                    assert m == fm
                if ft:
                    assert t == ft
        assert np.array_equal(votes, exp_votes), (k, fm, ft, enable)
        for a, b in zip(got, e):
            assert np.array_equal(a, b), (k, fm, ft, enable)
    assert k < 6 or int(exp_votes.sum()) > 0
    # which geometries hold exact entries: 2(K-m) + ceil(log2 W) <= 17 with the m and kp the library chose
    assert modes[1:] == [1, 0] and modes[0] in (1, 2)
    if fm and k < 32:
        w = (fkp or min(k, fm + 8)) - fm + 1
        assert (modes[0] == 2) == (2 * (k - fm) + (w - 1).bit_length() <= 17), (k, fm, fkp, modes)


@pytest.mark.parametrize("k,fm,mode", [(21, 0, 1), (15, 8, 2)])
def test_filter_follows_the_table(built, oracle_lib, k, fm, mode):
    """Keys added after a classification must be seen by the next one (the filter is rebuilt), erased keys must stop
    counting (with prints the table decides; exact entries -- mode 2 -- answer on their own, so an erase must reach the
    filter), and a second table in the same context must not see the first one's entries."""
    L, n_keys, n_bc, n_reads = 150, 20000, 50, 8000
    p = make_params(k, L, n_keys, n_bc)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    bases, ids = hast_amd.synth_reads_host(p, 0, n_reads)
    off = np.arange(n_reads + 1, dtype=np.uint64) * L

    def expect(k0, k1):
        oc = oracle_from_keys(oracle_lib, k, k0, k1)
        r = oracle_counts(oracle_lib, oc, bases, off, ids, n_bc)
        oracle_lib.ho_free(oc)
        return r

    with hast_amd.Context(k) as ctx:
        ctx.set_filter(1, fm)
        ctx.table_reserve(2 * n_keys)
        ctx.counts_resize(n_bc)
        d_b, d_i = ctx.to_device(bases), ctx.to_device(ids)

        def run():
            ctx.counts_zero()
            ctx.classify_device(d_b, bases.size, n_reads, L, d_barcode_ids=d_i)
            assert ctx.filter_mode() == mode
            return ctx.counts_read(n_bc)

        ctx.table_insert_keys(0, keys[0][:n_keys // 2])
        for a, b in zip(run(), expect(keys[0][:n_keys // 2], keys[1][:0])):
            assert np.array_equal(a, b)
        ctx.table_insert_keys(0, keys[0][n_keys // 2:])                 # more keys: the filter must follow
        ctx.table_insert_keys(1, keys[1])
        full = expect(keys[0], keys[1])
        for a, b in zip(run(), full):
            assert np.array_equal(a, b)
        assert int(full[0].sum()) > 0 and int(full[1].sum()) > 0
        gone = keys[1][:n_keys // 2]                                    # erased keys: the table says no (InitAdaptor removes
        ctx.table_erase(gone)                                           # a key from BOTH sets, classify.cpp:314-339)
        for a, b in zip(run(), expect(keys[0][~np.isin(keys[0], gone)], keys[1][~np.isin(keys[1], gone)])):
            assert np.array_equal(a, b)
        ctx.table_reserve(2 * n_keys)                                   # a new, empty table
        ctx.table_insert_keys(1, keys[1][:100])
        for a, b in zip(run(), expect(keys[0][:0], keys[1][:100])):
            assert np.array_equal(a, b)


ADAPTOR_F = b"CTGTCTCTTATACACATCTTAGGAAGACAAGCACTGACGACATGA"   # classify.cpp:312
ADAPTOR_R = b"TCTGCTGAGTCGAGAACGTCTCTGTGAGCCAAGGAGTTGCTCTGG"   # classify.cpp:313


@pytest.mark.heavy
@pytest.mark.parametrize("name,n_keys,n_bc,n_reads,clustered", [
    ("C2", 50_000_000, 1_000_000, 1_000_000, False),
    ("C3", 200_000_000, 10_000_000, 2_000_000, False),
    ("C3-clustered-keys", 200_000_000, 10_000_000, 500_000, True),
])
def test_baseline_size_configs_vs_oracle(built, oracle_lib, name, n_keys, n_bc, n_reads, clustered):
    """BASELINE configs 2 and 3 at their full table and barcode sizes (50M+50M keys / 1M barcodes; 200M+200M keys / 10M
    barcodes, K=21, 150-bp reads): the GPU classifies n_reads synthetic reads against the full-size merged table and the
    CPU oracle classifies the SAME reads against its own two full-size sets built from the same keys (load_kmers +
    InitAdaptor, classify.cpp:30-46,314-339; process_reads :186-209).  Set sizes after the adaptor scrub, per-read
    votes and all n_bc per-barcode (c0, c1, neg) counters must be identical."""
    k, L = 21, 150
    threads = len(os.sched_getaffinity(0))
    p = make_params(k, L, n_keys, n_bc, clustered=clustered)
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(2 * n_keys, 0.2)
        ctx.synth_table_build(p)
        akeys = []
        for ad in (ADAPTOR_F, ADAPTOR_R):
            for km in hast_amd.chop_read(ad, k):
                if km not in akeys:
                    akeys.append(km)
        ctx.table_erase(np.array(akeys, dtype=np.uint64))
        sizes = ctx.table_sizes()
        ctx.counts_resize(n_bc)
        d_b, d_i, d_v = ctx.alloc(n_reads * L + 64), ctx.alloc(n_reads * 4), ctx.alloc(n_reads * 8)
        first = 123_456_789
        ctx.synth_reads_device(p, first, n_reads, d_b, d_i)
        ctx.classify_device(d_b, n_reads * L, n_reads, L, d_barcode_ids=d_i, d_votes=d_v)
        got = ctx.counts_read(n_bc)
        votes = ctx.to_host(d_v, (n_reads, 2), np.uint32)
        bases = ctx.to_host(d_b, (n_reads * L,), np.uint8)
        ids = ctx.to_host(d_i, (n_reads,), np.uint32)
        # the oracle's sets from the same keys (device generator == host generator is a test of its own)
        oc = oracle_lib.ho_new()
        d_k = ctx.alloc(n_keys * 8)
        for h in (0, 1):
            ctx.synth_keys_device(p, h, 0, n_keys, d_k)
            ctx.sync()
            keys = ctx.to_host(d_k, (n_keys,), np.uint64)
            assert oracle_lib.ho_load_keys_mt(oc, keys.ctypes.data, keys.size, h, k, threads) == 0
            del keys
        for f in (d_b, d_i, d_v, d_k):
            ctx.free(f)
    oracle_lib.ho_init_adaptor(oc, ADAPTOR_F, ADAPTOR_R, None)
    assert sizes == (oracle_lib.ho_set_size(oc, 0), oracle_lib.ho_set_size(oc, 1)), name
    off = np.arange(n_reads + 1, dtype=np.uint64) * L
    exp_votes = np.zeros((n_reads, 2), np.uint32)
    e = [np.zeros(n_bc, np.uint32) for _ in range(3)]
    oracle_lib.ho_classify_ids_votes(oc, bases.ctypes.data, off.ctypes.data, ids.ctypes.data, n_reads, e[0].ctypes.data,
                                     e[1].ctypes.data, e[2].ctypes.data, None, exp_votes.ctypes.data, threads)
    oracle_lib.ho_free(oc)
    assert np.array_equal(votes, exp_votes), name
    for a, b in zip(got, e):
        assert np.array_equal(a, b), name
    assert int(e[0].sum()) + int(e[1].sum()) > n_reads // 2 and int(e[2].sum()) > 0


@pytest.mark.heavy
def test_baseline_config5_size_vs_oracle(built, oracle_lib):
    """BASELINE config 5 at its full table size on one GPU: K=31, 400M + 400M keys, PacBio-style 20-kb reads, barcode-free
    per-read (hits0, hits1) (S03/src_main/classify.cpp:203-218).  The stage-03 oracle keeps its keys as strings (800M of
    them do not fit a test); on reads without 'N' its hits are the stage-01 votes -- the number of windows whose canonical
    k-mer is in set h (classify.cpp:194-206) -- so the integer oracle with its two full-size sets is the checker for those
    reads, and a read WITH an 'N' must count exactly the windows that do not hold it (checked on the windows' own keys)."""
    k, L, n_keys, n_reads = 31, 20000, 400_000_000, 1500
    threads = len(os.sched_getaffinity(0))
    p = make_params(k, L, n_keys, 1)
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(2 * n_keys, 0.2)
        ctx.synth_table_build(p)
        sizes = ctx.table_sizes()
        d_b, d_v = ctx.alloc(n_reads * L + 64), ctx.alloc(n_reads * 8)
        d_o = ctx.to_device(np.arange(n_reads + 1, dtype=np.uint64) * L)
        ctx.synth_reads_device(p, 77_000_000, n_reads, d_b, 0)
        ctx.classify_perread_device(d_b, n_reads * L, d_o, n_reads, d_v)
        ctx.sync()
        votes = ctx.to_host(d_v, (n_reads, 2), np.uint32)
        bases = ctx.to_host(d_b, (n_reads * L,), np.uint8)
        assert ctx.filter_mode() == 1                                  # K - m = 16 bases do not fit an exact entry: prints
        assert ctx.filter_info()[1] in (14, 15)                        # 4^15 blocks (137 GB), or 4^14 when HBM is short
        oc = oracle_lib.ho_new()
        d_k = ctx.alloc(n_keys * 8)
        for h in (0, 1):
            ctx.synth_keys_device(p, h, 0, n_keys, d_k)
            ctx.sync()
            keys = ctx.to_host(d_k, (n_keys,), np.uint64)
            assert oracle_lib.ho_load_keys_mt(oc, keys.ctypes.data, keys.size, h, k, threads) == 0
            del keys
        for f in (d_b, d_v, d_k, d_o):
            ctx.free(f)
    assert sizes == (oracle_lib.ho_set_size(oc, 0), oracle_lib.ho_set_size(oc, 1))
    # reads with an 'N': the oracle sees them with the N replaced (a window over the N then matches nothing it should not:
    # the replacement bases are part of no planted k-mer), cut at the N into two reads whose windows are exactly the
    # windows that do not hold it
    rows = bases.reshape(n_reads, L)
    has_n = (rows == ord("N")).any(axis=1)
    pieces, owner = [], []
    for i in range(n_reads):
        if not has_n[i]:
            pieces.append(rows[i])
            owner.append(i)
        else:
            cut = np.flatnonzero(rows[i] == ord("N"))
            assert cut.size == 1
            for part in (rows[i][:cut[0]], rows[i][cut[0] + 1:]):
                if part.size >= k:
                    pieces.append(part)
                    owner.append(i)
    lens = np.array([x.size for x in pieces], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    flat = np.ascontiguousarray(np.concatenate(pieces))
    ids = np.zeros(len(pieces), np.uint32)
    pv = np.zeros((len(pieces), 2), np.uint32)
    e = [np.zeros(1, np.uint32) for _ in range(3)]
    oracle_lib.ho_classify_ids_votes(oc, flat.ctypes.data, off.ctypes.data, ids.ctypes.data, len(pieces), e[0].ctypes.data,
                                     e[1].ctypes.data, e[2].ctypes.data, None, pv.ctypes.data, threads)
    oracle_lib.ho_free(oc)
    exp = np.zeros((n_reads, 2), np.uint32)
    np.add.at(exp, np.array(owner), pv)
    assert np.array_equal(votes, exp)
    assert int(exp.sum()) > n_reads // 2 and int(has_n.sum()) > 0


def _n_gpus():
    import ctypes
    try:
        n = ctypes.c_int(0)
        hip = ctypes.CDLL("libamdhip64.so")
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs (RCCL over xGMI between two contexts of one process)")
def test_counts_allreduce_two_devices(built):
    """collectBarcodes / BarcodeCache::Add (classify.cpp:57-63,226-229) across two GPUs of one process: each context
    classifies its half of the reads, hast_counts_allreduce sums the counters in place over RCCL, and both contexts must
    then hold exactly what one context holds after classifying all the reads."""
    k, L, n_keys, n_bc, n = 21, 150, 200_000, 5000, 400_000
    p = make_params(k, L, n_keys, n_bc)
    ctxs = [hast_amd.Context(k, d) for d in (0, 1)]
    try:
        bufs = []
        for r, ctx in enumerate(ctxs):
            ctx.table_reserve(2 * n_keys)
            ctx.synth_table_build(p)
            ctx.counts_resize(n_bc)
            h = n // 2
            d_b, d_i = ctx.alloc(h * L + 64), ctx.alloc(h * 4)
            ctx.synth_reads_device(p, r * h, h, d_b, d_i)
            ctx.classify_device(d_b, h * L, h, L, d_barcode_ids=d_i)
            bufs.append((d_b, d_i))
        arr = (C.c_void_p * 2)(ctxs[0]._h, ctxs[1]._h)
        st = hast_amd.lib().hast_counts_allreduce(arr, 2)
        assert st == 0, hast_amd.lib().hast_last_error()
        merged = [ctx.counts_read(n_bc) for ctx in ctxs]
        ctx = ctxs[0]
        ctx.counts_zero()
        d_b, d_i = ctx.alloc(n * L + 64), ctx.alloc(n * 4)
        ctx.synth_reads_device(p, 0, n, d_b, d_i)
        ctx.classify_device(d_b, n * L, n, L, d_barcode_ids=d_i)
        single = ctx.counts_read(n_bc)
    finally:
        for ctx in ctxs:
            ctx.close()
    for m in merged:
        for a, b in zip(m, single):
            assert np.array_equal(a, b)
    assert int(single[0].sum()) > 0


def test_randomized_configurations_vs_oracle(built, oracle_lib):
    """Differential fuzz over the kernel's configuration space: K, minimizer length (incl. W > 9 -> runtime-loop
    instantiation), load factor (chain walks), fixed lengths down to L == K (plain-division instantiation), ragged
    lengths, tile tails.  Per-read votes and per-barcode counts must equal the oracle's in every configuration."""
    rng = random.Random(int(os.environ.get("HAST_FUZZ_SEED", "20261003")))
    for it in range(int(os.environ.get("HAST_FUZZ_ITERS", "36"))):
        k = rng.choice([3, 8, 12, 16, 19, 21, 24, 28, 31, 32])
        m = rng.choice([1, max(1, k - 12), max(1, k - 5), k - 1 if k > 1 else 1, k])
        lf = rng.choice([0.2, 0.5, 0.85])
        fixed = rng.random() < 0.5
        L = rng.choice([k, k + 1, k + 7, 64, 100, 151, 257]) if fixed else rng.randint(k + 1, 400)
        n_keys = rng.choice([50, 2000, 20000]) if k >= 8 else 20
        n_bc = rng.choice([1, 7, 300])
        n_reads = rng.choice([1, 63, 64, 1000, 4097])
        p = make_params(k, L, n_keys, n_bc, seed_k=rng.getrandbits(40) | 1, seed_r=rng.getrandbits(40) | 1, seed_b=rng.getrandbits(40) | 1)
        keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
        oc = oracle_from_keys(oracle_lib, k, keys[0], keys[1])
        if fixed:
            bases, ids = hast_amd.synth_reads_host(p, rng.randrange(10 ** 6), n_reads)
            off = np.arange(n_reads + 1, dtype=np.uint64) * L
        else:
            seqs = ragged_reads(rng, k, np.concatenate(keys), n_reads, L)
            lens = np.array([len(s) for s in seqs], dtype=np.uint64)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
            bases = np.frombuffer(b"".join(seqs) + b"A", dtype=np.uint8)[:-1].copy()
            ids = np.array([rng.randrange(n_bc) for _ in seqs], dtype=np.uint32)
            L = max(1, int(lens.max())) if len(seqs) else 1
        cfg = dict(it=it, k=k, m=m, lf=lf, fixed=fixed, L=L, n_keys=n_keys, n_bc=n_bc, n_reads=n_reads)
        with hast_amd.Context(k, minimizer=m) as ctx:
            ctx.table_reserve(2 * n_keys, lf)
            ctx.table_insert_keys(0, keys[0])
            ctx.table_insert_keys(1, keys[1])
            assert ctx.table_sizes() == (oracle_lib.ho_set_size(oc, 0), oracle_lib.ho_set_size(oc, 1)), cfg
            ctx.counts_resize(n_bc)
            d_b, d_i, d_v = ctx.to_device(bases), ctx.to_device(ids), ctx.alloc(max(1, n_reads) * 8)
            d_o = None if fixed else ctx.to_device(off)
            ctx.classify_device(d_b, bases.size, n_reads, L, d_offsets=d_o, d_barcode_ids=d_i, d_votes=d_v)
            ctx.sync()
            got = ctx.counts_read(n_bc)
            votes = ctx.to_host(d_v, (n_reads, 2), np.uint32)
        exp = oracle_counts(oracle_lib, oc, bases if bases.size else np.zeros(1, np.uint8), off, ids, n_bc, threads=2)
        assert np.array_equal(votes, oracle_votes(oracle_lib, oc, bases, off)), cfg
        for g, e in zip(got, exp):
            assert np.array_equal(g, e), cfg
        oracle_lib.ho_free(oc)


def test_table_save_load_roundtrip(built, oracle_lib, tmp_path):
    """Binary key-set cache (8(f)#4): save after build + erase, load into a new context with a different minimizer
    and load factor: same set sizes, same membership/tags, same classification; the file is deterministic."""
    k, n_keys, n_bc, L, n = 21, 30000, 50, 150, 5000
    p = make_params(k, L, n_keys, n_bc)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    keys[1][:200] = keys[0][:200]
    bases, ids = hast_amd.synth_reads_host(p, 0, n)
    off = np.arange(n + 1, dtype=np.uint64) * L
    f1, f2 = str(tmp_path / "t1.keys"), str(tmp_path / "t2.keys")
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(2 * n_keys)
        ctx.table_insert_keys(0, keys[0])
        ctx.table_insert_keys(1, keys[1])
        ctx.table_erase(np.concatenate([keys[0][:50], keys[1][300:320]]))
        sizes = ctx.table_sizes()
        ctx.counts_resize(n_bc)
        ctx.classify_batch(bases, off, ids, L)
        want = ctx.counts_read(n_bc)
        probe = np.concatenate([keys[0][:1000], keys[1][:1000]])
        tags = ctx.table_lookup(probe)
        ctx.table_save(f1)
    kk, nn = C.c_int(), C.c_uint64()
    assert hast_amd.lib().hast_table_file_info(f1.encode(), C.byref(kk), C.byref(nn)) == 0 and kk.value == k
    assert os.path.getsize(f1) == 32 + 8 * nn.value
    with hast_amd.Context(k, minimizer=11) as ctx:
        ctx.table_load(f1, 0.6)
        assert ctx.table_sizes() == sizes
        assert np.array_equal(ctx.table_lookup(probe), tags)
        ctx.counts_resize(n_bc)
        ctx.classify_batch(bases, off, ids, L)
        got = ctx.counts_read(n_bc)
        ctx.table_save(f2)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert open(f1, "rb").read() == open(f2, "rb").read()
    with hast_amd.Context(15) as ctx:
        with pytest.raises(hast_amd.HastError):
            ctx.table_load(f1)                       # K mismatch
    (tmp_path / "junk").write_bytes(b"not a table")
    with hast_amd.Context(k) as ctx:
        with pytest.raises(hast_amd.HastError):
            ctx.table_load(str(tmp_path / "junk"))


def test_poly_a_key_zero_and_hot_read(built, oracle_lib):
    """Canonical key 0 (poly-A / poly-T) is a legal key: slot value (0<<2)|tags must not look empty, erasing it must
    leave a tombstone rather than an empty slot, and reads made only of that k-mer (every window a hit, all on one
    LDS counter) count correctly."""
    k = 21
    polyA, polyT = b"A" * k, b"T" * k
    other = b"ACGTTGCATCGATTGCAAGTT"
    text0 = polyA + b"\n" + other + b"\n"
    text1 = polyT + b"\n"                                  # same canonical key as poly-A => key in both sets
    oc = oracle_lib.ho_new()
    assert oracle_lib.ho_load_kmers_text(oc, text0, len(text0), 0) == 0 and oracle_lib.ho_load_kmers_text(oc, text1, len(text1), 1) == 0
    seqs = [b"A" * 150, b"T" * 150, b"A" * 70 + b"C" + b"T" * 79, other * 3, b"A" * 20]
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    ids = np.arange(len(seqs), dtype=np.uint32)
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(16)
        assert ctx.table_insert_text(0, text0) == 2 and ctx.table_insert_text(1, text1) == 1
        assert ctx.table_sizes() == (2, 1)
        assert list(ctx.table_lookup(np.array([0], dtype=np.uint64))) == [3]
        ctx.counts_resize(len(seqs))
        ctx.classify_batch(bases, off, ids, 150)
        got = ctx.counts_read(len(seqs))
        exp = oracle_counts(oracle_lib, oc, bases, off, ids, len(seqs), threads=1)
        for g, e in zip(got, exp):
            assert np.array_equal(g, e)
        assert got[0][0] == 130 and got[1][0] == 130          # every window of the poly-A read hits, in both sets
        # erase key 0 (as InitAdaptor would for a poly-A adaptor): reported in both sets, then gone, sizes drop
        assert list(ctx.table_erase(np.array([0], dtype=np.uint64))) == [3]
        assert ctx.table_sizes() == (1, 0)
        assert list(ctx.table_lookup(np.array([0], dtype=np.uint64))) == [0]
        ctx.counts_zero()
        ctx.classify_batch(bases, off, ids, 150)
        got = ctx.counts_read(len(seqs))
        assert got[0][0] == 0 and got[1][0] == 0 and got[0][3] > 0
    oracle_lib.ho_free(oc)


def test_keys_piled_on_one_minimizer(built, oracle_lib):
    """Real genomes have m-mers that very many k-mers share (poly-A, low complexity): thousands of keys whose minimizer
    is A^16, and reads full of windows with that minimizer.  Results must match the oracle and the run must not crawl
    (overflowing keys jump to a bucket chosen by the key itself instead of piling up behind the home bucket)."""
    import time
    k, n_bc = 21, 16
    rng = random.Random(77)
    code = {"A": 0, "C": 1, "T": 2, "G": 3}
    def kmer_with_poly_a():
        left = rng.randint(0, 5)
        s = [rng.choice("ACGT") for _ in range(left)] + ["A"] * 16 + [rng.choice("ACGT") for _ in range(5 - left)]
        return "".join(s)
    texts = [[kmer_with_poly_a() for _ in range(20000)] + ["".join(rng.choice("ACGT") for _ in range(k)) for _ in range(3000)] for _ in (0, 1)]
    keys = [np.array(sorted({hast_amd.canon_kmer(t.encode()) for t in ts}), dtype=np.uint64) for ts in texts]
    oc = oracle_from_keys(oracle_lib, k, keys[0], keys[1])
    seqs = []
    for i in range(30000):
        s = [rng.choice("ACGT") for _ in range(150)]
        if i % 3:
            o = rng.randint(0, 100)
            s[o:o + 30] = "A" * 30                               # ~40 windows of this read have the A^16 minimizer
        if i % 5 == 0:
            o = rng.randint(0, 129)
            s[o:o + k] = rng.choice(texts[i & 1])
        seqs.append("".join(s).encode())
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    ids = np.array([rng.randrange(n_bc) for _ in seqs], dtype=np.uint32)
    for lf in (0.2, 0.8):
        with hast_amd.Context(k) as ctx:
            ctx.table_reserve(int(keys[0].size + keys[1].size), lf)
            t0 = time.time()
            ctx.table_insert_keys(0, keys[0])
            ctx.table_insert_keys(1, keys[1])
            assert ctx.table_sizes() == (keys[0].size, keys[1].size)
            ctx.counts_resize(n_bc)
            ctx.classify_batch(bases, off, ids, 150)
            got = ctx.counts_read(n_bc)
            dt = time.time() - t0
        exp = oracle_counts(oracle_lib, oc, bases, off, ids, n_bc)
        for g, e in zip(got, exp):
            assert np.array_equal(g, e)
        assert int(exp[0].sum()) > 1000
        assert dt < 20, "piled-up keys: %.1f s" % dt
    oracle_lib.ho_free(oc)


def test_kernel_timing_ring(built):
    """hast_classify_timing / hast_classify_times (bench.py's roofline clock): the last n calls, oldest first, both kernels"""
    k, L, n = 21, 150, 200_000
    p = make_params(k, L, 5000, 50)
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(10000)
        ctx.synth_table_build(p)
        ctx.counts_resize(50)
        d_b, d_i = ctx.alloc(n * L), ctx.alloc(n * 4)
        ctx.synth_reads_device(p, 0, n, d_b, d_i)
        ctx.classify_timing(4)
        for _ in range(6):
            ctx.classify_device(d_b, n * L, n, L, d_barcode_ids=d_i)
        a, b = ctx.classify_times()
        assert len(a) == 4 and len(b) == 4 and all(0 < x < 1000 for x in a) and all(0 < x < 1000 for x in b)
        assert ctx.classify_times() == ([], [])                      # forgotten once read
        ctx.classify_device(d_b, n * L, n, L, d_votes=ctx.alloc(n * 8))   # votes only: no bookkeeping kernel
        a, b = ctx.classify_times()
        assert len(a) == 1 and a[0] > 0 and b[0] < a[0]
        ctx.classify_timing(0)
        ctx.classify_device(d_b, n * L, n, L, d_barcode_ids=d_i)
        assert ctx.classify_times() == ([], [])
        ctx.sync()


@pytest.mark.parametrize("n_bc,hot", [(300, 0.0), (100_000, 0.0), (3_000_000, 0.0), (100_000, 0.4), (2_000_000, 0.9)])
def test_partitioned_commit_equals_atomic_commit_and_oracle(built, oracle_lib, monkeypatch, n_bc, hot):
    """The per-barcode bookkeeping (classify.cpp:203-208) two ways: one memory-side atomic per read (k_commit_votes) and the
    partitioned commit of large batches (pairs grouped by barcode range in LDS, bins summed in LDS, plain adds; the context
    option "commit" -- HAST_COMMIT in the environment when the context is created -- forces either).  hot: share of the reads that belong to ONE barcode (real stLFR data: "0_0_0" owns 10-20 %) -- its bin
    overflows and the rest goes through the overflow list.  All three counters of every barcode == oracle, both ways."""
    k, L, n_keys, n_reads = 21, 150, 20000, 300_000
    p = make_params(k, L, n_keys, n_bc)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    bases, ids = hast_amd.synth_reads_host(p, 11, n_reads)
    rng = np.random.default_rng(n_bc + int(hot * 10))
    if hot:
        ids = ids.copy()
        ids[rng.random(n_reads) < hot] = min(7, n_bc - 1)
    off = np.arange(n_reads + 1, dtype=np.uint64) * L
    oc = oracle_from_keys(oracle_lib, k, keys[0], keys[1])
    exp = oracle_counts(oracle_lib, oc, bases, off, ids, n_bc)
    oracle_lib.ho_free(oc)
    assert int(exp[2].sum()) > 0 and int(exp[0].sum()) > 0
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(2 * n_keys)
        ctx.table_insert_keys(0, keys[0])
        ctx.table_insert_keys(1, keys[1])
        ctx.counts_resize(n_bc)
        d_b, d_i = ctx.to_device(bases), ctx.to_device(ids)
        assert ctx.options() == ""
        for mode in ("partition", "atomic", "partition"):
            ctx.set_option("commit", {"atomic": 1, "partition": 2}[mode])
            assert ctx.options() == "commit=%d" % {"atomic": 1, "partition": 2}[mode]
            ctx.counts_zero()
            ctx.classify_device(d_b, bases.size, n_reads, L, d_barcode_ids=d_i)
            # twice into the same counters: the sums add up (plain adds of the partitioned path must not lose the first round)
            ctx.classify_device(d_b, bases.size, n_reads, L, d_barcode_ids=d_i)
            got = ctx.counts_read(n_bc)
            for a, b in zip(got, exp):
                assert np.array_equal(a, 2 * b), (mode, n_bc, hot)


@pytest.mark.parametrize("first", ["partition", "atomic"])
def test_counters_past_2_to_32_both_commit_paths(built, first):
    """SURVEY section 7 "Hot barcode": the no-barcode bucket "0_0_0" owns 10-20 % of real stLFR reads and the reference's `int`
    counters (classify.cpp:51) overflow on it.  The device counts in 64-bit words: synthetic per-read votes (hast_counts_add_votes:
    the bookkeeping of classify.cpp:203-208 alone) drive ONE barcode past 2^32 in c0 AND c1 through both commit paths -- one atomic
    per counter and read, and the partitioned path whose bin for that barcode overflows into the overflow list -- and both must
    give the exact sums: no wrap, no carry of c0 into c1, neighbours untouched."""
    n_bc, n, hot, rounds = 300_000, 4_000_000, 123_457, 20
    rng = np.random.default_rng(5)
    ids = rng.integers(0, n_bc, n, dtype=np.uint32)
    ids[rng.random(n) < 0.8] = hot                                   # 80 % of the reads belong to one barcode
    votes = rng.integers(0, 256, (n, 2), dtype=np.uint32)            # <= 255 per read: the partitioned path is usable
    votes[rng.random(n) < 0.25] = 0                                  # reads without a hit: neg += 1
    votes[::7, 1] = 0
    e0 = np.zeros(n_bc, np.uint64); e1 = np.zeros(n_bc, np.uint64); eneg = np.zeros(n_bc, np.uint64)
    np.add.at(e0, ids, votes[:, 0].astype(np.uint64))
    np.add.at(e1, ids, votes[:, 1].astype(np.uint64))
    np.add.at(eneg, ids, ((votes[:, 0] | votes[:, 1]) == 0).astype(np.uint64))
    assert rounds * int(e0[hot]) > 2**32 and rounds * int(e1[hot]) > 2**32 and int(e0[hot]) < 2**32
    with hast_amd.Context(21) as ctx:
        ctx.counts_resize(n_bc)
        d_v, d_i = ctx.to_device(votes), ctx.to_device(ids)
        modes = [first, "atomic" if first == "partition" else "partition"]
        for r in range(rounds):
            ctx.set_option("commit", {"atomic": 1, "partition": 2}[modes[r % 2]])
            ctx.counts_add_votes(d_v, d_i, n, 255)
        ctx.sync()
        c0, c1, neg = ctx.counts_read(n_bc)
    assert c0.dtype == np.uint64
    assert int(c0[hot]) == rounds * int(e0[hot]) > 2**32 and int(c1[hot]) == rounds * int(e1[hot]) > 2**32
    assert np.array_equal(c0, rounds * e0) and np.array_equal(c1, rounds * e1) and np.array_equal(neg, rounds * eneg)


def test_offsets_that_run_backwards_are_an_error_not_garbage(built):
    """Caller-supplied read offsets whose lengths add up to more than the buffer (overlapping or non-monotonic: a caller's bug)
    used to overflow the device-side segment table silently; now such reads get no rows and the next call that waits for results
    says so."""
    k = 21
    p = make_params(k, 150, 5000, 1)
    rng = np.random.default_rng(1)
    bases = rng.choice(np.frombuffer(b"ACGT", np.uint8), 60000)
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(10000)
        ctx.synth_table_build(p)
        good = np.array([0, 20000, 40000, 60000], np.uint64)
        v = ctx.classify_perread(bases, good)                                   # fine
        assert v.shape == (3, 2)
        for bad in (np.array([0, 30000, 10, 60000], np.uint64),                 # runs backwards: a "length" of 2^64 - 29990
                    np.array([0, 20000, 40000, 70000], np.uint64)):             # ends past the buffer
            d_b, d_o, d_v = ctx.to_device(bases), ctx.to_device(bad), ctx.alloc(3 * 8)
            ctx.classify_perread_device(d_b, bases.size, d_o, 3, d_v)
            with pytest.raises(hast_amd.HastError) as ei:
                ctx.sync()
            assert "offsets" in str(ei.value)
            ctx.sync()                                                          # reported once
        assert np.array_equal(ctx.classify_perread(bases, good), v)             # the context still works


def test_get_hap_and_cli_rows_past_int_max(built):
    """getHap (classify.cpp:66-86) on counts the reference's `int` cannot hold: the call goes by the true 64-bit counts."""
    n = 1000
    assert hast_amd.get_hap(b"1_2_3", 2**32 + 5, 7, n, n) == 0 and hast_amd.get_hap(b"1_2_3", 7, 2**33, n, n) == 1
    assert hast_amd.get_hap(b"1_2_3", 2**32, 2**32, n, n) == -1 and hast_amd.get_hap(b"0_0_0", 2**40, 1, n, n) == -1
    assert hast_amd.get_hap(b"1_2_3", 2**32, 0, n, n) == 0          # (a wrapped 32-bit c0 would read 0 here: "no hit at all")


@pytest.mark.parametrize("L", [100, 150, 151])
def test_kernels_with_geometry_and_row_length_compiled_in_vs_oracle(built, oracle_lib, monkeypatch, L):
    """k_classify_f exists with the BASELINE geometry (K = 21, m = 14, t = 6, kp = 21) and the row length (100, 150) compiled in;
    L = 151 takes the instantiation with the geometry alone, HAST_F_GEO=0 the generic one.  Fixed-length rows and ragged rows of
    at most L bases: votes and counters == oracle, and all three kernels agree."""
    k, n_keys, n_bc, n_reads = 21, 30000, 97, 20000
    p = make_params(k, L, n_keys, n_bc)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    bases, ids = hast_amd.synth_reads_host(p, 3, n_reads)
    rng = np.random.default_rng(L)
    lens = rng.integers(k - 2, L + 1, n_reads).astype(np.uint64)
    lens[::5] = L
    off = np.zeros(n_reads + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    rag = np.concatenate([bases[i * L:i * L + int(lens[i])] for i in range(n_reads)])
    oc = oracle_from_keys(oracle_lib, k, keys[0], keys[1])
    fixed_off = np.arange(n_reads + 1, dtype=np.uint64) * L
    exp_fixed = oracle_counts(oracle_lib, oc, bases, fixed_off, ids, n_bc)
    exp_rag = oracle_counts(oracle_lib, oc, rag, off, ids, n_bc)
    oracle_lib.ho_free(oc)
    assert int(exp_fixed[0].sum()) > 0 and int(exp_rag[0].sum()) > 0
    for geo in ("1", "0", "rl0"):
        # the switches are read from the environment ONCE, when the context is created (a variable that appears later cannot
        # re-route a live context), or set on the context itself
        monkeypatch.delenv("HAST_F_GEO", raising=False)
        if geo != "rl0":
            monkeypatch.setenv("HAST_F_GEO", geo)
        with hast_amd.Context(k) as ctx:
            monkeypatch.setenv("HAST_F_GEO", "1")                       # too late for this context
            if geo == "rl0":
                ctx.set_option("kernel_rl", 0)
            assert ctx.options() == {"1": "", "0": "kernel_geo=0", "rl0": "kernel_rl=0"}[geo]
            ctx.set_filter(1, 14, 6, 21)
            ctx.table_reserve(2 * n_keys)
            ctx.table_insert_keys(0, keys[0])
            ctx.table_insert_keys(1, keys[1])
            ctx.counts_resize(n_bc)
            d_b, d_i = ctx.to_device(bases), ctx.to_device(ids)
            ctx.classify_device(d_b, bases.size, n_reads, L, d_barcode_ids=d_i)
            assert ctx.filter_mode() == 2 and ctx.filter_info()[1:4] == (14, 6, 21)
            for a, b in zip(ctx.counts_read(n_bc), exp_fixed):
                assert np.array_equal(a, b), (L, geo, "fixed")
            ctx.counts_zero()
            d_r, d_o = ctx.to_device(rag), ctx.to_device(off)
            ctx.classify_device(d_r, rag.size, n_reads, L, d_offsets=d_o, d_barcode_ids=d_i)
            for a, b in zip(ctx.counts_read(n_bc), exp_rag):
                assert np.array_equal(a, b), (L, geo, "ragged")


def test_table_clone_is_ordered_in_front_of_what_the_copy_is_asked_next(built):
    """hast_table_clone between two contexts of ONE GPU is a device-to-device copy: it does not wait for the host, and the copy's
    stream is a non-blocking one -- nothing used to order the first reads classified on the new context behind the arrival of its
    table (round 5: `classify --devices 0,0,0` lost hits in one run of three once nothing else in the process stopped the device
    at the right moment).  A table large enough for the copy to take milliseconds, reads that hit it, classified on the copy the
    moment the clone returns: the same counters as on the original, every time."""
    k, L, n_keys, n_reads, n_bc = 21, 150, 30_000_000, 200_000, 1000
    p = make_params(k, L, n_keys, n_bc)
    with hast_amd.Context(k) as a:
        a.table_reserve(2 * n_keys)
        a.synth_table_build(p)
        a.counts_resize(n_bc)
        d_b, d_i = a.alloc(n_reads * L), a.alloc(n_reads * 4)
        a.synth_reads_device(p, 0, n_reads, d_b, d_i)
        a.classify_device(d_b, n_reads * L, n_reads, L, d_barcode_ids=d_i)
        a.sync()
        want = a.counts_read(n_bc)
        assert int(want[0].sum()) + int(want[1].sum()) > n_reads // 10
        for _ in range(6):
            with hast_amd.Context(k) as b:
                b.counts_resize(n_bc)
                b.table_clone_from(a)
                b.classify_device(d_b, n_reads * L, n_reads, L, d_barcode_ids=d_i)
                b.sync()
                got = b.counts_read(n_bc)
            for g, w in zip(got, want):
                assert np.array_equal(g, w)

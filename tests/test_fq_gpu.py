"""FASTQ framing on the GPU (-m gpu): hast_fq_* (fq_kernels.hip) against a plain restatement of the reference's reader
(processFastq, classify.cpp:257-268: four getlines per record, header must be newline-terminated, the rest may hit EOF;
parseName :112-119), on byte streams cut into blocks at arbitrary places -- records, headers and barcodes straddle block
borders all the time -- and the per-barcode counters after the commit against the oracle."""
import ctypes as C
import os
import random
import re

import numpy as np
import pytest

import hast_amd
from hast_amd.binding import FqBlock, FqRouted, make_params
from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu


def reference_framing(data: bytes, oracle_lib):
    """[(barcode, bases)] exactly as the reference's producer loop frames `data`"""
    lines = data.split(b"\n")          # the last piece is what follows the last newline (unterminated or empty)
    out, i = [], 0
    while 4 * i < len(lines) - 1:      # a header line that reaches EOF before its newline ends the input
        head = lines[4 * i]
        seq = lines[4 * i + 1] if 4 * i + 1 < len(lines) else b""
        out.append((ob.parse_name(oracle_lib, head), seq))
        i += 1
    return out


def make_fastq(rng, n, k, keys, tail):
    recs = []
    for i in range(n):
        L = rng.choice([k, k + 1, 60, 100, 150, 150, 150, rng.randint(k, 300)])
        s = [rng.choice("ACGT") for _ in range(L)]
        if rng.random() < 0.5:
            key = int(rng.choice(keys))
            ks = "".join("ACTG"[(key >> (2 * (k - 1 - j))) & 3] for j in range(k))
            p = rng.randint(0, L - k)
            s[p:p + k] = ks
        if rng.random() < 0.03:
            s[rng.randrange(L)] = "N"
        if rng.random() < 0.02:                      # a read shorter than K is only legal with an 'N' in it (kmer.h:171)
            s = list("ACGN"[:rng.randint(1, 4)]) if k > 4 else s
            if "N" not in s:
                s.append("N")
        bc = rng.choice(["0_0_0", "%d_%d_%d" % (rng.randint(1, 30), rng.randint(1, 3), rng.randint(1, 3)), "lib7_%d_222222_3333" % rng.randint(1, 9), ""])
        head = rng.choice(["@r{i}#{bc}/1", "@r{i}#{bc}/2\tx\t1", "@r{i}/x#{bc}", "@r{i}#{bc}", "@#{bc}/1/2", "@nohash{i}", "@r{i}#a#{bc}/1/"]).format(i=i, bc=bc)
        qual = "".join(rng.choice("@+IF#/") for _ in range(len(s)))     # quality lines may start with '@' or hold '#', '/'
        recs.append("%s\n%s\n+\n%s\n" % (head, "".join(s), qual))
    text = "".join(recs)
    if tail == "no_final_newline":
        text = text[:-1]
    elif tail == "header_only":
        text += "@last#9_9_9/1\n"
    elif tail == "unterminated_header":
        text += "@dropped#8_8_8/1"
    elif tail == "bases_no_newline":
        text += "@last#7_7_7/1\n" + "ACGT" * 10
    return text.encode()


def stream_through_framer(ctx, data, lo, hi, cache, rng, n_buffers=3, more_ctxs=(), eager=False, dict_mode=False, keep_names=None):
    """feed `data` to hast_fq_* in pieces of lo..hi bytes, name the records the way the CLI does; returns the barcodes in record
    order, the dictionary, the base count, the per-block short-read flags and how many records the host had to name.
    dict_mode: the table is the DICTIONARY (hast_names_create_dict): the device hands out the ids below hast_names_limit itself, the
    caller names what is left to it (long texts, and what arrives when every id is out) from the limit upwards; checked here: one id
    per text, dense from 0, below dict_ids, and the texts the dictionary files under its ids are the barcodes.
    more_ctxs: further contexts -> a striped stream (block i on context i % n; n_buffers per context; lo == hi: full blocks);
    eager: open a block as soon as the one behind it has been submitted (else: as late as the buffers allow)"""
    lib = hast_amd.lib()
    fq, nm = C.c_void_p(), C.c_void_p()
    ctxs = [ctx] + list(more_ctxs)
    nms = []
    if len(ctxs) > 1:
        assert lo == hi
        for c in ctxs:
            h = C.c_void_p()
            if cache:
                assert (lib.hast_names_create_dict if dict_mode else lib.hast_names_create)(c._h, cache, C.byref(h)) == 0, lib.hast_last_error()
            nms.append(h)
        arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
        narr = (C.c_void_p * len(ctxs))(*[h.value for h in nms])
        assert lib.hast_fq_create_striped(arr, len(ctxs), hi, n_buffers, narr, C.byref(fq)) == 0, lib.hast_last_error()
        assert lib.hast_fq_lanes(fq) == len(ctxs) and lib.hast_fq_block_bytes(fq) == hi
    else:
        if cache:
            assert (lib.hast_names_create_dict if dict_mode else lib.hast_names_create)(ctx._h, cache, C.byref(nm)) == 0, lib.hast_last_error()
        assert lib.hast_fq_create(ctx._h, hi, n_buffers, nm, C.byref(fq)) == 0, lib.hast_last_error()
    names, got, pos, pending = {}, [], 0, 0
    st = {"host_named": 0, "n_bases": 0}
    short = []
    limit = lib.hast_names_limit(nms[0] if nms else nm) if dict_mode else 0
    host_names = {}

    def drain():
        b = FqBlock()
        assert lib.hast_fq_next(fq, C.byref(b)) == 0, lib.hast_last_error()
        short.append(b.short_read)
        for i in range(b.n_records):
            bc = bytes(b.bytes[b.bc_pos[i]:b.bc_pos[i] + b.bc_len[i]])
            if b.bc_text:                            # the framer's compact copy of the text: length byte + up to 15 bytes
                t = bytes(b.bc_text[16 * i:16 * i + 16])
                assert (t[0] == 0xFF and len(bc) > 15) or (t[0] == len(bc) and t[1:1 + t[0]] == bc), (bc, t)
            got.append(bc)
        if b.unknown and dict_mode:
            todo = [b.unknown[j] for j in range(b.n_unknown)]
            base = len(got) - b.n_records
            for i in set(range(b.n_records)) - set(todo):      # the dictionary's own ids: one per text, below what it says it has handed out
                bc = got[base + i]
                assert b.ids[i] < b.dict_ids <= limit and len(bc) <= 15, (bc, b.ids[i], b.dict_ids)
                assert names.setdefault(bc, b.ids[i]) == b.ids[i], (bc, b.ids[i], names[bc])
            for i in todo:                           # left to the caller: longer than a text record, or every id was out
                bc = got[base + i]
                assert len(bc) > 15 or b.dict_ids >= limit or bc in host_names, (bc, b.dict_ids, limit)
                b.ids[i] = names.setdefault(bc, limit + host_names.setdefault(bc, len(host_names)))
            st["host_named"] += len(todo)
            st["n_bases"] += b.n_bases
            assert lib.hast_fq_commit(fq) == 0, lib.hast_last_error()
            return
        if b.unknown:
            assert cache and b.n_unknown <= b.n_records
            todo = [b.unknown[j] for j in range(b.n_unknown)]
            known = set(range(b.n_records)) - set(todo)
            base = len(got) - b.n_records
            for i in known:                          # what the cache answered must be what the caller said before
                assert b.ids[i] == names[got[base + i]], (got[base + i], b.ids[i])
        else:
            assert b.n_unknown == b.n_records
            todo = range(b.n_records)
        for i in todo:
            b.ids[i] = names.setdefault(got[len(got) - b.n_records + i], len(names))
        st["host_named"] += len(todo)
        st["n_bases"] += b.n_bases
        assert lib.hast_fq_commit(fq) == 0, lib.hast_last_error()

    while True:
        n = min(len(data) - pos, rng.randint(lo, hi))
        buf = C.POINTER(C.c_uint8)()
        assert lib.hast_fq_acquire(fq, C.byref(buf)) == 0, lib.hast_last_error()
        C.memmove(buf, data[pos:pos + n], n)
        pos += n
        last = pos >= len(data)
        assert lib.hast_fq_submit(fq, n, 1 if last else 0) == 0, lib.hast_last_error()
        pending += 1
        if len(ctxs) > 1:
            # a block of a striped stream can be opened once the block behind it has been submitted
            keep = 0 if last else 1 if eager else n_buffers * len(ctxs) - 1
            while pending > keep:
                drain()
                pending -= 1
        elif pending == n_buffers - 1 or last:     # keep block(s) in flight behind the one being named
            while pending > (0 if last else n_buffers - 2):
                drain()
                pending -= 1
        if last:
            break
    if len(ctxs) > 1:
        st["lane_records"] = [lib.hast_fq_lane_records(fq, g) for g in range(len(ctxs))]
        assert sum(st["lane_records"]) == len(got)
    lib.hast_fq_destroy(fq)
    if dict_mode and len(ctxs) == 1:
        # the ids are dense from 0, and what the dictionary filed under them is the text
        n = C.c_size_t()
        assert lib.hast_names_count(nm, C.byref(n)) == 0
        dev = {bc: i for bc, i in names.items() if i < limit}
        assert sorted(dev.values()) == list(range(n.value)), (n.value, len(dev))
        txt = (C.c_uint8 * (16 * max(n.value, 1)))()
        assert lib.hast_names_texts(nm, 0, n.value, txt) == 0, lib.hast_last_error()
        raw = bytes(txt)
        for bc, i in dev.items():
            assert raw[16 * i] == len(bc) and raw[16 * i + 1:16 * i + 1 + len(bc)] == bc, (bc, i)
    for h in nms + [nm]:
        if h and keep_names is not None:
            keep_names.append(h)               # (the caller destroys it)
        elif h:
            lib.hast_names_destroy(h)
    if len(ctxs) > 1:
        return got, names, st["n_bases"], short, st["host_named"], st["lane_records"]
    return got, names, st["n_bases"], short, st["host_named"]


SEEDS = {"plain": 11, "no_final_newline": 23, "header_only": 37, "unterminated_header": 41, "bases_no_newline": 59}


@pytest.mark.parametrize("cache", [0, 64, 1 << 16, -64, -(1 << 16)])
@pytest.mark.parametrize("tail", ["plain", "no_final_newline", "header_only", "unterminated_header", "bases_no_newline"])
@pytest.mark.parametrize("chunk", [(700, 4096), (4096, 4096), (50_000, 65536)])
def test_fq_framing_and_counts(oracle_lib, tail, chunk, cache):
    """cache: size of the device-side barcode name cache (0: none, the caller names every record; 64: far too small for the
    ~300 barcodes of the input, so it fills up and stops learning; 65536: ample; negative: the table is the DICTIONARY of that size --
    it hands out the ids itself, -64: 64 of them, the rest is the caller's, in its own id range above)"""
    dict_mode, cache = cache < 0, abs(cache)
    lo, hi = chunk
    rng = random.Random(SEEDS[tail] * 1000 + chunk[0] % 997)       # literal seeds: a failure can be replayed
    k, n_keys = 21, 3000
    p = make_params(k, 100, n_keys, 1)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    data = make_fastq(rng, 1500, k, np.concatenate(keys), tail)
    want = reference_framing(data, oracle_lib)
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(2 * n_keys)
        ctx.table_insert_keys(0, keys[0])
        ctx.table_insert_keys(1, keys[1])
        ctx.counts_resize((cache if dict_mode else 0) + 4096)
        got, names, n_bases, short, host_named = stream_through_framer(ctx, data, lo, hi, cache, rng, dict_mode=dict_mode)
        n_ids = max(names.values()) + 1 if names else 0
        counts = ctx.counts_read(n_ids)
    if dict_mode and cache >= 1 << 16:
        assert host_named == sum(1 for bc in got if len(bc) > 15)          # only what does not fit a text record
    elif cache >= 1 << 16 and chunk[0] < 5000:
        # many small blocks: the cache learns early; what stays with the host are the barcodes longer than 15 bytes (a quarter of
        # the records here) and first sightings
        n_long = sum(1 for bc in got if len(bc) > 15)
        assert n_long <= host_named < n_long + (len(got) - n_long) // 2
    assert got == [bc for bc, _ in want]
    assert n_bases == sum(len(s) for _, s in want)
    if tail == "header_only":
        # a last record with a terminated header and nothing after it is a record with an EMPTY read: the reference frames
        # it and then aborts on it (kmer.h:171); here the block that holds it reports a short read and is not classified
        assert short[-1] == 1 and not any(short[:-1])
        return
    assert not any(short)
    # counters == oracle on the framed reads
    oc = oracle_lib.ho_new()
    for h in (0, 1):
        assert oracle_lib.ho_load_keys(oc, keys[h].ctypes.data, keys[h].size, h, k) == 0
    bases = np.frombuffer(b"".join(s for _, s in want), dtype=np.uint8)
    off = np.zeros(len(want) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for _, s in want])
    ids = np.array([names[bc] for bc, _ in want], dtype=np.uint32)
    e = [np.zeros(n_ids, np.uint32) for _ in range(3)]
    oracle_lib.ho_classify_ids(oc, bases.ctypes.data, off.ctypes.data, ids.ctypes.data, ids.size, e[0].ctypes.data, e[1].ctypes.data,
                               e[2].ctypes.data, None, 2)
    oracle_lib.ho_free(oc)
    for a, b in zip(counts, e):
        assert np.array_equal(a, b)
    assert int(e[0].sum()) + int(e[1].sum()) > 100


def stream_device_blocks(ctx, gz_path, block, cache, n_buffers, lag, gz_chunk=4096, gz_pass=7, more_ctxs=()):
    """the same walk with DEVICE-side blocks: a .gz file inflated on the GPU (hast_gz_read_device) straight into the framer's block
    buffers (hast_fq_device_block / hast_fq_submit_device); `lag` blocks are kept in hand, filled but not yet submitted.
    more_ctxs: a STRIPED stream of device blocks over those contexts as well (n_buffers per context), fed by ONE deflate stream whose
    passes go to the contexts' GPUs in turn (hast_gz_open_multi_ex)"""
    lib = hast_amd.lib()
    fq, nm = C.c_void_p(), C.c_void_p()
    ctxs = [ctx] + list(more_ctxs)
    nms = []
    if len(ctxs) > 1:
        for c in ctxs:
            h = C.c_void_p()
            if cache:
                assert lib.hast_names_create(c._h, cache, C.byref(h)) == 0, lib.hast_last_error()
            nms.append(h)
        arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
        narr = (C.c_void_p * len(ctxs))(*[h.value for h in nms])
        assert lib.hast_fq_create_striped_ex(arr, len(ctxs), block, n_buffers, narr, 1, C.byref(fq)) == 0, lib.hast_last_error()
        n_buffers *= len(ctxs)
    else:
        if cache:
            assert lib.hast_names_create(ctx._h, cache, C.byref(nm)) == 0, lib.hast_last_error()
        assert lib.hast_fq_create_ex(ctx._h, block, n_buffers, nm, 1, C.byref(fq)) == 0, lib.hast_last_error()
    names, got, st = {}, [], {"n_bases": 0, "fetched": 0}
    short = []

    def drain():
        b = FqBlock()
        assert lib.hast_fq_next(fq, C.byref(b)) == 0, lib.hast_last_error()
        assert not b.bytes                                       # a device block has no host view ...
        short.append(b.short_read)
        host = None
        for i in range(b.n_records):
            t = bytes(b.bc_text[16 * i:16 * i + 16])
            if t[0] == 0xFF:                                     # ... until it is asked for (a barcode longer than 15 bytes)
                if host is None:
                    hp = C.POINTER(C.c_uint8)()
                    assert lib.hast_fq_block_host_bytes(fq, C.byref(hp)) == 0, lib.hast_last_error()
                    host = hp
                    st["fetched"] += 1
                got.append(bytes(host[b.bc_pos[i]:b.bc_pos[i] + b.bc_len[i]]))
            else:
                got.append(t[1:1 + t[0]])
        todo = [b.unknown[j] for j in range(b.n_unknown)] if b.unknown else range(b.n_records)
        for i in todo:
            b.ids[i] = names.setdefault(got[len(got) - b.n_records + i], len(names))
        st["n_bases"] += b.n_bases
        assert lib.hast_fq_commit(fq) == 0, lib.hast_last_error()

    with hast_amd.GzReader(ctx, gz_path, gz_chunk, gz_pass, ctxs=ctxs if len(ctxs) > 1 else None) as z:
        st["units"] = z.units()
        in_hand, submitted, opened, eof = [], 0, 0, False
        while not (eof and not in_hand and opened == submitted):
            in_use = (submitted - opened) + len(in_hand)
            # (a block of a striped stream can be opened once the block behind it has been submitted, or it ends the file)
            can_open = opened < submitted and (len(ctxs) == 1 or submitted - opened >= 2 or (eof and not in_hand))
            if not eof and len(in_hand) < lag and in_use < n_buffers:
                buf = C.POINTER(C.c_uint8)()
                assert lib.hast_fq_acquire(fq, C.byref(buf)) == 0, lib.hast_last_error()
                assert submitted or not buf                          # no pinned host copy on such a stream (until a block was fetched)
                d, fs = C.c_void_p(), C.c_void_p()
                assert lib.hast_fq_device_block(fq, C.byref(d), C.byref(fs)) == 0, lib.hast_last_error()
                n = z.read_device(d.value, block, fs)
                in_hand.append(n)
                eof = n < block
            elif in_hand and (len(in_hand) >= lag or eof or in_use >= n_buffers):
                n = in_hand.pop(0)
                assert lib.hast_fq_submit_device(fq, n, 1 if (eof and not in_hand) else 0) == 0, lib.hast_last_error()
                submitted += 1
            else:
                assert can_open, (opened, submitted, in_hand, eof)
                drain()
                opened += 1
    if len(ctxs) > 1:
        st["lane_records"] = [lib.hast_fq_lane_records(fq, g) for g in range(len(ctxs))]
    lib.hast_fq_destroy(fq)
    for h in nms + [nm]:
        if h:
            lib.hast_names_destroy(h)
    if len(ctxs) > 1:
        return got, names, st["n_bases"], short, st["fetched"], st["lane_records"], st["units"]
    return got, names, st["n_bases"], short, st["fetched"]


@pytest.mark.parametrize("block,n_buffers,lag", [(4096, 3, 2), (4096, 6, 5), (65536, 2, 1), (20480, 4, 3), (8192, 6, 1)])
@pytest.mark.parametrize("tail", ["plain", "bases_no_newline"])
def test_fq_device_blocks_filled_by_the_gpu_inflate(oracle_lib, tmp_path, tail, block, n_buffers, lag):
    """Blocks whose bytes never see the host: inflated on the GPU into the framer's buffers.  Small blocks, so that every buffer is
    reused dozens of times and records, headers and barcodes straddle block borders; `lag` = n_buffers - 1 is the most the API allows
    in hand unsubmitted (a buffer's old tail is still read by the framing of the block behind it: one more is refused).  Records,
    barcode text (fetched from the device for barcodes longer than 15 bytes), base counts and counters == the reference's framing."""
    import gzip
    rng = random.Random(SEEDS[tail] * 77 + block + lag)
    k, n_keys = 21, 3000
    p = make_params(k, 100, n_keys, 1)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    data = make_fastq(rng, 1500, k, np.concatenate(keys), tail)
    want = reference_framing(data, oracle_lib)
    gz = tmp_path / "reads.fq.gz"
    with gzip.open(gz, "wb", compresslevel=6) as f:
        f.write(data)
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(2 * n_keys)
        ctx.table_insert_keys(0, keys[0])
        ctx.table_insert_keys(1, keys[1])
        ctx.counts_resize(4096)
        got, names, n_bases, short, fetched = stream_device_blocks(ctx, str(gz), block, 1 << 16, n_buffers, lag)
        counts = ctx.counts_read(len(names))
    assert got == [bc for bc, _ in want] and fetched > 0
    assert n_bases == sum(len(s) for _, s in want) and not any(short)
    oc = oracle_lib.ho_new()
    for h in (0, 1):
        assert oracle_lib.ho_load_keys(oc, keys[h].ctypes.data, keys[h].size, h, k) == 0
    bases = np.frombuffer(b"".join(s for _, s in want), dtype=np.uint8)
    off = np.zeros(len(want) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for _, s in want])
    ids = np.array([names[bc] for bc, _ in want], dtype=np.uint32)
    e = [np.zeros(len(names), np.uint32) for _ in range(3)]
    oracle_lib.ho_classify_ids(oc, bases.ctypes.data, off.ctypes.data, ids.ctypes.data, ids.size, e[0].ctypes.data, e[1].ctypes.data, e[2].ctypes.data, None, 2)
    oracle_lib.ho_free(oc)
    for a, b in zip(counts, e):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("block,n_ctx,n_buffers,lag", [(4096, 2, 2, 2), (4096, 3, 2, 4), (20480, 4, 2, 3), (65536, 2, 3, 1)])
@pytest.mark.parametrize("tail", ["plain", "bases_no_newline"])
def test_fq_striped_device_blocks_one_gz_stream_over_several_contexts(oracle_lib, tmp_path, monkeypatch, tail, block, n_ctx, n_buffers, lag, split):
    """ONE .gz file over several contexts (the reference deals the reads of one file to all its workers whatever its
    encoding, classify.cpp:211-219,245-254): the passes of the deflate stream rotate over the contexts' GPUs (hast_gz_open_multi_ex),
    the inflated bytes are written on the device into the blocks of a STRIPED framer (block i on context i % n, framed from the
    newline count in front of it, the first bytes of block i + 1 copied into block i's view device to device).  All contexts sit on
    the test box's one GPU; split = HAST_GZ_SPLIT=contexts makes every context a decode unit of its own with its own copy of the
    compressed bytes, the window handed from unit to unit, and every block translated into a hand-over buffer and copied "peer to
    peer" -- the several-GPU code, on one GPU.  Records, barcode text, base counts, summed counters == the reference's framing."""
    import gzip
    if split:
        monkeypatch.setenv("HAST_GZ_SPLIT", "contexts")
    rng = random.Random(SEEDS[tail] * 131 + block + lag + n_ctx)
    k, n_keys = 21, 3000
    p = make_params(k, 100, n_keys, 1)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    data = make_fastq(rng, 3000, k, np.concatenate(keys), tail)
    want = reference_framing(data, oracle_lib)
    gz = tmp_path / "reads.fq.gz"
    with gzip.open(gz, "wb", compresslevel=6) as f:
        f.write(data)
    lib = hast_amd.lib()
    ctxs = [hast_amd.Context(k) for _ in range(n_ctx)]
    try:
        ctxs[0].table_reserve(2 * n_keys)
        ctxs[0].table_insert_keys(0, keys[0])
        ctxs[0].table_insert_keys(1, keys[1])
        for c in ctxs[1:]:
            assert lib.hast_table_clone(c._h, ctxs[0]._h) == 0, lib.hast_last_error()
        for c in ctxs:
            c.counts_resize(4096)
        got, names, n_bases, short, fetched, lanes, units = stream_device_blocks(ctxs[0], str(gz), block, 1 << 16, n_buffers, lag, gz_chunk=2048, gz_pass=5,
                                                                                 more_ctxs=ctxs[1:])
        arr = (C.c_void_p * n_ctx)(*[c._h for c in ctxs])
        assert lib.hast_counts_allreduce(arr, n_ctx) == 0, lib.hast_last_error()
        counts = ctxs[0].counts_read(len(names))
    finally:
        for c in ctxs:
            c.close()
    assert units == (n_ctx if split else 1)
    assert got == [bc for bc, _ in want] and fetched > 0
    assert n_bases == sum(len(s) for _, s in want) and not any(short)
    assert all(x > 0 for x in lanes), lanes
    for a, b in zip(counts, _oracle_counts_of(oracle_lib, keys, k, want, names)):
        assert np.array_equal(a, b)


def test_fq_device_blocks_too_many_in_hand_is_refused(tmp_path):
    import gzip
    gz = tmp_path / "x.fq.gz"
    with gzip.open(gz, "wb") as f:
        f.write(b"@a#1_1_1/1\nACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIII\n" * 2000)
    lib = hast_amd.lib()
    with hast_amd.Context(21) as ctx:
        ctx.table_reserve(100)
        ctx.table_insert_keys(0, np.array([12345], dtype=np.uint64))
        ctx.counts_resize(16)
        fq = C.c_void_p()
        assert lib.hast_fq_create_ex(ctx._h, 4096, 3, None, 1, C.byref(fq)) == 0
        with hast_amd.GzReader(ctx, str(gz)) as z:
            def fill():
                buf, d, fs = C.POINTER(C.c_uint8)(), C.c_void_p(), C.c_void_p()
                assert lib.hast_fq_acquire(fq, C.byref(buf)) == 0
                st = lib.hast_fq_device_block(fq, C.byref(d), C.byref(fs))
                if st == 0:
                    assert z.read_device(d.value, 4096, fs) == 4096
                return st
            blk = FqBlock()
            for _ in range(3):                                # the first round of buffers: nothing to wait for
                assert fill() == 0
                assert lib.hast_fq_submit_device(fq, 4096, 0) == 0
                assert lib.hast_fq_next(fq, C.byref(blk)) == 0 and lib.hast_fq_commit(fq) == 0
            assert fill() == 0 and fill() == 0                # two in hand of three buffers: the most there may be
            assert lib.hast_fq_submit_device(fq, 4096, 0) == 0
            assert lib.hast_fq_next(fq, C.byref(blk)) == 0 and lib.hast_fq_commit(fq) == 0
            assert fill() == 0                                # again two in hand
            assert lib.hast_fq_submit_device(fq, 4096, 0) == 0 and lib.hast_fq_submit_device(fq, 4096, 0) == 0
            for _ in range(2):
                assert lib.hast_fq_next(fq, C.byref(blk)) == 0 and lib.hast_fq_commit(fq) == 0
            # three acquired, none submitted: the third would overwrite a tail that the framing of the first has yet to read
            assert fill() == 0 and fill() == 0
            assert fill() == 1 and b"in hand" in lib.hast_last_error()
        lib.hast_fq_destroy(fq)


def test_fq_short_read_is_reported(oracle_lib):
    k = 21
    lib = hast_amd.lib()
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(100)
        ctx.table_insert_keys(0, np.array([12345], dtype=np.uint64))
        ctx.counts_resize(16)
        fq = C.c_void_p()
        assert lib.hast_fq_create(ctx._h, 4096, 2, None, C.byref(fq)) == 0
        data = b"@a#1_1_1/1\n" + b"ACGT" * 10 + b"\n+\n" + b"I" * 40 + b"\n@b#1_1_1/1\nACGTACGT\n+\nIIIIIIII\n"
        buf = C.POINTER(C.c_uint8)()
        assert lib.hast_fq_acquire(fq, C.byref(buf)) == 0
        C.memmove(buf, data, len(data))
        assert lib.hast_fq_submit(fq, len(data), 1) == 0
        b = FqBlock()
        assert lib.hast_fq_next(fq, C.byref(b)) == 0
        assert b.n_records == 2 and b.short_read == 1
        lib.hast_fq_destroy(fq)


@pytest.mark.parametrize("n_buffers", [2, 3])
def test_fq_slot_grown_on_the_copy_path_is_reused(oracle_lib, monkeypatch, n_buffers):
    """A block with more records than the slot's pinned per-record arrays hold takes the copy path and GROWS the slot; the
    slot's next block is then framed with the larger capacity, so every per-record array (extents + text, ids, device text,
    unknown list, publications) must have grown with it.  Tiny records (25 bytes) in 64-KB blocks: ~2600 records per block
    against an initial capacity of 7, every slot reused several times, device name cache on."""
    monkeypatch.setenv("HAST_FQ_HOST_RECORDS", "7")
    k = 5
    rng = random.Random(20260)
    kmers = ["".join(rng.choice("ACGT") for _ in range(k)).encode() for _ in range(200)]
    keys = [np.array(sorted({hast_amd.canon_kmer(x) for x in kmers[h::2]}), dtype=np.uint64) for h in (0, 1)]
    recs = []
    for i in range(40000):
        L = rng.randint(k, k + 3)
        s = "".join(rng.choice("ACGT") for _ in range(L))
        recs.append("@%d#%d_%d/1\n%s\n+\n%s\n" % (i % 10, rng.randint(1, 40), rng.randint(1, 9), s, "I" * L))
    data = "".join(recs).encode()
    want = reference_framing(data, oracle_lib)
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(1000)
        ctx.table_insert_keys(0, keys[0])
        ctx.table_insert_keys(1, keys[1])
        ctx.counts_resize(4096)
        got, names, n_bases, short, host_named = stream_through_framer(ctx, data, 60_000, 65536, 1 << 12, rng, n_buffers=n_buffers)
        counts = ctx.counts_read(len(names))
    assert got == [bc for bc, _ in want] and not any(short)
    assert n_bases == sum(len(s) for _, s in want)
    assert len(data) > 12 * 65536                      # every slot came round at least four times
    assert host_named < len(got) // 2                   # ... and after the first round the cache answered (the normal path ran)
    oc = oracle_lib.ho_new()
    for h in (0, 1):
        assert oracle_lib.ho_load_keys(oc, keys[h].ctypes.data, keys[h].size, h, k) == 0
    bases = np.frombuffer(b"".join(s for _, s in want), dtype=np.uint8)
    off = np.zeros(len(want) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for _, s in want])
    ids = np.array([names[bc] for bc, _ in want], dtype=np.uint32)
    e = [np.zeros(len(names), np.uint32) for _ in range(3)]
    oracle_lib.ho_classify_ids(oc, bases.ctypes.data, off.ctypes.data, ids.ctypes.data, ids.size, e[0].ctypes.data, e[1].ctypes.data,
                               e[2].ctypes.data, None, 2)
    oracle_lib.ho_free(oc)
    for a, b in zip(counts, e):
        assert np.array_equal(a, b)
    assert int(e[0].sum()) > 1000


def _oracle_counts_of(oracle_lib, keys, k, want, names):
    oc = oracle_lib.ho_new()
    for h in (0, 1):
        assert oracle_lib.ho_load_keys(oc, keys[h].ctypes.data, keys[h].size, h, k) == 0
    bases = np.frombuffer(b"".join(s for _, s in want), dtype=np.uint8)
    off = np.zeros(len(want) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for _, s in want])
    ids = np.array([names[bc] for bc, _ in want], dtype=np.uint32)
    e = [np.zeros(len(names), np.uint32) for _ in range(3)]
    oracle_lib.ho_classify_ids(oc, bases.ctypes.data, off.ctypes.data, ids.ctypes.data, ids.size, e[0].ctypes.data, e[1].ctypes.data,
                               e[2].ctypes.data, None, 2)
    oracle_lib.ho_free(oc)
    return e


@pytest.mark.parametrize("n_ctx,cache,eager", [(2, 1 << 16, False), (3, 0, True), (4, 64, False)])
@pytest.mark.parametrize("tail", ["plain", "no_final_newline", "header_only", "unterminated_header", "bases_no_newline"])
@pytest.mark.parametrize("block", [4096, 65536])
def test_fq_striped_stream_framing_and_counts(oracle_lib, tail, block, n_ctx, cache, eager):
    """ONE byte stream whose blocks rotate over several contexts (hast_fq_create_striped; here all on the one GPU of the test
    box: the same code up to the device ordinal): every block is framed on its own from the number of newlines in front of it,
    with a context switch at every block border -- records, headers and barcodes straddle them all the time (4-KB blocks hold
    a dozen records).  Records, barcode text and base counts == the reference's reader, the summed per-barcode counters ==
    oracle, every context got records."""
    rng = random.Random(SEEDS[tail] * 7919 + block + n_ctx)
    k, n_keys = 21, 3000
    p = make_params(k, 100, n_keys, 1)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    data = make_fastq(rng, 1500, k, np.concatenate(keys), tail)
    if tail == "plain" and block == 4096:
        # one more record that makes the file end exactly at a block border: the last block is then empty
        need = (-len(data)) % block
        need += block if need < 64 else 0
        head = b"@pad#5_5_5/1" + (b"x" if (need - 17) % 2 else b"")
        L = (need - len(head) - 5) // 2
        data += head + b"\n" + b"ACGT" * (L // 4) + b"A" * (L % 4) + b"\n+\n" + b"I" * L + b"\n"
        assert len(data) % block == 0
    want = reference_framing(data, oracle_lib)
    lib = hast_amd.lib()
    ctxs = [hast_amd.Context(k) for _ in range(n_ctx)]
    try:
        ctxs[0].table_reserve(2 * n_keys)
        ctxs[0].table_insert_keys(0, keys[0])
        ctxs[0].table_insert_keys(1, keys[1])
        for c in ctxs[1:]:
            assert lib.hast_table_clone(c._h, ctxs[0]._h) == 0, lib.hast_last_error()
        for c in ctxs:
            c.counts_resize(4096)
        got, names, n_bases, short, host_named, lanes = stream_through_framer(ctxs[0], data, block, block, cache, rng, n_buffers=2,
                                                                              more_ctxs=ctxs[1:], eager=eager)
        arr = (C.c_void_p * n_ctx)(*[c._h for c in ctxs])
        assert lib.hast_counts_allreduce(arr, n_ctx) == 0, lib.hast_last_error()
        counts = ctxs[0].counts_read(len(names))
    finally:
        for c in ctxs:
            c.close()
    assert got == [bc for bc, _ in want]
    assert n_bases == sum(len(s) for _, s in want)
    assert all(x > 0 for x in lanes), lanes
    if tail == "header_only":
        assert short[-1] == 1 or any(short)
        return
    assert not any(short)
    for a, b in zip(counts, _oracle_counts_of(oracle_lib, keys, k, want, names)):
        assert np.array_equal(a, b)


def test_fq_striped_stream_rejects_partial_blocks_and_early_opens():
    lib = hast_amd.lib()
    with hast_amd.Context(21) as a, hast_amd.Context(21) as b:
        arr = (C.c_void_p * 2)(a._h, b._h)
        fq = C.c_void_p()
        assert lib.hast_fq_create_striped(arr, 1, 4096, 2, None, C.byref(fq)) != 0 and b"buffers in all" in lib.hast_last_error()
        assert lib.hast_fq_create_striped(arr, 2, 4096, 2, None, C.byref(fq)) == 0, lib.hast_last_error()
        buf = C.POINTER(C.c_uint8)()
        assert lib.hast_fq_acquire(fq, C.byref(buf)) == 0
        assert lib.hast_fq_submit(fq, 100, 0) != 0 and b"full blocks" in lib.hast_last_error()
        C.memmove(buf, b"@r#1_1_1/1\n" + b"A" * 4084 + b"\n", 4096)
        assert lib.hast_fq_submit(fq, 4096, 0) == 0, lib.hast_last_error()
        blk = FqBlock()
        assert lib.hast_fq_poll(fq) == 0
        assert lib.hast_fq_next(fq, C.byref(blk)) != 0 and b"block behind it" in lib.hast_last_error()
        lib.hast_fq_destroy(fq)


def test_fq_striped_stream_reports_a_record_larger_than_the_overlap():
    """a block of a striped stream sees min(1 MB, block size) bytes of the next block: a record that needs more is an error
    (HAST_ERR_FORMAT), never a silently cut read"""
    lib = hast_amd.lib()
    data = b"@a#1_1_1/1\nACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIII\n" * 40
    data += b"@long#2_2_2/1\n" + b"ACGT" * 3000 + b"\n+\n" + b"I" * 12000 + b"\n" + b"@b#3_3_3/1\nACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIII\n" * 100
    with hast_amd.Context(21) as a, hast_amd.Context(21) as b:
        a.table_reserve(100)
        a.table_insert_keys(0, np.array([12345], dtype=np.uint64))
        assert lib.hast_table_clone(b._h, a._h) == 0
        a.counts_resize(16)
        b.counts_resize(16)
        arr = (C.c_void_p * 2)(a._h, b._h)
        fq = C.c_void_p()
        assert lib.hast_fq_create_striped(arr, 2, 4096, 4, None, C.byref(fq)) == 0, lib.hast_last_error()
        pos, statuses = 0, []
        pending = 0
        while pos < len(data):
            n = min(4096, len(data) - pos)
            buf = C.POINTER(C.c_uint8)()
            assert lib.hast_fq_acquire(fq, C.byref(buf)) == 0, lib.hast_last_error()
            C.memmove(buf, data[pos:pos + n], n)
            pos += n
            assert lib.hast_fq_submit(fq, n, 1 if pos >= len(data) else 0) == 0, lib.hast_last_error()
            pending += 1
            while pending > (0 if pos >= len(data) else 1):
                blk = FqBlock()
                st = lib.hast_fq_next(fq, C.byref(blk))
                statuses.append(st)
                if st != 0:
                    break
                for i in range(blk.n_records):
                    blk.ids[i] = 0
                assert lib.hast_fq_commit(fq) == 0
                pending -= 1
            if statuses and statuses[-1] != 0:
                break
        assert statuses and statuses[-1] == 6 and b"larger than" in lib.hast_last_error(), (statuses, lib.hast_last_error())
        lib.hast_fq_destroy(fq)


def test_fq_striped_stream_fuzz_line_phases(oracle_lib):
    """Randomised inputs for the line-phase arithmetic of striped streams: records of 1 .. 900 bytes with empty header, base or
    quality lines, runs of newlines, '@' and '+' anywhere, a block border after every possible line of a record, files that end
    after any line with and without a final newline -- framed over 3 contexts in 4-KB blocks; records, barcodes and base counts
    must equal the reference's four-getlines reader (classify.cpp:257-268).  K = 1, so every read is long enough and classified."""
    lib = hast_amd.lib()
    k = 1
    ctxs = [hast_amd.Context(k) for _ in range(3)]
    try:
        ctxs[0].table_reserve(16)
        ctxs[0].table_insert_keys(0, np.array([0], dtype=np.uint64))        # 'A' (canonical of A/T)
        ctxs[0].table_insert_keys(1, np.array([1], dtype=np.uint64))        # 'C' (canonical of C/G)
        for c in ctxs[1:]:
            assert lib.hast_table_clone(c._h, ctxs[0]._h) == 0, lib.hast_last_error()
        for it in range(40):
            rng = random.Random(9000 + it)
            recs = []
            for i in range(rng.randint(30, 160)):
                L = rng.choice([1, 2, 3, 30, 100, 150, rng.randint(1, 400)])
                head = rng.choice(["@r%d#%d_%d/1" % (i, rng.randint(1, 9), rng.randint(1, 9)), "", "@", "@x#", "+", "@a#b#c/d/e"])
                seq = "".join(rng.choice("ACGT") for _ in range(L)) if rng.random() < 0.95 else "ACGN"
                plus = rng.choice(["+", "", "+r%d" % i])
                qual = rng.choice(["I" * len(seq), "", "@" * 3, "+"])
                recs.append("%s\n%s\n%s\n%s\n" % (head, seq, plus, qual))
            text = "".join(recs)
            cut = rng.choice([0, 0, 1, 2, 3])                          # end the file after `cut` lines of one more record
            if cut:
                extra = "@tail#7_7/1\nACGTACGT\n+\nIIIIIIII\n".split("\n")
                text += "\n".join(extra[:cut]) + ("\n" if rng.random() < 0.5 else "")
            elif rng.random() < 0.3:
                text = text[:-1]
            data = text.encode()
            want = reference_framing(data, oracle_lib)
            # a read shorter than K = 1 is an empty base line: the reference would abort on it unless ... it holds no 'N' either:
            # keep those inputs out (they are the header_only case of the test above)
            if any(len(s) == 0 for _, s in want):
                continue
            for c in ctxs:
                c.counts_resize(2048)
            got, names, n_bases, short, host_named, lanes = stream_through_framer(ctxs[0], data, 4096, 4096, 1 << 12, rng, n_buffers=2,
                                                                                  more_ctxs=ctxs[1:], eager=bool(it & 1))
            assert got == [bc for bc, _ in want], it
            assert n_bases == sum(len(s) for _, s in want) and not any(short), it
    finally:
        for c in ctxs:
            c.close()


def _awk_route(data, cls_of):
    """quartering_fastq.awk:21-49 on a byte string: (runs per class 0..3, dropped field-2 texts in order)"""
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    out, dropped, cls = [bytearray() for _ in range(4)], [], 0
    for i, line in enumerate(lines):
        if i % 4 == 0:
            f = re.split(rb"[#/]", line)
            if len(f) > 1 and f[1] != b"0_0_0":
                cls = cls_of.get(f[1], -1)
                if cls < 0:
                    dropped.append(f[1])
            else:
                cls = 0
        if cls >= 0:
            out[cls] += line + b"\n"
    return [bytes(o) for o in out], dropped


def route_through_abi(data, cls_of, lo, hi, n_ctx, rng, k=21):
    """`data` through a routing stream (hast_fq_set_route / hast_fq_next_routed), plain or striped over n_ctx contexts of one GPU, in
    pieces of lo..hi bytes; what the device hands over (a block with a text longer than a text record or a barcode in no list, the partial
    record at the end of the file) is decided here by awk's rule.  Returns (the four runs, the dropped field-2 texts, block counts)."""
    lib = hast_amd.lib()
    ctxs = [hast_amd.Context(k) for _ in range(n_ctx)]
    got, dropped, st = [bytearray() for _ in range(4)], [], {"host_blocks": 0, "blocks": 0}
    try:
        short = [(t, c) for t, c in cls_of.items() if len(t) <= 15]
        text16 = np.zeros((max(len(short), 1), 16), np.uint8)
        ids = np.zeros(max(len(short), 1), np.uint32)
        for i, (t, c) in enumerate(short):
            text16[i, 0] = len(t)
            text16[i, 1:1 + len(t)] = np.frombuffer(t, np.uint8)
            ids[i] = c
        tab = C.c_void_p()
        assert lib.hast_names_create(ctxs[0]._h, max(4096, len(short)), C.byref(tab)) == 0, lib.hast_last_error()
        assert lib.hast_names_insert(tab, text16.ctypes.data_as(C.POINTER(C.c_uint8)), ids.ctypes.data_as(C.POINTER(C.c_uint32)), len(short)) == 0, lib.hast_last_error()
        fq = C.c_void_p()
        if n_ctx > 1:
            arr = (C.c_void_p * n_ctx)(*[c._h for c in ctxs])
            assert lib.hast_fq_create_striped(arr, n_ctx, hi, 2, None, C.byref(fq)) == 0, lib.hast_last_error()
        else:
            assert lib.hast_fq_create(ctxs[0]._h, hi, 3, None, C.byref(fq)) == 0, lib.hast_last_error()
        tabs = (C.c_void_p * n_ctx)(*[tab.value] * n_ctx)
        assert lib.hast_fq_set_route(fq, tabs, n_ctx) == 0, lib.hast_last_error()

        def decide(head):
            f = re.split(rb"[#/]", head)
            if len(f) <= 1 or f[1] == b"0_0_0":
                return 0
            c = cls_of.get(f[1], -1)
            if c < 0:
                dropped.append(f[1])
            return c

        def drain():
            b = FqRouted()
            assert lib.hast_fq_next_routed(fq, C.byref(b)) == 0, lib.hast_last_error()
            st["blocks"] += 1
            if not b.host_block:
                for c in range(4):
                    got[c] += bytes(b.run[c][:b.run_bytes[c]])
            else:
                st["host_blocks"] += 1
                for i in range(b.n_slots):
                    cl = b.rec_class[i]
                    if cl == 0xFD:
                        continue
                    rec = bytes(b.bytes[b.rec_start[i]:b.rec_start[i] + b.rec_len[i]])
                    if cl > 3:
                        cl = decide(rec.split(b"\n", 1)[0])
                    if cl >= 0:
                        got[cl] += rec
            if b.tail_bytes:
                rest = bytes(b.tail[:b.tail_bytes])
                cl = decide(rest.split(b"\n", 1)[0])
                if cl >= 0:
                    got[cl] += rest + (b"" if rest.endswith(b"\n") else b"\n")
            assert lib.hast_fq_commit(fq) == 0, lib.hast_last_error()

        pos, pending = 0, 0
        while True:
            n = min(len(data) - pos, hi if n_ctx > 1 else rng.randint(lo, hi))
            buf = C.POINTER(C.c_uint8)()
            assert lib.hast_fq_acquire(fq, C.byref(buf)) == 0, lib.hast_last_error()
            C.memmove(buf, data[pos:pos + n], n)
            pos += n
            last = pos >= len(data)
            assert lib.hast_fq_submit(fq, n, 1 if last else 0) == 0, lib.hast_last_error()
            pending += 1
            while pending > (0 if last else 1):
                drain()
                pending -= 1
            if last:
                break
        lib.hast_fq_destroy(fq)
        lib.hast_names_destroy(tab)
    finally:
        for c in ctxs:
            c.close()
    return [bytes(g) for g in got], dropped, st


@pytest.mark.parametrize("tail", ["plain", "no_final_newline", "unterminated_header", "bases_no_newline"])
@pytest.mark.parametrize("chunk,n_ctx", [((700, 4096), 1), ((4096, 4096), 1), ((50_000, 65536), 1), ((4096, 4096), 2), ((8192, 8192), 3)])
def test_fq_routing_through_the_abi(tail, chunk, n_ctx):
    """hast_fq_set_route / hast_fq_next_routed (include/hast.h "routing": the wrapper's steps 10-11, quartering_fastq.awk) through the C
    ABI: a FASTQ whose headers take every branch -- no field 2, "0_0_0", empty field, a field 2 that is not what parseName takes, texts
    longer than a text record (18 bytes), barcodes in no list -- goes through a routing stream (plain and striped over contexts of one
    GPU) in blocks of 700 bytes to 64 KB; the four runs of every block (or, where the device hands a block over, the records by its
    extents and classes with the undecided ones decided here by awk's rule) put together must be what the awk program writes, and the
    dropped barcodes the ones it reports, in order.  (_awk_route is pinned to the real awk program's outputs by
    tests/test_quartering_cpu.py::test_awk_model_of_the_gpu_tests_equals_the_awk_program.)"""
    import zlib
    lo, hi = chunk
    rng = random.Random(1000 + lo + n_ctx)
    k = 21
    keys = hast_amd.synth_keys_host(make_params(k, 100, 500, 1), 0, 0, 500)
    data = make_fastq(rng, 1200, k, keys, tail)
    # the lists: every short barcode of the input by a hash of its text -- a fifth of them in no list (awk: ERROR line, record dropped)
    fields = set()
    for i, line in enumerate(data.split(b"\n")):
        if i % 4 == 0:
            f = re.split(rb"[#/]", line)
            if len(f) > 1:
                fields.add(f[1])
    cls_of = {t: 1 + zlib.crc32(t) % 3 for t in fields if zlib.crc32(t) % 5 != 0 and t != b"0_0_0"}
    want, want_dropped = _awk_route(data, cls_of)
    got, dropped, st = route_through_abi(data, cls_of, lo, hi, n_ctx, rng)
    assert got == want
    assert dropped == want_dropped and len(dropped) > 10
    assert 0 < st["host_blocks"] <= st["blocks"]              # long texts and unlisted barcodes: those blocks are the caller's
    assert all(len(w) > 1000 for w in want)


@pytest.mark.parametrize("n_ctx,block", [(1, 16384), (2, 8192)])
def test_fq_routing_equals_the_awk_program_on_its_goldens(n_ctx, block):
    """the device router through the ABI on the quartering goldens -- inputs, barcode lists and the outputs of the REAL awk program
    (01.classify_stlfr_reads/quartering_fastq.awk under mawk; tests/golden/gen_golden.py): the hand-made edge case byte for byte incl.
    the ERROR line, the two rand_k21 files (the second with its unterminated tail record) by size and md5 of each of the four files"""
    import gzip
    import hashlib
    import json
    from tests.conftest import GOLDEN
    exp = json.load(open(os.path.join(GOLDEN, "quartering", "expected.json")))
    rng = random.Random(4)
    # the edge case
    e = exp["edge"]
    cls_of = {}
    for name, c in (("p.bc", 1), ("m.bc", 2), ("h.bc", 3)):
        for line in e["inputs"][name].encode().splitlines():
            cls_of.setdefault(re.split(rb"[#/]", line)[0], c)
    got, dropped, st = route_through_abi(e["inputs"]["e.fq"].encode(), cls_of, 4096, 4096, n_ctx, rng, k=7)
    names = {0: "e.fq.nobarcode.fastq", 1: "e.fq.paternal.fastq", 2: "e.fq.maternal.fastq", 3: "e.fq.homozygous.fastq"}
    for c in range(4):
        assert got[c].decode() == e["outputs"].get(names[c], ""), names[c]
    assert "".join("ERROR : unclassify barcode : %s\n" % d.decode() for d in dropped) == e["stderr"]
    # rand_k21 with the lists the wrapper derived from the reference's own output
    cls_of = {}
    for name, c in (("paternal", 1), ("maternal", 2), ("homozygous", 3)):
        for line in open(os.path.join(GOLDEN, "quartering", name + ".unique.barcodes"), "rb").read().splitlines():
            cls_of.setdefault(re.split(rb"[#/]", line)[0], c)
    for fq in ("r1.fq", "r2.fq"):
        data = gzip.open(os.path.join(GOLDEN, "rand_k21", fq + ".gz")).read()
        if fq == "r2.fq":
            data = data[:-1] + b"\n" + exp["r2_tail"].encode()
        got, dropped, st = route_through_abi(data, cls_of, block, block, n_ctx, rng)
        want = exp["files"][fq]
        for c, cls in enumerate(("nobarcode", "paternal", "maternal", "homozygous")):
            if cls in want:
                assert (len(got[c]), hashlib.md5(got[c]).hexdigest()) == (want[cls]["bytes"], want[cls]["md5"]), (fq, cls)
            else:
                assert got[c] == b"", (fq, cls)
        err = "".join("ERROR : unclassify barcode : %s\n" % d.decode() for d in dropped).encode()
        assert hashlib.md5(err).hexdigest() == want["stderr_md5"], fq
        assert st["blocks"] > 10


def test_two_dictionaries_one_numbering():
    """hast_names_merge (include/hast.h): what several GPUs need -- each GPU's dictionary numbers the barcodes in the order in which ITS blocks
    meet them, and before the counters can be summed the second dictionary's ids are expressed in the first one's numbering, ON THE GPU: its
    text records go through the first dictionary's naming kernel (an id that is there, or the next new one).  Two dictionaries on one GPU,
    fed two FASTQ streams with overlapping barcode sets; then perm = merge(A <- B): a text both know maps to A's id, a text only B knows
    gets a new id at the end of A, whose text record is that text; merging again changes nothing; and counters of B renumbered by perm
    (hast_counts_permute) add up with A's to what one dictionary over both streams counts."""
    lib = hast_amd.lib()
    k, n_keys = 21, 2000
    p = make_params(k, 100, n_keys, 1)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    rng = random.Random(77)
    data = [make_fastq(random.Random(s0), 900, k, np.concatenate(keys), "plain") for s0 in (5, 6)]
    kept = []
    res = []
    with hast_amd.Context(k) as ctx:
        ctx.table_reserve(2 * n_keys)
        ctx.table_insert_keys(0, keys[0])
        ctx.table_insert_keys(1, keys[1])
        counts = []
        for d in data + [data[0] + data[1]]:
            ctx.counts_resize((1 << 12) + 4096)
            got, names, n_bases, short, host_named = stream_through_framer(ctx, d, 4096, 4096, 1 << 12, rng, dict_mode=True, keep_names=kept)
            assert not any(short)
            res.append(names)
            limit = lib.hast_names_limit(kept[-1])
            counts.append((ctx.counts_read(limit + 4096), limit))
        A, B, AB = kept
        nA, nB = C.c_size_t(), C.c_size_t()
        assert lib.hast_names_count(A, C.byref(nA)) == 0 and lib.hast_names_count(B, C.byref(nB)) == 0
        perm = (C.c_uint32 * nB.value)()
        assert lib.hast_names_merge(A, B, 0, nB.value, perm) == 0, lib.hast_last_error()
        nA2 = C.c_size_t()
        assert lib.hast_names_count(A, C.byref(nA2)) == 0
        limit = counts[0][1]
        devA = {t: i for t, i in res[0].items() if i < limit}
        devB = {t: i for t, i in res[1].items() if i < limit}
        only_b = [t for t in devB if t not in devA]
        assert len(only_b) > 5 and len(set(devA) & set(devB)) > 5
        assert nA2.value == nA.value + len(only_b)
        txt = (C.c_uint8 * (16 * nA2.value))()
        assert lib.hast_names_texts(A, 0, nA2.value, txt) == 0
        raw = bytes(txt)
        for t, i in devB.items():
            g = perm[i]
            if t in devA:
                assert g == devA[t], (t, g, devA[t])
            else:
                assert nA.value <= g < nA2.value
            assert raw[16 * g] == len(t) and raw[16 * g + 1:16 * g + 1 + len(t)] == t, (t, g)
        perm2 = (C.c_uint32 * nB.value)()
        assert lib.hast_names_merge(A, B, 0, nB.value, perm2) == 0 and list(perm2) == list(perm)
        assert lib.hast_names_count(A, C.byref(nA)) == 0 and nA.value == nA2.value
        # the counters: B's, renumbered on the device (hast_counts_permute), + A's == what ONE dictionary counted over both streams
        (ca, _), (cb, _), (cab, _) = counts
        n_cnt = limit + 4096
        ctx.counts_resize(n_cnt)
        packed = np.concatenate([cb[w] for w in range(3)]).astype(np.uint64)
        d_packed = ctx.alloc(packed.nbytes)
        assert lib.hast_memcpy_h2d(ctx._h, C.c_void_p(d_packed), packed.ctypes.data_as(C.c_void_p), packed.nbytes) == 0
        ctx.counts_unpack(d_packed, n_cnt)
        ctx.sync()
        assert lib.hast_counts_permute(ctx._h, perm, nB.value, n_cnt) == 0, lib.hast_last_error()
        moved_dev = ctx.counts_read(n_cnt)
        for w in range(3):
            moved = np.zeros(n_cnt, np.uint64)
            np.add.at(moved, np.array(list(perm), dtype=np.int64), cb[w][:nB.value])
            moved[nB.value:] += cb[w][nB.value:]              # (the records behind the dictionary's ids keep their places)
            assert np.array_equal(moved_dev[w], moved), w
        ab_dev = {t: i for t, i in res[2].items() if i < limit}
        assert set(ab_dev) == set(devA) | set(devB)
        for t, i in ab_dev.items():
            g = devA[t] if t in devA else perm[devB[t]]
            for w in range(3):
                assert int(ca[w][g] if t in devA else 0) + int(moved_dev[w][g]) == int(cab[w][i]), (t, w)
        for h in kept:
            lib.hast_names_destroy(h)

"""The drop-in boundary on the GPU (-m gpu): the `classify` executable (HIP path) must print exactly
what the real reference binary printed for the committed golden inputs (tests/golden/, made by
gen_golden.py from oracle/_ref/classify)."""
import os
import subprocess

import pytest

import numpy as np

import hast_amd
from tests.conftest import ROOT, golden_cases, load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(hast_amd.classify_exe()):
        hast_amd.build()
    return hast_amd.classify_exe()


@pytest.mark.parametrize("extra", [[], ["--host-parse"]])
@pytest.mark.parametrize("case,run", golden_cases("s01"))
def test_cli_matches_reference_golden(exe, golden_workdir, case, run, extra):
    meta = load_case(case)["runs"][run]
    d = golden_workdir / case
    res = subprocess.run([exe] + meta["argv"] + extra, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert res.stdout == open(d / meta["expected"], "rb").read()
    mine = [l for l in res.stderr.decode().splitlines() if l.startswith("Recorded") or "erase a adaptor" in l]
    assert mine == meta["ref_log"]


def test_cli_kmer_files_may_be_pipes(exe, golden_workdir):
    """regular k-mer files are streamed into the table by pread workers; anything else (here: both files through `cat`) can
    only be read once, front to back -- same stdout, same "Recorded" lines"""
    case, run = golden_cases("s01")[0]
    meta = load_case(case)["runs"][run]
    d = golden_workdir / case
    argv = list(meta["argv"])
    files = {}
    for flag in ("--hap0", "--hap1", "-p", "-m"):
        if flag in argv:
            files[flag] = argv[argv.index(flag) + 1]
    assert len(files) == 2
    import shlex
    cmd = []
    for a in [exe] + argv:
        cmd.append("<(cat %s)" % shlex.quote(a) if a in files.values() else shlex.quote(a))
    res = subprocess.run(["bash", "-c", " ".join(cmd)], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert res.stdout == open(d / meta["expected"], "rb").read()
    mine = [l for l in res.stderr.decode().splitlines() if l.startswith("Recorded") or "erase a adaptor" in l]
    assert mine == meta["ref_log"]


def _n_gpus():
    import ctypes
    try:
        n = ctypes.c_int(0)
        return n.value if ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


@pytest.mark.parametrize("devices,deal", [("0,0", "blocks"), ("0,0,0", "blocks"), ("0,0,0", "files"),
                                          pytest.param("0,1", "blocks", marks=pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs"))])
@pytest.mark.parametrize("case,run", golden_cases("s01"))
def test_cli_multi_gpu_matches_reference_golden(exe, golden_workdir, case, run, devices, deal):
    """--devices: the table is built once and copied to the other contexts, the BLOCKS of every input file are dealt to the
    contexts in turn and framed there (classify.cpp:211-219 spreads the reads of one file over all workers; deal == "files":
    whole files in turn, HAST_DEAL=files), the counters are summed once at the end (classify.cpp:226-229,276-277 across GPUs).
    Integer sums: stdout must be byte-identical to the real reference's.  "0,0" = several contexts on the one GPU of the test
    box (same code path up to the sum: a kernel instead of RCCL); "0,1" = two GPUs over RCCL.  With blocks dealt, every context
    must have framed records of the (one or two) input files."""
    if (devices, deal) != ("0,0,0", "blocks") and case != "rand_k21":
        pytest.skip("the other dealings run on the rand_k21 case only (suite time)")
    meta = load_case(case)["runs"][run]
    d = golden_workdir / case
    env = dict(os.environ, HAST_DEAL="files") if deal == "files" else None
    gz = any(a.endswith(".gz") for a in meta["argv"])
    argv = [exe] + meta["argv"] + ["--devices", devices, "--batch-reads", "50", "--initial-barcodes", "50", "--stats"]
    res = subprocess.run(argv, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert res.stdout == open(d / meta["expected"], "rb").read()
    if gz and deal == "blocks":
        # an ordinary .gz file is inflated on ONE GPU (a deflate stream is serial; whole files are dealt to the contexts in turn):
        # the blocks of its inflated bytes are spread over the contexts when the host inflates (HAST_INFLATE=host) -- both routes
        # must print the reference's bytes, the per-context record counts below are the second run's
        res = subprocess.run(argv, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, HAST_INFLATE="host"))
        assert res.returncode == 0, res.stderr.decode()[-2000:]
        assert res.stdout == open(d / meta["expected"], "rb").read()
    # --stats says which measurement switches the context was created with (none here: the environment sets none)
    assert [l for l in res.stderr.decode().splitlines() if l.startswith("__stats_switches__")] == ["__stats_switches__ none"]
    line = [l for l in res.stderr.decode().splitlines() if l.startswith("__stats_devices__")]
    if deal == "files":
        assert not line
        return
    assert len(line) == 1, res.stderr.decode()[-2000:]
    per = [int(x) for x in line[0].split("records_per_context=")[1].split(",")]
    assert len(per) == len(devices.split(","))
    if case.startswith("rand_"):                      # (the hand-made edge case is a single 2-KB block)
        assert all(x > 0 for x in per), line


@pytest.mark.parametrize("env,extra", [({"HAST_NAME_DICT": "0"}, []), ({"HAST_NAME_CACHE": "64"}, []), ({"HAST_NAME_CACHE": "64"}, ["--initial-barcodes", "3", "--batch-reads", "97"]),
                                       ({"HAST_NAME_DICT": "context"}, ["--devices", "0,0,0"]),
                                       ({"HAST_NAME_DICT": "context"}, ["--devices", "0,0,0", "--batch-reads", "50", "--initial-barcodes", "50"]),
                                       ({"HAST_NAME_DICT": "context", "HAST_NAME_CACHE": "64"}, ["--devices", "0,0", "--batch-reads", "120", "--initial-barcodes", "7"]),
                                       ({"HAST_NAME_DICT": "context", "HAST_DEAL": "files"}, ["--devices", "0,0"]),
                                       ({}, ["--initial-barcodes", "3", "--batch-reads", "31"]),
                                       ({"HAST_FQ_HOST_RECORDS": "7"}, ["--batch-reads", "40"]), ({"HAST_FQ_HOST_RECORDS": "7", "HAST_NAME_DICT": "0"}, ["--batch-reads", "40"])])
@pytest.mark.parametrize("case,run", golden_cases("s01"))
def test_cli_who_numbers_the_barcodes(exe, golden_workdir, case, run, env, extra):
    """The ids of the barcodes come from the GPU's own dictionary (round 6; classify.cpp:52-56 on the device: the naming kernel claims
    a slot for a new text and takes the next id from one counter), the host reads the texts by id once, for printing.  Same stdout
    as the real reference binary with: the host's dictionary as up to round 5 (HAST_NAME_DICT=0); a dictionary of 64 ids, so that
    most barcodes are named by the host in its own id range above the device's (and the counters regrow under both numberings);
    one dictionary per CONTEXT (what several GPUs have: each numbers in its own order), the counters merged by text -- at the end, and
    at every regrowth in the middle of the run; whole files dealt to the contexts; blocks with more records than the pinned per-record
    arrays hold (HAST_FQ_HOST_RECORDS=7: the first block of every buffer) among blocks that fit -- a barcode named by the host in one
    block and by the device in the next would print two rows."""
    if case not in ("rand_k21", "edge_k7") and (extra or len(env) > 1):
        pytest.skip("the combinations run on two cases (suite time)")
    meta = load_case(case)["runs"][run]
    d = golden_workdir / case
    res = subprocess.run([exe] + meta["argv"] + extra + ["--stats"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, **env))
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert res.stdout == open(d / meta["expected"], "rb").read()
    line = [l for l in res.stderr.decode().splitlines() if l.startswith("__stats_read_phase__")][0]
    kv = dict(x.split("=") for x in line.split()[1:])
    if not env and case.startswith("rand_"):
        assert int(kv["records_named_on_host"]) == 0, line       # stLFR barcodes fit a text record: nothing is left to the host


@pytest.mark.parametrize("extra", [["--batch-reads", "257"], ["--batch-reads", "13"], ["--batch-reads", "257", "--host-parse"]])
def test_cli_small_batches_and_counter_growth(exe, golden_workdir, extra, monkeypatch):
    """Tiny blocks (82 KB / 4 KB: every other record straddles a block border of the GPU framer) force many launches and the
    counter-array regrowth path; output unchanged.  --host-parse: the round-1 host framing path."""
    meta = load_case("rand_k21")["runs"]["pair_w104"]
    d = golden_workdir / "rand_k21"
    if extra == ["--batch-reads", "257"]:
        monkeypatch.setenv("HAST_FQ_HOST_RECORDS", "7")        # blocks with more records than the framer wrote to the host itself
    res = subprocess.run([exe] + meta["argv"] + extra + ["--initial-barcodes", "3", "--stats"], cwd=d,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert res.stdout == open(d / meta["expected"], "rb").read()
    assert "__stats__" in res.stderr.decode()


def test_cli_usage_and_errors(exe, golden_workdir, tmp_path):
    d = golden_workdir / "edge_k7"
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 255 and b"--hap0" in r.stderr and r.stdout == b""         # classify.cpp:425-428
    r = subprocess.run([exe, "--hap0", "hap0.mer", "--hap1", "hap1.mer", "--read", "r1.fq", "-t", "0"], cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 255
    r = subprocess.run([exe, "--hap0", "missing.mer", "--hap1", "hap1.mer", "--read", "r1.fq"], cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 2 and r.stdout == b""       # the reference would hang here (classify.cpp:41)
    short = tmp_path / "short.fq"
    short.write_text("@x#1_1_1/1\nACG\n+\nFFF\n")
    r = subprocess.run([exe, "--hap0", "hap0.mer", "--hap1", "hap1.mer", "--read", str(short)], cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 3                           # the reference aborts (kmer.h:171)
    # the same behind 40 000 good records, with --stats (the HBM sampler thread polls the runtime) and the library's threads at work: still
    # status 3 and the message, not a SIGSEGV out of a runtime that exit() was taking down under them (round 6)
    late = tmp_path / "late_short.fq"
    good = (golden_workdir / "edge_k7" / "r1.fq").read_text()
    n_good = good.count("\n") // 4
    late.write_text(good * (40000 // max(n_good, 1) + 1) + "@x#1_1_1/1\nACG\n+\nFFF\n")
    for _ in range(3):
        r = subprocess.run([exe, "--hap0", "hap0.mer", "--hap1", "hap1.mer", "--read", str(late), "-t", "8", "--stats"], cwd=d,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 3 and b"read shorter than K" in r.stderr, (r.returncode, r.stderr[-300:])


def _write_case(tmp_path, n_records, seed, gz, long_reads=False):
    """A FASTQ the oracle and the CLI both read: ragged lengths, odd headers, N reads, last line unterminated."""
    import gzip
    import random
    import numpy as np
    from hast_amd.binding import make_params
    rng = random.Random(seed)
    k, n_keys = 21, 4000
    p = make_params(k, 100, n_keys, 1)
    keys = [hast_amd.synth_keys_host(p, h, 0, n_keys) for h in (0, 1)]
    tostr = lambda key: "".join("ACTG"[(int(key) >> (2 * (k - 1 - j))) & 3] for j in range(k))
    for h in (0, 1):
        (tmp_path / ("hap%d.mer" % h)).write_text("".join(tostr(x) + "\n" for x in keys[h]))
    allk = np.concatenate(keys)
    recs = []
    for i in range(n_records):
        L = rng.choice([k, 30, 100, 100, 100, 150, rng.randint(k, 400)])
        if long_reads and rng.random() < 0.05:
            L = rng.randint(4000, 90000)                       # beyond one kernel row: segmented path
        s = [rng.choice("ACGT") for _ in range(L)]
        for _ in range(rng.randint(0, 2)):
            o = rng.randint(0, L - k)
            s[o:o + k] = tostr(rng.choice(allk))
        if rng.random() < (0.3 if L > 4000 else 0.02):
            s[rng.randrange(L)] = "N"
        bc = "0_0_0" if rng.random() < 0.15 else "%d_%d_%d" % (rng.randint(1, 40), rng.randint(1, 40), rng.randint(1, 3))
        head = "@R%07d#%s/%d\tx%d" % (i, bc, 1 + (i & 1), i) if rng.random() < 0.9 else "@odd%d/%s#%s" % (i, "y" * rng.randint(0, 5), bc)
        recs.append("%s\n%s\n+\n%s\n" % (head, "".join(s), "#@+\t"[0:1] * L))
    text = "".join(recs)[:-1]                                   # last quality line unterminated
    path = tmp_path / ("reads.fq.gz" if gz else "reads.fq")
    if gz:
        with gzip.open(path, "wb", compresslevel=1) as f:
            f.write(text.encode())
    else:
        path.write_text(text)
    return path


@pytest.mark.parametrize("gz,threads,batch", [(False, 1, 0), (False, 7, 900), (True, 4, 2500), (False, 16, 64)])
def test_cli_parallel_ingest_matches_oracle(exe, oracle_dir, tmp_path, gz, threads, batch):
    """Multi-threaded block ingest (carry across blocks, tail record, sharded barcode dictionary, counter
    regrowth) against the oracle's line-by-line reader on the same file."""
    path = _write_case(tmp_path, 30000, seed=threads * 10 + batch, gz=gz)
    args = ["--hap0", "hap0.mer", "--hap1", "hap1.mer", "--read", path.name, "--read", path.name, "--weight0", "1.04"]
    ref = subprocess.run([os.path.join(oracle_dir, "oracle_classify")] + args, cwd=tmp_path, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert ref.returncode == 0, ref.stderr.decode()[-500:]
    extra = ["-t", str(threads), "--initial-barcodes", "16"] + (["--batch-reads", str(batch)] if batch else [])
    got = subprocess.run([exe] + args + extra, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert got.returncode == 0, got.stderr.decode()[-2000:]
    assert got.stdout == ref.stdout
    assert len(got.stdout.splitlines()) > 1000


def test_cli_empty_crlf_and_many_files(exe, oracle_dir, tmp_path):
    """Inputs at the edges of the framing (classify.cpp:257-268): an empty file, a file of one record without a final newline,
    CR LF line ends (the reference does not strip the CR: it is one more base, coded as T), and more input files than
    streams are kept open at once -- against the oracle's line-by-line reader, framed on the GPU and on the host."""
    path = _write_case(tmp_path, 3000, seed=5, gz=False)
    text = path.read_bytes()
    (tmp_path / "empty.fq").write_bytes(b"")
    (tmp_path / "one.fq").write_bytes(text.split(b"\n@R")[0].rstrip(b"\n"))
    crlf = b"\r\n".join(text.split(b"\n")[:4000 - 4000 % 4]) + b"\r\n"
    (tmp_path / "crlf.fq").write_bytes(crlf)
    files = ["empty.fq", "one.fq", "crlf.fq", path.name, "empty.fq", path.name, "crlf.fq", "one.fq", path.name, path.name]
    args = ["--hap0", "hap0.mer", "--hap1", "hap1.mer", "--weight0", "1.04"]
    for f in files:
        args += ["--read", f]
    ref = subprocess.run([os.path.join(oracle_dir, "oracle_classify")] + args, cwd=tmp_path, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert ref.returncode == 0, ref.stderr.decode()[-500:]
    for extra in ([], ["--batch-reads", "40"], ["--host-parse"], ["--devices", "0,0,0"]):
        got = subprocess.run([exe] + args + ["-t", "5"] + extra, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert got.returncode == 0, got.stderr.decode()[-2000:]
        assert got.stdout == ref.stdout, extra
    assert len(ref.stdout.splitlines()) > 500


def test_cli_gz_decoders_agree_and_damaged_gz_is_an_error(exe, golden_workdir, tmp_path):
    """.gz reads through every inflate route on a golden case: ON THE GPU (the default for ordinary gzip files: hast_gz_* +
    device-side blocks of the framer; --stats says so), zlib (HAST_INFLATE=zlib), the host's parallel and serial decoders
    (HAST_INFLATE=host); also with two contexts (a .gz file stays on one GPU).  A truncated or bit-flipped gz file must not pass as
    a shorter / other input on any route."""
    import shutil
    d = tmp_path / "gz"
    shutil.copytree(golden_workdir / "rand_k21", d)
    args = ["--hap0", "hap0.mer", "--hap1", "hap1.mer", "--read", "r1.fq.gz", "--read", "r2.fq.gz"]
    a = subprocess.run([exe] + args + ["--stats"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    b = subprocess.run([exe] + args + ["--stats"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, HAST_INFLATE="zlib"))
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and len(a.stdout) > 100
    assert a.stderr.count(b"__stats_gz__") == 2 and b"__stats_gz__" not in b.stderr          # both files inflated on the device / none
    for extra in (["--devices", "0,0"], ["--devices", "0,0,0"], ["--batch-reads", "13"]):
        c = subprocess.run([exe] + args + extra + ["--stats"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert c.returncode == 0 and c.stdout == a.stdout and c.stderr.count(b"__stats_gz__") == 2, (extra, c.stderr[-300:])
    # one gzip stream inflated by several host threads (par_inflate.h), and by the serial decoder alone
    for gz_threads in ("1", "2", "5", "16"):
        c = subprocess.run([exe] + args + ["--stats"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, HAST_INFLATE="host", HAST_GZ_THREADS=gz_threads))
        assert c.returncode == 0 and c.stdout == a.stdout and b"__stats_gz__" not in c.stderr, (gz_threads, c.stderr[-300:])
    whole = (d / "r2.fq.gz").read_bytes()
    flipped = bytearray(whole)
    flipped[len(whole) // 3] ^= 0x10
    for damaged in (whole[:len(whole) // 2], whole[:-4], bytes(flipped)):
        (d / "r2.fq.gz").write_bytes(damaged)
        for env in (None, dict(os.environ, HAST_INFLATE="zlib"), dict(os.environ, HAST_INFLATE="host", HAST_GZ_THREADS="1"), dict(os.environ, HAST_INFLATE="host", HAST_GZ_THREADS="6")):
            r = subprocess.run([exe] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert r.returncode == 2 and r.stdout == b"", (len(damaged), r.stderr[-300:])


def test_cli_gz_damaged_behind_the_first_pass_is_an_error(exe, golden_workdir, tmp_path):
    """Damage that the device inflate meets AFTER it has delivered bytes (a later pass of a large file; here: passes of 3 chunks of
    1 KB, so that the golden file spans some twenty of them): hast_gz_read_device hands over what it had decoded in front of the damage
    and reports the error with the NEXT call, as gzread does -- the program must make that call instead of taking the short block for
    the end of the file (it printed a table of the reads in front of the damage and left with 0)."""
    import shutil
    d = tmp_path / "gz"
    shutil.copytree(golden_workdir / "rand_k21", d)
    args = ["--hap0", "hap0.mer", "--hap1", "hap1.mer", "--read", "r1.fq.gz", "--read", "r2.fq.gz"]
    small = dict(os.environ, HAST_GZ_CHUNK_BYTES="1024", HAST_GZ_PASS_CHUNKS="3")
    a = subprocess.run([exe] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    b = subprocess.run([exe] + args + ["--stats"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=small)
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and len(a.stdout) > 100
    chunks = [int(l.split(b"chunks=")[1].split()[0]) for l in b.stderr.splitlines() if l.startswith(b"__stats_gz__")]
    assert len(chunks) == 2 and min(chunks) > 40, chunks                         # many passes per file
    whole = (d / "r2.fq.gz").read_bytes()
    for frac in (0.5, 0.75, 0.97):
        flipped = bytearray(whole)
        flipped[int(len(whole) * frac)] ^= 0x10
        for damaged in (whole[:int(len(whole) * frac)], bytes(flipped)):
            (d / "r2.fq.gz").write_bytes(damaged)
            for blocks in ([], ["--batch-reads", "200"], ["--devices", "0,0"]):
                r = subprocess.run([exe] + args + blocks, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=small)
                assert r.returncode == 2 and r.stdout == b"", (frac, len(damaged), blocks, r.returncode, r.stderr[-300:])


def test_cli_gz_file_larger_than_its_ring_on_the_device(exe, tmp_path):
    """A .gz file beyond 2 GB is not kept whole on the device: its bytes go round a ring whose first piece is mirrored behind its
    end (gz_api.cpp "THE RING"), every job is told in which lap its words lie, and the uploader waits for the chain before it
    overwrites anything.  Here with chunks of 16 KB, passes of 2 chunks, pieces of 64 KB (a deflate block of zlib's is at most about
    that long, and a job may read one piece past its lap) and a ring of 384-448 KB, so that two files of ~1.2 MB go round it three
    times: stdout == the same program with the whole files on the device (whose parity with the reference the golden tests pin),
    alone, with small blocks, with --devices 0,0 (striped blocks), with a decode unit per context (every piece goes to both rings),
    with whole files dealt to contexts; the ring was in use and the uploader did wait for it; damage in a later lap is an error."""
    import gzip
    rng = np.random.default_rng(2026)
    k = 21
    keys = ["".join(rng.choice(list("ACGT"), k)) for _ in range(600)]
    (tmp_path / "hap0.mer").write_text("\n".join(keys[:300]) + "\n")
    (tmp_path / "hap1.mer").write_text("\n".join(keys[300:]) + "\n")
    for name in ("r1", "r2"):
        recs = []
        for i in range(18000):
            seq = "".join(rng.choice(list("ACGT"), 100))
            if i % 3 == 0:
                at = int(rng.integers(0, 100 - k))
                seq = seq[:at] + keys[int(rng.integers(0, 600))] + seq[at + k:]
            qual = "".join(rng.choice(list("FFFFF:,#"), 100))
            recs.append("@r%d#%d_%d_%d/1\n%s\n+\n%s\n" % (i, rng.integers(1, 40), rng.integers(1, 40), rng.integers(1, 40), seq, qual))
        with gzip.open(tmp_path / (name + ".fq.gz"), "wb", compresslevel=6) as f:
            f.write("".join(recs).encode())
    assert min((tmp_path / n).stat().st_size for n in ("r1.fq.gz", "r2.fq.gz")) > 900_000
    args = ["--hap0", "hap0.mer", "--hap1", "hap1.mer", "--read", "r1.fq.gz", "--read", "r2.fq.gz"]
    ring = dict(os.environ, HAST_GZ_CHUNK_BYTES="16384", HAST_GZ_PASS_CHUNKS="2", HAST_GZ_PIECE_BYTES="65536", HAST_GZ_RING_BYTES="131072")
    a = subprocess.run([exe] + args, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert a.returncode == 0 and len(a.stdout) > 1000, a.stderr[-300:]
    for extra, env in (([], ring), (["--batch-reads", "500"], ring), (["--devices", "0,0"], ring),
                       (["--devices", "0,0"], dict(ring, HAST_GZ_SPLIT="contexts")), (["--devices", "0,0,0"], dict(ring, HAST_DEAL="files"))):
        b = subprocess.run([exe] + args + extra + ["--stats"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
        assert b.returncode == 0 and b.stdout == a.stdout, (extra, b.returncode, b.stderr[-400:])
        st = [l for l in b.stderr.splitlines() if l.startswith(b"__stats_gz__")]
        rings = [int(l.split(b"ring_bytes=")[1].split()[0]) for l in st]
        waits = [int(l.split(b"upload_waited_for_ring=")[1].split()[0]) for l in st]
        assert len(st) == 2 and all(131072 <= r <= 524288 for r in rings) and all(w > 0 for w in waits), (extra, rings, waits)
    whole = (tmp_path / "r2.fq.gz").read_bytes()
    for frac in (0.5, 0.9):
        flipped = bytearray(whole)
        flipped[int(len(whole) * frac)] ^= 0x10
        for damaged in (whole[:int(len(whole) * frac)], bytes(flipped)):
            (tmp_path / "r2.fq.gz").write_bytes(damaged)
            r = subprocess.run([exe] + args, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=ring, timeout=300)
            assert r.returncode == 2 and r.stdout == b"", (frac, len(damaged), r.returncode, r.stderr[-300:])


def test_cli_output_errors_are_not_exit_0(exe, golden_workdir):
    """stdout on a full device (the wrapper redirects it into phased.barcodes and tests only the exit status,
    classify_stlfr_reads.sh:149): exit 2 and a message, not a truncated table behind exit 0 -- `classify` and `classify_read`"""
    case, run = golden_cases("s01")[0]
    meta = load_case(case)["runs"][run]
    with open("/dev/full", "wb") as full:
        r = subprocess.run([exe] + meta["argv"], cwd=golden_workdir / case, stdout=full, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 2 and b"writing the result to stdout failed" in r.stderr, r.stderr[-300:]
    case, run = golden_cases("s03")[0]
    meta = load_case(case)["runs"][run]
    with open("/dev/full", "wb") as full:
        r = subprocess.run([hast_amd.classify_read_exe()] + meta["argv"], cwd=golden_workdir / case, stdout=full, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 2 and b"writing the result to stdout failed" in r.stderr, r.stderr[-300:]


def test_cli_device_inflate_long_barcodes_and_odd_files(exe, oracle_dir, tmp_path):
    """Device-side blocks hand the host no copy of their bytes: a barcode longer than the 15 bytes of the framer's compact copy makes
    the program fetch the block; a ".gz" that is not gzip, an empty .gz and a blocked-gzip (BGZF) file take the host route; == oracle."""
    import gzip
    rng = np.random.default_rng(12)
    k = 21
    keys = ["".join(rng.choice(list("ACGT"), k)) for _ in range(400)]
    (tmp_path / "hap0.mer").write_text("\n".join(keys[:200]) + "\n")
    (tmp_path / "hap1.mer").write_text("\n".join(keys[200:]) + "\n")
    recs = []
    for i in range(4000):
        seq = "".join(rng.choice(list("ACGT"), 150))
        if i % 3 == 0:
            kk = keys[int(rng.integers(0, 400))]
            seq = seq[:40] + kk + seq[40 + k:]
        bc = "%d_%d_%d" % (i % 50 + 1, i % 7 + 1, i % 3 + 1) if i % 11 else "a_barcode_that_is_much_longer_than_fifteen_bytes_%d" % (i % 5)
        recs.append("@r%d#%s/1\n%s\n+\n%s\n" % (i, bc, seq, "F" * 150))
    text = "".join(recs).encode()
    with gzip.open(tmp_path / "reads.fq.gz", "wb", compresslevel=6) as f:
        f.write(text)
    (tmp_path / "plain.fq.gz").write_bytes(text)                                   # not gzip: passed through, as zlib does
    (tmp_path / "empty.fq.gz").write_bytes(b"")
    (tmp_path / "reads.fq").write_bytes(text)
    if not os.path.exists(os.path.join(ROOT, "tools", "to_bgzf")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tools"), "to_bgzf"], check=True)
    subprocess.run([os.path.join(ROOT, "tools", "to_bgzf"), str(tmp_path / "reads.fq"), str(tmp_path / "blocked.fq.gz"), "2", "6"], check=True)
    base = ["--hap0", "hap0.mer", "--hap1", "hap1.mer"]
    ref = subprocess.run([os.path.join(oracle_dir, "oracle_classify")] + base + ["--read", "reads.fq.gz"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert ref.returncode == 0 and len(ref.stdout.splitlines()) > 300
    for name, n_dev in (("reads.fq.gz", 1), ("plain.fq.gz", 0), ("blocked.fq.gz", 0)):
        if not (tmp_path / name).exists():
            continue
        for extra in ([], ["--batch-reads", "40"]):
            got = subprocess.run([exe] + base + ["--read", name, "--read", "empty.fq.gz", "--stats"] + extra, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            assert got.returncode == 0, got.stderr.decode()[-2000:]
            assert got.stdout == ref.stdout, (name, extra)
            assert got.stderr.count(b"__stats_gz__") == n_dev + 1, (name, got.stderr[-500:])     # (+1: the empty file is taken by the device route)


def test_cli_long_reads_stage01_semantics(exe, oracle_dir, tmp_path):
    """Reads far longer than one kernel row (segments + whole-read N skip by a pre-pass) mixed with short ones."""
    path = _write_case(tmp_path, 1500, seed=77, gz=False, long_reads=True)
    args = ["--hap0", "hap0.mer", "--hap1", "hap1.mer", "--read", path.name]
    ref = subprocess.run([os.path.join(oracle_dir, "oracle_classify")] + args, cwd=tmp_path, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert ref.returncode == 0, ref.stderr.decode()[-500:]
    for extra in (["-t", "4"], ["-t", "3", "--batch-reads", "100"]):
        got = subprocess.run([exe] + args + extra, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert got.returncode == 0, got.stderr.decode()[-2000:]
        assert got.stdout == ref.stdout


@pytest.mark.parametrize("devices", [None, "0,0,0", pytest.param("0,1", marks=pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs"))])
@pytest.mark.parametrize("case,run", golden_cases("s03"))
def test_classify_read_cli_matches_s03_reference_golden(exe, golden_workdir, case, run, devices):
    """Drop-in for the per-read classifier of stage 03 (config 5 analogue): stdout identical to the real reference
    binary's on FASTA (multi-line) and gz FASTQ inputs with reads up to 20 kb.  --devices: every batch of reads is shared out
    between the contexts (a copy of the table each), rows stay in read order -- BASELINE config 5's read sharding."""
    meta = load_case(case)["runs"][run]
    d = golden_workdir / case
    # our loader drops the unterminated tail exactly like the reference; the fixture's hap1.mer has one
    res = subprocess.run([hast_amd.classify_read_exe()] + meta["argv"] + (["--devices", devices] if devices else []), cwd=d,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert res.stdout == open(d / meta["expected"], "rb").read()


def test_cli_save_and_load_table(exe, golden_workdir, tmp_path):
    """--save-table / --load-table: the cached key set reproduces the reference's output without the k-mer text files."""
    meta = load_case("rand_k21")["runs"]["pair_w104"]
    d = golden_workdir / "rand_k21"
    cache = str(tmp_path / "rand_k21.keys")
    r = subprocess.run([exe] + meta["argv"] + ["--save-table", cache], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and r.stdout == open(d / meta["expected"], "rb").read()
    argv = [a for a in meta["argv"]]
    for flag in ("--hap0", "--hap1"):
        i = argv.index(flag)
        del argv[i:i + 2]
    r = subprocess.run([exe] + argv + ["--load-table", cache], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    assert r.stdout == open(d / meta["expected"], "rb").read()


@pytest.mark.parametrize("fmt,gz,threads,block", [("fasta", False, 1, 0), ("fasta", False, 6, 70000), ("fasta", True, 3, 9000),
                                                  ("fastq", False, 5, 50000), ("fastq", True, 2, 0)])
def test_classify_read_parallel_ingest_matches_oracle(exe, oracle_dir, tmp_path, fmt, gz, threads, block):
    """classify_read's block ingest (records spanning blocks, multi-line FASTA, junk before the first header, empty
    lines, unterminated last line, tail FASTQ record) against the stage-03 oracle's line-by-line reader."""
    import gzip
    import random
    rng = random.Random(threads * 1000 + block)
    k = 21
    keys = [["".join(rng.choice("ACGT") for _ in range(k)) for _ in range(300)] for _ in range(2)]
    for h in (0, 1):
        (tmp_path / ("h%d.mer" % h)).write_text("\n".join(keys[h]) + "\n")
    recs = []
    for i in range(1500):
        L = rng.choice([0, 5, k, 60, 300, 3000, 12000]) if i % 7 else rng.randint(0, 400)
        s = [rng.choice("ACGT") for _ in range(L)]
        for _ in range(rng.randint(0, 1 + L // 200)):
            if L >= k:
                o = rng.randint(0, L - k)
                s[o:o + k] = rng.choice(keys[rng.randint(0, 1)])
        if L and rng.random() < 0.1:
            s[rng.randrange(L)] = rng.choice("Nn")
        recs.append(("r%d some/desc %d" % (i, i), "".join(s)))
    if fmt == "fasta":
        text = "junk line before any header\nACGT\n"
        for n, s in recs:
            w = rng.choice([60, 70, 1000000])
            text += ">" + n + "\n" + "".join(s[j:j + w] + "\n" + ("\n" if rng.random() < 0.05 else "") for j in range(0, len(s), w))
        text += ">last\nACGTACGTACGTACGTACGTACGT\nTTTT"                      # unterminated last line is dropped
    else:
        text = "".join("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)) for n, s in recs) + "@tail x\nACGTACGTACGTACGTACGTACGTA"
    name = "reads." + ("fa" if fmt == "fasta" else "fq") + (".gz" if gz else "")
    if gz:
        with gzip.open(tmp_path / name, "wb", compresslevel=1) as f:
            f.write(text.encode())
    else:
        (tmp_path / name).write_text(text)
    args = ["--hap", "h0.mer", "--hap", "h1.mer", "--read", name, "--read", name, "--format", fmt]
    ref = subprocess.run([os.path.join(oracle_dir, "oracle_classify_s03")] + args, cwd=tmp_path, stdout=subprocess.PIPE, check=True)
    env = dict(os.environ)
    if block:
        env["HAST_READ_BLOCK_BYTES"] = str(block)
    for extra in ([], ["--devices", "0,0,0,0"] if threads in (3, 5) else None):
        if extra is None:
            continue
        got = subprocess.run([hast_amd.classify_read_exe()] + args + ["--thread", str(threads)] + extra, cwd=tmp_path, env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert got.returncode == 0, got.stderr.decode()[-2000:]
        assert got.stdout == ref.stdout, extra
    assert len(ref.stdout.splitlines()) == 2 * (len(recs) + 1)


def test_cli_read_error_is_not_end_of_file(exe, golden_workdir, tmp_path):
    """a read() that FAILS (here: the input is a directory, EISDIR) must end the run with exit code 2, not pass as an empty or
    shorter input with exit code 0 -- with the GPU framer and with --host-parse"""
    d = golden_workdir / "rand_k21"
    (tmp_path / "adir.fq").mkdir()
    for extra in ([], ["--host-parse"]):
        r = subprocess.run([exe, "--hap0", "hap0.mer", "--hap1", "hap1.mer", "--read", str(tmp_path / "adir.fq")] + extra, cwd=d,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 2 and r.stdout == b"", (extra, r.returncode, r.stderr[-300:])


@pytest.mark.heavy
def test_cli_baseline_config1_full_size_vs_oracle(exe, oracle_dir, tmp_path):
    """BASELINE config 1 at its full size through the boundary: 1M synthetic 150-bp stLFR read pairs in two FASTQ files, 1M +
    1M 21-mers as k-mer text files, 10k barcodes (tools/gen_fastq writes SURVEY 8(d)'s generator to files), the wrapper's
    argument list.  stdout of the drop-in == stdout of the oracle's line-by-line restatement of classify.cpp, byte for byte."""
    import shutil
    from tests.conftest import ROOT
    gen = os.path.join(ROOT, "tools", "gen_fastq")
    if not os.path.exists(gen):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tools"), "gen_fastq"], check=True)
    d = tmp_path / "c1"
    d.mkdir()
    try:
        subprocess.run([gen, str(d), "1000000", "1000000", "10000", "21", "150", "8"], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=900)
        args = ["--hap0", "hap0.mer", "--hap1", "hap1.mer", "--thread", "8", "--weight0", "1.04", "--read", "r1.fq", "--read", "r2.fq",
                "--adaptor_f", "CTGTCTCTTATACACATCTTAGGAAGACAAGCACTGACGACATGA", "--adaptor_r", "TCTGCTGAGTCGAGAACGTCTCTGTGAGCCAAGGAGTTGCTCTGG"]
        ref = subprocess.run([os.path.join(oracle_dir, "oracle_classify")] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert ref.returncode == 0, ref.stderr.decode()[-500:]
        rows = ref.stdout.splitlines()
        assert len(rows) == 10000 and sum(1 for r in rows if r.split(b"\t")[1] in (b"0", b"1")) > 9000
        for extra in ([], ["--devices", "0,0"]):
            got = subprocess.run([exe] + args + extra, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            assert got.returncode == 0, got.stderr.decode()[-2000:]
            assert got.stdout == ref.stdout, extra
    finally:
        shutil.rmtree(d, ignore_errors=True)


@pytest.mark.heavy
def test_cli_one_million_barcodes_vs_oracle(exe, oracle_dir, tmp_path):
    """BASELINE config 2's barcode cardinality through the boundary (classify.cpp:50-64,93-102: one map entry and one output row per
    barcode): 600k synthetic read pairs over 1M barcodes -- some 450k of them seen -- as plain FASTQ and as one .gz, one context and
    two.  What only such a run reaches: the host dictionary under ~10^6 inserts, a device-side name cache that keeps learning, the
    bucketed parallel sort of the rows (below 65536 barcodes the program sorts on one thread) and their parallel formatting.
    stdout == the oracle's line-by-line restatement of classify.cpp, byte for byte (tools/gpu/cli_cardinality.sh measures the same
    at 20M reads over 1M and 10M barcodes, with the real reference binary on a subsample: profiles/round5_cli_cardinality.txt)."""
    import gzip
    import shutil
    from tests.conftest import ROOT
    gen = os.path.join(ROOT, "tools", "gen_fastq")
    if not os.path.exists(gen):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tools"), "gen_fastq"], check=True)
    d = tmp_path / "c2"
    d.mkdir()
    try:
        subprocess.run([gen, str(d), "600000", "200000", "1000000", "21", "150", "8"], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=900)
        with open(d / "r2.fq", "rb") as f, gzip.open(d / "r2.fq.gz", "wb", compresslevel=1) as g:
            shutil.copyfileobj(f, g, 1 << 24)
        args = ["--hap0", "hap0.mer", "--hap1", "hap1.mer", "--thread", "16", "--weight0", "1.04", "--read", "r1.fq"]
        ref = subprocess.run([os.path.join(oracle_dir, "oracle_classify")] + args + ["--read", "r2.fq"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert ref.returncode == 0, ref.stderr.decode()[-500:]
        rows = ref.stdout.splitlines()
        assert 400_000 < len(rows) < 500_000                  # 1M x (1 - exp(-0.6)): a pair shares its barcode
        assert rows == sorted(rows, key=lambda r: r.split(b"\t")[0])                       # byte order of the barcodes, as std::map's
        for extra in (["--read", "r2.fq"], ["--read", "r2.fq.gz"], ["--read", "r2.fq.gz", "--devices", "0,0"], ["--read", "r2.fq", "--thread", "1"]):
            got = subprocess.run([exe] + args + extra, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            assert got.returncode == 0, got.stderr.decode()[-2000:]
            assert got.stdout == ref.stdout, extra
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_cli_stats_json(exe, golden_workdir, tmp_path):
    """--stats-json FILE: every __stats_*__ line of --stats as ONE JSON object (SURVEY section 5's machine-readable summary): sections by
    name, numbers as numbers, a section printed once per .gz input as an array"""
    import json
    meta = load_case("rand_k21")["runs"]["pair_w104"]
    d = golden_workdir / "rand_k21"
    out = tmp_path / "stats.json"
    r = subprocess.run([exe] + meta["argv"] + ["--stats-json", str(out)], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    assert r.stdout == open(d / meta["expected"], "rb").read()
    js = json.load(open(out))
    assert js["stats"]["K"] == 21 and js["stats"]["reads"] > 0 and js["stats"]["barcodes"] > 0
    assert js["stats_phases"]["total_s"] >= js["stats_phases"]["read_phase_s"] >= 0
    assert js["stats_read_phase"]["records_named_on_host"] == 0
    assert js["stats_filter"]["mode"] in ("exact_entries", "prints") and js["stats_hbm"]["total_bytes"] > 1e11
    assert js["stats_dictionary"]["on"] == "device" and js["stats_dictionary"]["dictionaries"] == 1
    if any(a.endswith(".gz") for a in meta["argv"]):
        gz = js["stats_gz"] if isinstance(js["stats_gz"], list) else [js["stats_gz"]]
        assert all(g["inflated_bytes"] > g["compressed_bytes"] > 0 and g["chain_walk_s"] >= 0 for g in gz)
    # the same lines are on stderr
    assert "__stats_phases__" in r.stderr.decode()
    r = subprocess.run([exe] + meta["argv"] + ["--stats-json", "-"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and json.loads(r.stderr.decode().strip().splitlines()[-1])["stats"]["K"] == 21


def _free_bytes(path):
    st = os.statvfs(path)
    return st.f_bavail * st.f_frsize


@pytest.mark.heavy
@pytest.mark.skipif(not os.environ.get("HAST_HEAVY_FILES"), reason="opt-in (HAST_HEAVY_FILES=1 or =<read pairs>): tens of GB of files and minutes of host time")
def test_cli_baseline_size_files_through_the_boundary(exe, oracle_dir):
    """The boundary at the size the reference is built for (classify.cpp:238-278 streams inputs of any size; HAST.sh:162-166: two .fq.gz;
    BASELINE config 2: 200M reads): HAST_HEAVY_FILES=1 writes 100M read pairs (68.6 GB of FASTQ, 50M + 50M k-mers as text, 1M barcodes)
    into /dev/shm, compresses each file into ONE gzip member of 6.3 GB (tools/pgzip1: the device inflate's ring engages at its 2-GB
    default and a stream hands out > 4 GiB), and runs `classify` on the plain files, on the .gz files, over two contexts, and with
    --phase-reads (every byte routed: the routed files add up to the inputs).  HAST_HEAVY_FILES=<pairs> scales it; HAST_HEAVY_ORACLE=1
    adds the oracle's program over the same files (minutes on the host's cores).  Skipped where /dev/shm lacks the room.
    tools/gpu/cli_c2.sh is the same job with timings (profiles/round6_cli_c2.txt)."""
    import shutil
    import tempfile
    from tests.conftest import ROOT
    v = os.environ["HAST_HEAVY_FILES"]
    pairs = 100_000_000 if v == "1" else int(v)
    need = pairs * 2 * 343 * 2.3 + 3e9                       # FASTQ + .gz + one routed copy
    if _free_bytes("/dev/shm") < need:
        pytest.skip("needs %.0f GB in /dev/shm" % (need / 1e9))
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tools"), "gen_fastq", "pgzip1"], check=True)
    d = tempfile.mkdtemp(prefix="hast_heavy.", dir="/dev/shm")
    try:
        keys = min(50_000_000, max(1000, pairs // 2))
        subprocess.run([os.path.join(ROOT, "tools", "gen_fastq"), d, str(pairs), str(keys), "1000000", "21", "150", "32"], check=True, timeout=1200,
                       env=dict(os.environ, GEN_FASTQ_MAX_GB="90"))
        for f in ("r1.fq", "r2.fq"):
            subprocess.run([os.path.join(ROOT, "tools", "pgzip1"), os.path.join(d, f), os.path.join(d, f + ".gz"), "6", "16", "32"], check=True, timeout=1800)
        args = ["--hap0", "hap0.mer", "--hap1", "hap1.mer", "--thread", "32", "--weight0", "1.04", "--stats"]
        plain, gz = ["--read", "r1.fq", "--read", "r2.fq"], ["--read", "r1.fq.gz", "--read", "r2.fq.gz"]
        ref = subprocess.run([exe] + args + plain, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
        assert ref.returncode == 0, ref.stderr.decode()[-2000:]
        assert len(ref.stdout.splitlines()) > 1000
        for extra in (gz, gz + ["--devices", "0,0"]):
            got = subprocess.run([exe] + args + extra, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
            assert got.returncode == 0, got.stderr.decode()[-2000:]
            assert got.stdout == ref.stdout, extra
            if os.path.getsize(os.path.join(d, "r1.fq.gz")) > 2 << 30:
                lines = [l for l in got.stderr.decode().splitlines() if l.startswith("__stats_gz__")]
                assert len(lines) == 2 and all("ring_bytes=2147483648" in l and "ring_laps=0 " not in l for l in lines), lines
        w = os.path.join(d, "w")
        os.mkdir(w)
        got = subprocess.run([exe] + args + [a if not a.endswith(".gz") else os.path.join("..", a) for a in gz] + ["--phase-reads", "--hap0", "../hap0.mer", "--hap1", "../hap1.mer"],
                             cwd=w, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
        assert got.returncode == 0, got.stderr.decode()[-2000:]
        assert got.stdout == ref.stdout
        for f in ("r1.fq", "r2.fq"):
            routed = sum(os.path.getsize(os.path.join(w, n)) for n in os.listdir(w) if n.startswith(f + ".") and n.endswith(".fastq"))
            assert routed == os.path.getsize(os.path.join(d, f)), f
        shutil.rmtree(w)
        if os.environ.get("HAST_HEAVY_ORACLE"):
            o = subprocess.run([os.path.join(oracle_dir, "oracle_classify")] + args[:-1] + plain, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=3000)
            assert o.returncode == 0 and o.stdout == ref.stdout
    finally:
        shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("extra,env", [([], {}), (["--route", "host"], {}), (["--devices", "0,0"], {}), (["--devices", "0,0,0", "--batch-reads", "150"], {}),
                                       (["--batch-reads", "40"], {}), ([], {"HAST_PHASE_READS": "1", "_no_flag": "1"}),
                                       (["--inflate", "host"], {})])
def test_cli_phase_reads_does_the_wrappers_steps_10_and_11(exe, golden_workdir, tmp_path, extra, env):
    """--phase-reads (HAST_PHASE_READS=1): `classify` writes the three barcode lists and routes every record of every input itself --
    steps 10 and 11 of classify_stlfr_reads.sh:155-190, which the unchanged wrapper then skips (step_10_done / step_11_done) -- ON THE
    GPU: the inputs go through the framer a second time (.gz inputs inflated on the device again), k_route_class / _scan / _copy sort
    the records of a block into four runs by the class of their barcode, the host only writes (--route host: the worker threads parse
    every record again, quartering.h).  Expected files: the lists as the wrapper's awk one-liners derive them from stdout, the routing
    as the stand-alone quartering_fastq does it from those lists (tests/test_quartering_cpu.py pins that program to the reference's
    awk program byte for byte), one input plain and one .gz, as the wrapper would call it; blocks of 40 records (every block border
    inside a record), several contexts of one GPU (the blocks of a file dealt to them in turn)."""
    import gzip
    import shutil
    from tests.conftest import ROOT
    qf = os.path.join(ROOT, "hast_amd", "quartering_fastq")
    a, b = tmp_path / "a", tmp_path / "b"
    for d in (a, b):
        shutil.copytree(golden_workdir / "rand_k21", d)
        with gzip.open(d / "r1.fq.gz") as f, open(d / "r1.fq", "wb") as g:
            shutil.copyfileobj(f, g)
    args = ["--hap0", "hap0.mer", "--hap1", "hap1.mer", "--weight0", "1.04", "--read", "r1.fq", "--read", "r2.fq.gz", "--thread", "5"]
    ref = subprocess.run([exe] + args, cwd=b, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert ref.returncode == 0 and not (b / "step_10_done").exists()
    env = dict(env)
    flag = [] if env.pop("_no_flag", None) else ["--phase-reads"]
    got = subprocess.run([exe] + args + ["--stats"] + flag + extra, cwd=a, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
    assert got.returncode == 0, got.stderr.decode()[-1000:]
    assert got.stdout == ref.stdout and b"__stats_phase_reads__" in got.stderr
    line = [l for l in got.stderr.decode().splitlines() if l.startswith("__stats_phase_reads__")][0]
    kv = dict(x.split("=") for x in line.split()[1:])
    if "--route" in extra:
        assert kv["route"] == "host"
    else:
        assert kv["route"] == "device" and int(kv["blocks_routed_on_device"]) >= 2 and int(kv["blocks_routed_by_host"]) == 0, line
        if "--batch-reads" in extra:
            assert int(kv["blocks_routed_on_device"]) >= 10, line
    assert (a / "step_10_done").exists() and (a / "step_11_done").exists()
    # the wrapper's own steps, in b
    rows = [r.split(b"\t") for r in ref.stdout.splitlines()]
    for name, hap in (("paternal", b"0"), ("maternal", b"1"), ("homozygous", b"-1")):
        (b / (name + ".unique.barcodes")).write_bytes(b"".join(r[0] + b"\n" for r in rows if r[1] == hap))
    lists = ["paternal.unique.barcodes", "maternal.unique.barcodes", "homozygous.unique.barcodes"]
    r1 = subprocess.run([qf, "--prefix", "r1.fq"] + lists + ["r1.fq"], cwd=b, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    r2 = subprocess.run([qf, "--prefix", "r2.fq"] + lists + ["-"], cwd=b, input=gzip.open(b / "r2.fq.gz").read(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r1.returncode == 0 and r2.returncode == 0
    made = sorted(p.name for p in b.iterdir() if p.name.endswith((".fastq", ".barcodes", "filter_reads.log")))
    assert len([m for m in made if m.endswith(".fastq")]) >= 6
    for m in made:
        assert (a / m).exists() and (a / m).read_bytes() == (b / m).read_bytes(), m
    assert sorted(p.name for p in a.iterdir() if p.name.endswith(".fastq")) == [m for m in made if m.endswith(".fastq")]


def _edge_fastq(k=21, seed=5):
    """records whose headers take every branch of quartering_fastq.awk:21-49 and the places where awk's field rule ('#' or '/' splits)
    and parseName's (last '#', last '/': classify.cpp:112-119) part ways"""
    rng = np.random.default_rng(seed)
    def seq(n=60):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    heads = ["@a1#1_2_3/1", "@a2#0_0_0/1", "@a3", "@a4/x#5_6_7/1", "@a5#averyveryverylongbarcode_123/1", "@a6#1_2_3", "@a7#/1", "@a8#7_8_9/2\t5\t1",
             "@a9#1_2_3/1", "@b1/1", "@b2#0_0/1", "@b3#0/1", "@b4#exactly15bytes_/1", "@b5#sixteen_bytes_xx/1", "@b6#4_4_4#5_5_5/1", "@b7#0_0_0"]
    recs = []
    for i in range(400):
        h = heads[i % len(heads)]
        recs.append("%s\n%s\n+\n%s\n" % (h, seq(), "F" * 60))
    return recs


@pytest.mark.parametrize("extra", [[], ["--batch-reads", "7"], ["--devices", "0,0"], ["--devices", "0,0,0", "--batch-reads", "16"]])
@pytest.mark.parametrize("tail", ["", "@t1#1_2_3/1\nACGTACGTACGTACGTACGTACGTACGT", "@t2#9_9_9/1\nACGTACGTACGTACGTACGTACGTACGT\n+\nFFFF", "@t3#1_2_3/1"])
def test_cli_phase_reads_device_equals_host_router_on_every_awk_branch(exe, golden_workdir, tmp_path, extra, tail):
    """the device router against the host router (which tests/test_quartering_cpu.py pins to the awk program) on headers without a
    field 2, with "0_0_0", with a field 2 that parseName does not take for the barcode (in no list: the ERROR line and the dropped
    record), with fields of 15 and 16 bytes and longer (a block with such a record comes back to the host), an empty field, and a
    file that ends inside a record (awk prints the lines it has) -- plain and .gz, small blocks, several contexts"""
    import gzip
    import shutil
    recs = _edge_fastq()
    text = ("".join(recs) + tail).encode()
    outs = {}
    for mode in ("device", "host"):
        d = tmp_path / mode
        shutil.copytree(golden_workdir / "rand_k21", d)
        (d / "e1.fq").write_bytes(text)
        with gzip.open(d / "e2.fq.gz", "wb") as g:
            g.write("".join(recs[:150]).encode() if not tail else text)
        args = ["--hap0", "hap0.mer", "--hap1", "hap1.mer", "--read", "e1.fq", "--read", "e2.fq.gz", "--thread", "3", "--stats", "--phase-reads", "--route", mode]
        r = subprocess.run([exe] + args + extra, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            # (a file that ends in a header line without its bases: classify itself refuses nothing here; anything else is a failure)
            assert False, r.stderr.decode()[-1500:]
        err = [l for l in r.stderr.decode().splitlines() if l.startswith("ERROR : unclassify")]
        files = {p.name: p.read_bytes() for p in d.iterdir() if p.name.endswith((".fastq", ".barcodes", "filter_reads.log"))}
        outs[mode] = (r.stdout, err, files)
        line = [l for l in r.stderr.decode().splitlines() if l.startswith("__stats_phase_reads__")][0]
        assert ("route=" + mode) in line, line
        if mode == "device":
            assert "blocks_routed_by_host=0" not in line, line          # (long fields and unlisted barcodes: those blocks are the host's)
    assert outs["device"][0] == outs["host"][0]
    assert outs["device"][1] == outs["host"][1] and len(outs["host"][1]) >= 20          # "x" of @a4/x#... is in no list
    assert sorted(outs["device"][2]) == sorted(outs["host"][2])
    for name, data in outs["host"][2].items():
        assert outs["device"][2][name] == data, name
    assert any(n.endswith(".nobarcode.fastq") for n in outs["host"][2]) and any(n.endswith(".homozygous.fastq") for n in outs["host"][2])


def test_cli_phase_reads_barcode_with_a_separator_goes_to_the_host_router(exe, golden_workdir, tmp_path):
    """a barcode that itself holds '/' ("#1_2_3/1/2": parseName takes "1_2_3/1") makes a list line that awk cuts at its first field:
    the device's table text -> class cannot stand for awk's three arrays then, and the whole run is routed by the host"""
    import shutil
    d = tmp_path / "w"
    shutil.copytree(golden_workdir / "rand_k21", d)
    recs = _edge_fastq()
    recs.append("@c1#1_2_3/1/2\n" + "ACGT" * 15 + "\n+\n" + "F" * 60 + "\n")
    (d / "e1.fq").write_bytes("".join(recs).encode())
    r = subprocess.run([exe, "--hap0", "hap0.mer", "--hap1", "hap1.mer", "--read", "e1.fq", "--stats", "--phase-reads"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    line = [l for l in r.stderr.decode().splitlines() if l.startswith("__stats_phase_reads__")][0]
    assert "route=host" in line and (d / "e1.fq.homozygous.fastq").exists()

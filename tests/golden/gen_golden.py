#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the build container only).

Every case directory holds the INPUTS we made up here (k-mer text files, FASTQ files, the
argv in case.json) and the EXPECTED stdout produced by the REAL reference binary
`oracle/_ref/classify` (compiled by oracle/Makefile straight from
/root/reference/01.classify_stlfr_reads/{classify.cpp,gzstream/gzstream.C}, shipped flags).
Nothing of the reference's source is stored; fixtures are data only.

    python tests/golden/gen_golden.py          # rewrites tests/golden/<case>/...

The reference is the sole author of every expected.tsv; tests then require
  oracle (CPU restatement) == expected.tsv     (tests/test_oracle_golden.py, CPU)
  product classify (HIP)   == expected.tsv     (tests/test_cli_gpu.py, GPU)
"""
import gzip
import json
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "classify")
ADAPTOR_F = "CTGTCTCTTATACACATCTTAGGAAGACAAGCACTGACGACATGA"
ADAPTOR_R = "TCTGCTGAGTCGAGAACGTCTCTGTGAGCCAAGGAGTTGCTCTGG"
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def rc(s):
    return "".join(COMP[c] for c in reversed(s))


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def write(path, data, gz=False):
    if gz:
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(data.encode())
    else:
        with open(path, "w") as f:
            f.write(data)


def run_ref(case_dir, args):
    out = subprocess.run([REF] + args, cwd=case_dir, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    return out.stdout.decode(), out.stderr.decode()


def finish_case(name, files, runs):
    """files: {fname: (text, gz?)}; runs: {run_name: argv list (relative file names)}"""
    d = os.path.join(HERE, name)
    os.makedirs(d, exist_ok=True)
    for fn, (text, gz) in files.items():
        write(os.path.join(d, fn), text, gz)
    meta = {"runs": {}}
    for rn, argv in runs.items():
        stdout, stderr = run_ref(d, argv)
        exp = "expected.%s.tsv" % rn
        write(os.path.join(d, exp), stdout)
        sizes = [l for l in stderr.splitlines() if l.startswith("Recorded") or "erase a adaptor" in l]
        meta["runs"][rn] = {"argv": argv, "expected": exp, "ref_log": sizes}
    with open(os.path.join(d, "case.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", name, {k: len(open(os.path.join(d, v["expected"])).read().splitlines())
                          for k, v in meta["runs"].items()})


def fq(records, trailing_newline=True):
    s = "".join("%s\n%s\n+\n%s\n" % (h, q, "F" * len(q)) for h, q in records)
    return s if trailing_newline else s[:-1]


# ------------------------------------------------------------------------------------------
def case_edge():
    """Hand-made K=7 case covering SURVEY Appendix A.3 traps."""
    k = 7
    p_only = ["ACGTACC", "GGATCCA", "TTGACCA", "CAGTTGA"]
    m_only = ["ACCGGTA", "TGCATGG", "GATTACA", "CCATGCA"]
    both = ["AGGCTTA"]
    # hap0: duplicate line, reverse-complement duplicate, lower-case line, key shared with hap1,
    #       an adaptor k-mer (erased by InitAdaptor), unterminated last line (dropped)
    hap0_lines = p_only + [p_only[0], rc(p_only[1]), p_only[2].lower()] + both + [ADAPTOR_F[3:3 + k]]
    hap0 = "\n".join(hap0_lines) + "\n" + "TTTTTTT"          # last piece unterminated -> dropped
    hap1_lines = m_only + both + [rc(ADAPTOR_R[5:5 + k])]
    hap1 = "\n".join(hap1_lines) + "\n"
    filler = "AAAAAAAAAA"  # poly-A: canonical AAAAAAA, in neither set
    recs = [
        ("@V1#1_1_1/1", filler + p_only[0] + filler),                  # fwd hit hap0
        ("@V2#1_1_1/2", filler + rc(p_only[1]) + filler),              # rc-strand hit hap0
        ("@V3#2_2_2/1\tx/y\t1", filler + m_only[0] + filler),          # extra '/' and tab fields
        ("@V4#0_0_0/1", filler + p_only[2] + filler),                  # 0_0_0 always -1, counts printed
        ("@V5#3_3_3/1", filler + p_only[0] + "N" + m_only[1]),         # N => whole read skipped
        ("@V6#3_3_3/2", (filler + m_only[1] + filler).lower()),        # lower-case read still hits
        ("@V7#4_4_4/1", "acgtn" + "ACGTACC" + "nnnn"),                 # lower-case n is NOT a skip
        ("@V8#5_5_5/1", filler + filler),                              # zero-hit barcode still printed
        ("@V9#6_6_6/1", m_only[2]),                                    # len == K read
        ("@V10#7_7_7/1", filler + both[0] + filler),                   # key in both sets => tie => -1
        ("@V11#8_8_8/1", p_only[3] + p_only[3] + p_only[3]),           # repeated positions count each time
        ("@V12#8_8_8/1", filler + m_only[3] + filler),
        ("@noBarcode/1", filler + p_only[0]),                          # no '#': barcode starts at 0
        ("@V13#9_9_9", filler + m_only[0]),                            # no '/': to end of line
        ("@a/b#10_10_10", filler + p_only[1]),                         # '/' before '#': to end of line
        ("@V14#0_0/1", filler + p_only[0]),                            # "0_0"
        ("@V15#0/1", filler + m_only[0]),                              # "0"
        ("@V16#11_11_11/1", ADAPTOR_F),                                # adaptor k-mers were erased
        ("@V17#11_11_11/2", ADAPTOR_R),
        ("@V18#100_1_1/1", filler + p_only[0]),                        # byte-wise row order
        ("@V19#1000_1_1/1", filler + m_only[0]),
        ("@V20#12_12_12/1", filler + p_only[0] + p_only[1] + m_only[0] + filler),
    ]
    files = {
        "hap0.mer": (hap0, False),
        "hap1.mer": (hap1, False),
        "r1.fq": (fq(recs), False),
        "r1_nonl.fq": (fq(recs, trailing_newline=False), False),       # last line unterminated
        "r2.fq.gz": (fq(recs[:9]), True),
    }
    base = ["--hap0", "hap0.mer", "--hap1", "hap1.mer"]
    runs = {
        "plain": base + ["--read", "r1.fq", "-t", "2"],
        "nonl": base + ["--read", "r1_nonl.fq", "-t", "1"],
        "two_files_w104": base + ["--read", "r1.fq", "--read", "r2.fq.gz", "--thread", "3", "--weight0", "1.04"],
        "weight1": base + ["-r", "r1.fq", "-u", "2.5", "-t", "8"],
        "custom_adaptors": ["-p", "hap0.mer", "-m", "hap1.mer", "-r", "r1.fq", "-t", "4",
                            "-f", filler + p_only[0] + filler, "-q", "GATTACAGATTACA"],
    }
    finish_case("edge_k7", files, runs)


def random_case(name, k, n_keys, n_pairs, n_barcodes, seed, read_len=100, extra_runs=True):
    rng = random.Random(seed)
    keys = [[], []]
    for h in range(2):
        for _ in range(n_keys):
            s = rand_seq(rng, k)
            keys[h].append(s if rng.random() < 0.5 else rc(s))   # files are not canonical-only
    shared = [rand_seq(rng, k) for _ in range(max(4, n_keys // 100))]
    keys[0] += shared
    keys[1] += [rc(s) for s in shared]
    keys[0] += keys[0][:5]                                       # duplicates
    keys[1] += [rc(s) for s in keys[1][:5]]
    keys[0].append(ADAPTOR_F[2:2 + k]) if k <= 40 else None
    keys[1].append(rc(ADAPTOR_R[1:1 + k]))
    rng.shuffle(keys[1])
    first = keys[0][0]
    rest = keys[0][1:]
    rng.shuffle(rest)
    keys[0] = [first] + rest
    barcodes = ["0_0_0"] + ["%d_%d_%d" % (rng.randint(1, 1536), rng.randint(1, 1536), rng.randint(1, 1536))
                            for _ in range(n_barcodes - 1)]
    truth = [rng.randint(0, 1) for _ in barcodes]
    recs = [[], []]
    for i in range(n_pairs):
        b = rng.randrange(len(barcodes)) if rng.random() > 0.1 else 0
        for mate in range(2):
            L = read_len if rng.random() < 0.9 else rng.randint(k, read_len)
            s = list(rand_seq(rng, L))
            for _ in range(rng.randint(0, 3)):
                h = truth[b] if rng.random() < 0.8 else 1 - truth[b]
                km = rng.choice(keys[h])
                if rng.random() < 0.5:
                    km = rc(km)
                if L > k:
                    off = rng.randint(0, L - k)
                    s[off:off + k] = list(km)
            if rng.random() < 0.01:
                s[rng.randrange(L)] = "N"
            if rng.random() < 0.01:
                s = [c.lower() for c in s]
            if rng.random() < 0.005 and L >= 45:
                s[0:45] = list(ADAPTOR_F)[:45]
            recs[mate].append(("@V300R%09d#%s/%d\t%d\t1" % (i, barcodes[b], mate + 1, i), "".join(s)))
    files = {
        "hap0.mer": ("\n".join(keys[0]) + "\n", True),
        "hap1.mer": ("\n".join(keys[1]) + "\n", True),
        "r1.fq.gz": (fq(recs[0]), True),
        "r2.fq.gz": (fq(recs[1]), True),
    }
    # k-mer files must be plain text for the reference (std::ifstream): tests gunzip them to a
    # temp dir; they are stored compressed only to keep the repository small.
    d = os.path.join(HERE, name)
    os.makedirs(d, exist_ok=True)
    write(os.path.join(d, "hap0.mer"), files["hap0.mer"][0])
    write(os.path.join(d, "hap1.mer"), files["hap1.mer"][0])
    base = ["--hap0", "hap0.mer", "--hap1", "hap1.mer"]
    runs = {"pair_w104": base + ["--read", "r1.fq.gz", "--read", "r2.fq.gz", "--thread", "8", "--weight0", "1.04"]}
    if extra_runs:
        runs["single_w1"] = base + ["--read", "r2.fq.gz", "-t", "3"]
        runs["wrapper_argv"] = WRAPPER_ARGV
    try:
        finish_case(name, {k_: v for k_, v in files.items() if k_.startswith("r")}, runs)
    finally:
        for fn in ("hap0.mer", "hap1.mer"):
            p = os.path.join(d, fn)
            write(p + ".gz", open(p).read(), gz=True)
            os.remove(p)


# The literal command line of the stage-01 wrapper (classify_stlfr_reads.sh:44-45,142-149): --hap0/--hap1, --thread, --weight0 1.04,
# one --read per filial file, and the LONG --adaptor_f/--adaptor_r spellings with the wrapper's default sequences.
WRAPPER_ARGV = ["--hap0", "hap0.mer", "--hap1", "hap1.mer", "--thread", "8", "--weight0", "1.04", "--read", "r1.fq.gz", "--read", "r2.fq.gz",
                "--adaptor_f", ADAPTOR_F, "--adaptor_r", ADAPTOR_R]


def case_wrapper_argv():
    """adds the run `wrapper_argv` to the rand_k21 case as it is committed (inputs untouched)"""
    import shutil
    import tempfile
    d = os.path.join(HERE, "rand_k21")
    tmp = tempfile.mkdtemp()
    try:
        for fn in os.listdir(d):
            shutil.copy(os.path.join(d, fn), tmp)
        for fn in ("hap0.mer", "hap1.mer"):
            open(os.path.join(tmp, fn), "wb").write(gzip.open(os.path.join(tmp, fn + ".gz")).read())
        stdout, stderr = run_ref(tmp, WRAPPER_ARGV)
    finally:
        shutil.rmtree(tmp)
    write(os.path.join(d, "expected.wrapper_argv.tsv"), stdout)
    meta = json.load(open(os.path.join(d, "case.json")))
    meta["runs"]["wrapper_argv"] = {"argv": WRAPPER_ARGV, "expected": "expected.wrapper_argv.tsv",
                                    "ref_log": [l for l in stderr.splitlines() if l.startswith("Recorded") or "erase a adaptor" in l]}
    with open(os.path.join(d, "case.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote rand_k21/wrapper_argv:", len(stdout.splitlines()), "rows")


def case_s03(name, k, n_keys, seed, max_len):
    """Per-read classifier of stage 03 (config 5 analogue): FASTA (multi-line) and FASTQ, long reads."""
    ref = os.path.join(ROOT, "oracle", "_ref", "classify_s03")
    rng = random.Random(seed)
    keys = [[rand_seq(rng, k) for _ in range(n_keys)] for _ in range(2)]
    keys[1][:3] = [rc(s) for s in keys[0][:3]]                    # shared between the sets (via reverse complement)
    keys[0] += keys[0][:4]                                        # duplicate lines count in the denominator
    reads = []
    for i in range(40):
        L = rng.choice([k - 1, k, 50, 300, 2000, max_len]) if i > 5 else max_len
        s = list(rand_seq(rng, L))
        favoured = rng.randint(0, 1)
        for _ in range(rng.randint(0, 12)):
            if L < k:
                break
            h = favoured if rng.random() < 0.7 else 1 - favoured
            km = rng.choice(keys[h])
            km = rc(km) if rng.random() < 0.5 else km
            o = rng.randint(0, L - k)
            s[o:o + k] = list(km)
        for _ in range(rng.randint(0, 3)):
            if L:
                s[rng.randrange(L)] = rng.choice("NnacgtR")        # windows over these bytes simply miss
        reads.append(("read%d some text/%d" % (i, i), "".join(s)))
    fa = "".join(">%s\n%s\n\n" % (h, "\n".join(q[j:j + 70] for j in range(0, len(q), 70))) for h, q in reads)
    fqs = "".join("@%s\n%s\n+\n%s\n" % (h, q, "I" * len(q)) for h, q in reads if len(q) >= 1)
    d = os.path.join(HERE, name)
    os.makedirs(d, exist_ok=True)
    write(os.path.join(d, "hap0.mer"), "\n".join(keys[0]) + "\n")
    write(os.path.join(d, "hap1.mer"), "\n".join(keys[1]) + "\n" + "ACGT")   # unterminated tail dropped
    write(os.path.join(d, "reads.fa"), fa)
    write(os.path.join(d, "reads.fq.gz"), fqs, gz=True)
    meta = {"runs": {}}
    for rn, argv in {"fasta": ["--hap", "hap0.mer", "--hap", "hap1.mer", "--read", "reads.fa", "--thread", "3"],
                     "fastq_gz": ["--hap", "hap0.mer", "--hap", "hap1.mer", "--read", "reads.fq.gz", "--format", "fastq",
                                  "--read", "reads.fq.gz"]}.items():
        out = subprocess.run([ref] + argv, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        write(os.path.join(d, "expected.%s.tsv" % rn), out.stdout.decode())
        meta["runs"][rn] = {"argv": argv, "expected": "expected.%s.tsv" % rn, "program": "s03"}
    with open(os.path.join(d, "case.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    for fn in ("hap0.mer", "hap1.mer", "reads.fa"):
        pth = os.path.join(d, fn)
        write(pth + ".gz", open(pth).read(), gz=True)
        os.remove(pth)
    print("wrote", name)


def case_s03_edge():
    """Corner cases of the stage-03 call logic from the real binary: exact tie, no hits, one-sided hits, read shorter
    than K, lower-case read, empty FASTA lines, shared k-mer."""
    ref = os.path.join(ROOT, "oracle", "_ref", "classify_s03")
    a, b, c, d = "ACGTTGCATCGATTGCAAGTT", "GGATCCATTAGGCATCGATCA", "TTTGACCAGTAGGCATGCATG", "CATGCATGCCCGGGAAATTTA"
    pad = "AAAAAAAAAAAAAAAAAAAAAAAAA"
    reads = [("tie", pad + a + pad + b + pad), ("tie_rc", pad + rc(a) + pad + rc(d) + pad), ("none", pad * 3),
             ("only1", pad + d + pad), ("only0", c), ("short", "ACGT"), ("two0_one1", a + pad + c + pad + b),
             ("lower", (pad + a).lower()), ("withN", pad + a[:10] + "N" + a[11:] + pad + b)]
    d_ = os.path.join(HERE, "s03_edge")
    os.makedirs(d_, exist_ok=True)
    write(os.path.join(d_, "hap0.mer"), a + "\n" + c + "\n")
    write(os.path.join(d_, "hap1.mer"), b + "\n" + d + "\n")
    write(os.path.join(d_, "reads.fa"), "".join(">%s desc\n%s\n\n%s\n" % (n, q[:30], q[30:]) for n, q in reads))
    argv = ["--hap", "hap0.mer", "--hap", "hap1.mer", "--read", "reads.fa"]
    out = subprocess.run([ref] + argv, cwd=d_, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert out.returncode == 0
    write(os.path.join(d_, "expected.fasta.tsv"), out.stdout.decode())
    with open(os.path.join(d_, "case.json"), "w") as f:
        json.dump({"runs": {"fasta": {"argv": argv, "expected": "expected.fasta.tsv", "program": "s03"}}}, f, indent=1, sort_keys=True)
    print("wrote s03_edge:", out.stdout.decode().replace("\n", " | "))


def case_quartering():
    """Step 10-11 of classify_stlfr_reads.sh (:155-190) run with the reference's own awk program on the rand_k21
    inputs: barcode lists from the reference's phased.barcodes, then quartering_fastq.awk.  Stored: md5 + size of
    each routed FASTQ and the text appended to filter_reads.log (the routed files are as large as the input)."""
    import hashlib
    import shutil
    import tempfile
    awk_prog = "/root/reference/01.classify_stlfr_reads/quartering_fastq.awk"
    src = os.path.join(HERE, "rand_k21")
    tmp = tempfile.mkdtemp()
    try:
        phased = open(os.path.join(src, "expected.pair_w104.tsv")).read()
        # add rows the FASTQ never mentions + make one FASTQ barcode unclassified (exercises the ERROR path)
        lines = phased.splitlines()
        dropped = lines.pop(7).split("\t")[0]
        open(os.path.join(tmp, "phased.barcodes"), "w").write("\n".join(lines) + "\n")
        for name, val in (("paternal", "0"), ("maternal", "1"), ("homozygous", "-1")):
            subprocess.run("awk '{if($2 == %s) print $1;}' phased.barcodes > %s.unique.barcodes"
                           % (val if val != "-1" else '"-1"', name), shell=True, cwd=tmp, check=True)
        out = {"dropped_barcode": dropped, "files": {}}
        for fq_name in ("r1.fq", "r2.fq"):
            data = gzip.open(os.path.join(src, fq_name + ".gz")).read()
            if fq_name == "r2.fq":
                data = data[:-1] + b"\n@tail#%s/2\nACGT" % lines[3].split("\t")[0].encode()   # partial last record, unterminated
            open(os.path.join(tmp, fq_name), "wb").write(data)
            r = subprocess.run(["awk", "-v", "prefix=" + fq_name, "-F", "#|/", "-f", awk_prog, "paternal.unique.barcodes",
                                "maternal.unique.barcodes", "homozygous.unique.barcodes", fq_name], cwd=tmp,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
            out["files"][fq_name] = {"stderr_lines": len(r.stderr.decode().splitlines()),
                                     "stderr_md5": hashlib.md5(r.stderr).hexdigest()}
            for cls in ("paternal", "maternal", "homozygous", "nobarcode"):
                pth = os.path.join(tmp, "%s.%s.fastq" % (fq_name, cls))
                if os.path.exists(pth):
                    b = open(pth, "rb").read()
                    out["files"][fq_name][cls] = {"md5": hashlib.md5(b).hexdigest(), "bytes": len(b)}
        out["filter_reads_log"] = open(os.path.join(tmp, "filter_reads.log")).read()
        # tiny hand-made case with every branch of the awk program; inputs and outputs stored in full
        e = os.path.join(tmp, "edge")
        os.makedirs(e)
        edge_in = {"p.bc": "1_1_1\n2_2_2\n", "m.bc": "3_3_3\n1_1_1\n", "h.bc": "4_4_4\n\n5_5_5/x\n",
                   "e.fq": "@r1#1_1_1/1\tx\nACGT\n+\nFFFF\n@r2#3_3_3/2\nAC\n+\nFF\n@r3#0_0_0/1\nA\n+\nF\n@r4#9_9_9/1\nA\n+\nF\n"
                           "@r5\nA\n+\nF\n@r6#4_4_4\nAA\n+\nFF\n@r7##/1\nA\n+\nF\n@r8/5_5_5#z\nC\n+\nF\n@r9#2_2_2/1\nG\n+"}
        for fn, txt in edge_in.items():
            open(os.path.join(e, fn), "w").write(txt)
        r = subprocess.run(["awk", "-v", "prefix=e.fq", "-F", "#|/", "-f", awk_prog, "p.bc", "m.bc", "h.bc", "e.fq"], cwd=e,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        out["edge"] = {"inputs": edge_in, "stderr": r.stderr.decode(),
                       "outputs": {fn: open(os.path.join(e, fn)).read() for fn in sorted(os.listdir(e)) if fn not in edge_in}}
        out["r2_tail"] = "@tail#%s/2\nACGT" % lines[3].split("\t")[0]
        d = os.path.join(HERE, "quartering")
        os.makedirs(d, exist_ok=True)
        for name in ("paternal", "maternal", "homozygous"):
            shutil.copy(os.path.join(tmp, name + ".unique.barcodes"), os.path.join(d, name + ".unique.barcodes"))
        with open(os.path.join(d, "expected.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        print("wrote quartering", {k: sorted(v) for k, v in out["files"].items()})
    finally:
        shutil.rmtree(tmp)


# ------------------------------------------------------------------------------------------
# stage 00: parent-unique k-mer sets.  The REAL reference script is run where it lies (it calls the jellyfish binary
# vendored next to it); stored: our inputs, and its final products with the .mer lines SORTED (their order in the
# reference is jellyfish's hash order, which nothing downstream depends on).
S00_SCRIPT = "/root/reference/00.build_unshare_kmers_by_jellyfish/build_unshared_kmers.sh"


def mutate(rng, g, rate):
    g = list(g)
    for i in range(len(g)):
        if rng.random() < rate:
            g[i] = rng.choice([c for c in "ACGT" if c != g[i]])
    return "".join(g)


def sample_reads(rng, genome, cov, L, err, n_rate=0.02, lower_rate=0.03):
    out = []
    for i in range(int(len(genome) * cov / L)):
        s = rng.randrange(len(genome) - L)
        q = [(rng.choice("ACGT") if rng.random() < err else c) for c in genome[s:s + L]]
        q = "".join(q)
        if rng.random() < 0.5:
            q = rc(q)
        if rng.random() < n_rate:
            j = rng.randrange(L)
            q = q[:j] + "N" + q[j + 1:]
        if rng.random() < lower_rate:
            q = q.lower()
        out.append(q)
    return out


def s00_finish(name, files, runs):
    """files: {fname: (text, gz?)}; runs: {run: argv}.  Runs the reference script in a scratch copy."""
    import shutil
    import tempfile
    d = os.path.join(HERE, name)
    os.makedirs(d, exist_ok=True)
    for fn, (text, gz) in files.items():
        write(os.path.join(d, fn), text, gz)
    meta = {"runs": {}}
    for rn, argv in runs.items():
        tmp = tempfile.mkdtemp()
        try:
            for fn in files:
                shutil.copy(os.path.join(d, fn), tmp)
            r = subprocess.run(["bash", S00_SCRIPT] + argv, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=3600)
            assert r.returncode == 0, r.stdout.decode()[-3000:]
            rec = {"argv": argv, "program": "s00", "products": {}}
            for prod in ("paternal.unique.filter.mer", "maternal.unique.filter.mer"):
                lines = sorted(open(os.path.join(tmp, prod)).read().splitlines())
                exp = "expected.%s.%s" % (rn, prod)
                write(os.path.join(d, exp), "".join(l + "\n" for l in lines))
                rec["products"][prod] = {"expected": exp, "sorted": True, "lines": len(lines)}
            for prod in ("maternal.histo", "paternal.histo", "maternal.bounds.txt", "paternal.bounds.txt"):
                pth = os.path.join(tmp, prod)
                if os.path.exists(pth):
                    exp = "expected.%s.%s" % (rn, prod)
                    shutil.copy(pth, os.path.join(d, exp))
                    rec["products"][prod] = {"expected": exp, "sorted": False}
            rec["ref_log"] = [l for l in r.stdout.decode().splitlines() if "bounds of" in l]
            meta["runs"][rn] = rec
        finally:
            shutil.rmtree(tmp)
    with open(os.path.join(d, "case.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", name, {rn: {k: v.get("lines") for k, v in rec["products"].items() if "lines" in v} for rn, rec in meta["runs"].items()})


def case_s00_trio():
    rng = random.Random(2100)
    base = rand_seq(rng, 2500)
    mat, pat = mutate(rng, base, 0.012), mutate(rng, base, 0.012)
    m = sample_reads(rng, mat, 30, 100, 0.006)
    p = sample_reads(rng, pat, 28, 100, 0.006)
    h = len(p) // 3
    files = {"m.fq": (fq([("@m%d some text" % i, q) for i, q in enumerate(m)]), False),
             "p1.fq": (fq([("@p%d/1" % i, q) for i, q in enumerate(p[:h])]), False),
             "p2.fq": (fq([("@p%d/2" % i, q) for i, q in enumerate(p[h:])]), False)}
    common = ["--paternal", "p1.fq", "--paternal", "p2.fq", "--maternal", "m.fq", "--thread", "2", "--memory", "1"]
    s00_finish("s00_trio_k21", files, {
        "auto": common + ["--auto_bounds"],
        "default_bounds": common + ["--mer", "21"],
        "k17_bounds": common + ["--mer", "17", "--p-lower", "2", "--p-upper", "40", "--m-lower", "1", "--m-upper", "25"]})


def case_s00_gz():
    rng = random.Random(2500)
    base = rand_seq(rng, 1800)
    mat, pat = mutate(rng, base, 0.02), mutate(rng, base, 0.02)
    m = sample_reads(rng, mat, 20, 80, 0.004)
    p = sample_reads(rng, pat, 20, 80, 0.004)
    hm, hp = len(m) // 2, len(p) // 2
    files = {"m_a.fq.gz": (fq([("@a%d" % i, q) for i, q in enumerate(m[:hm])]), True),
             "m_b.fq.gz": (fq([("@b%d" % i, q) for i, q in enumerate(m[hm:])]), True),
             "p_a.fq.gz": (fq([("@a%d" % i, q) for i, q in enumerate(p[:hp])]), True),
             "p_b.fq.gz": (fq([("@b%d" % i, q) for i, q in enumerate(p[hp:])]), True)}
    argv = ["--maternal", "m_a.fq.gz", "--maternal", "m_b.fq.gz", "--paternal", "p_a.fq.gz", "--paternal", "p_b.fq.gz",
            "--mer", "25", "--thread", "3", "--memory", "1", "--p-lower", "3", "--p-upper", "60", "--m-lower", "3", "--m-upper", "60"]
    s00_finish("s00_gz_k25", files, {"gz": argv})


def case_s00_fasta():
    """multi-line FASTA, blank lines, lower case, N runs, CRLF; K = 11 (the script's minimum) on a tiny genome so that
    many k-mers are shared between the parents and repeated inside one"""
    rng = random.Random(1100)
    base = rand_seq(rng, 900)
    def fasta(genome, seed, crlf):
        r = random.Random(seed)
        recs = []
        for i in range(25):
            s = r.randrange(len(genome) - 300)
            q = genome[s:s + r.randint(5, 300)]
            if r.random() < 0.3:
                q = rc(q)
            if r.random() < 0.3:
                j = r.randrange(len(q))
                q = q[:j] + "N" * r.randint(1, 4) + q[j:]
            if r.random() < 0.2:
                q = q.lower()
            if r.random() < 0.2:
                j = r.randrange(len(q))
                q = q[:j] + r.choice("RYKMSW") + q[j + 1:]
            w = r.choice([7, 60, 70, 1000])
            lines = [q[k:k + w] for k in range(0, len(q), w)]
            if r.random() < 0.2:
                lines.insert(r.randrange(len(lines) + 1), "")
            recs.append(">rec%d len=%d\n" % (i, len(q)) + "\n".join(lines) + "\n")
        text = "".join(recs)
        return text.replace("\n", "\r\n") if crlf else text
    files = {"m.fa": (fasta(mutate(rng, base, 0.01), 1, False), False),
             "p.fa": (fasta(mutate(rng, base, 0.01), 2, False)[:-1], False),          # last line unterminated
             "p_crlf.fa": (fasta(mutate(rng, base, 0.01), 3, True), False)}
    s00_finish("s00_fasta_k11", files, {
        "all": ["--maternal", "m.fa", "--paternal", "p.fa", "--paternal", "p_crlf.fa", "--mer", "11", "--memory", "1",
                "--p-lower", "1", "--p-upper", "100000000", "--m-lower", "1", "--m-upper", "100000000"],
        "ge2": ["--maternal", "m.fa", "--paternal", "p.fa", "--mer", "11", "--memory", "1",
                "--p-lower", "2", "--p-upper", "3", "--m-lower", "2", "--m-upper", "1000"]})


def case_s00_edge():
    """multi-line FASTQ, quality lines that start with '@' or '+', reads shorter than K, one k-mer counted more than
    10000 times (jellyfish histo lumps counts > 10000 into row 10001), K = 31 and 32"""
    rng = random.Random(3100)
    g = rand_seq(rng, 700)
    def recs(seed, extra):
        r = random.Random(seed)
        out = []
        for i in range(60):
            s = r.randrange(len(g) - 150)
            q = g[s:s + r.randint(10, 150)]
            if r.random() < 0.1:
                q = q[:5] + "n" + q[6:]
            qual = "".join(r.choice("@+>FI#") for _ in q)
            if r.random() < 0.4 and len(q) > 40:                              # sequence and quality over several lines
                c = r.randint(1, len(q) - 1)
                out.append("@e%d\n%s\n%s\n+e%d\n%s\n%s\n" % (i, q[:c], q[c:], i, qual[:c + 3], qual[c + 3:]))
            else:
                out.append("@e%d\n%s\n+\n%s\n" % (i, q, qual))
        return "".join(out) + extra
    polya = "@polyA\n%s\n+\n%s\n" % ("A" * 10150, "I" * 10150)
    files = {"m.fq": (recs(1, polya), False), "p.fq": (recs(2, "@tail\nACGTTGCATGCATGCATTTAGCAGCATCAGCATCAGCAGGGAT\n+\n" + "I" * 43 + "\n\n"), False)}
    base = ["--maternal", "m.fq", "--paternal", "p.fq", "--memory", "1", "--thread", "1"]
    s00_finish("s00_edge_k31", files, {
        "k31_auto": base + ["--mer", "31", "--auto_bounds"],
        "k32": base + ["--mer", "32", "--p-lower", "1", "--p-upper", "99999", "--m-lower", "1", "--m-upper", "100000000"]})


def main():
    if len(sys.argv) > 1:                     # e.g. `gen_golden.py case_s00_trio case_s00_gz`: only these
        for fn in sys.argv[1:]:
            globals()[fn]()
        return
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    case_edge()
    random_case("rand_k21", 21, 3000, 1500, 60, seed=21)
    random_case("rand_k31", 31, 1500, 600, 40, seed=31, extra_runs=False)
    random_case("rand_k11", 11, 2000, 600, 40, seed=11, read_len=80, extra_runs=False)
    random_case("rand_k32", 32, 500, 200, 20, seed=32, read_len=120, extra_runs=False)
    case_s03("s03_k21", 21, 300, seed=521, max_len=20000)
    case_s03("s03_k31", 31, 300, seed=531, max_len=9000)
    case_s03_edge()
    case_quartering()
    case_s00_trio()
    case_s00_gz()
    case_s00_fasta()
    case_s00_edge()


if __name__ == "__main__":
    main()

"""Shared pytest fixtures.  `-m "not gpu"` runs on a CPU-only box; `-m gpu` needs an MI355X."""
import ctypes
import gzip
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "heavy: BASELINE-size case (full-size oracle sets, GB of input): run after everything else")


def pytest_collection_modifyitems(config, items):
    """cheap tests first, the BASELINE-size ones last: a run that is cut short (-x, a time limit on a slow box) then loses only
    the heavy tail"""
    items.sort(key=lambda it: 1 if it.get_closest_marker("heavy") else 0)       # (stable: the order within each group stays)


def _make(target_dir, *targets):
    subprocess.run(["make", "-s", "-C", target_dir, *targets], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


@pytest.fixture(scope="session")
def oracle_dir():
    """oracle/ with liboracle.so + oracle_classify built (test infrastructure)."""
    d = os.path.join(ROOT, "oracle")
    _make(d, "liboracle.so", "oracle_classify", "oracle_classify_s03")
    return d


@pytest.fixture(scope="session")
def oracle_lib(oracle_dir):
    from tests import oracle_binding
    return oracle_binding.load(os.path.join(oracle_dir, "liboracle.so"))


def golden_cases(program=None):
    """(case, run) pairs; program: None = all, "s01" = stage-01 classify runs, "s03" = per-read classifier runs"""
    out = []
    for name in sorted(os.listdir(GOLDEN)):
        cj = os.path.join(GOLDEN, name, "case.json")
        if os.path.exists(cj):
            meta = json.load(open(cj))
            for run in sorted(meta["runs"]):
                if program is None or meta["runs"][run].get("program", "s01") == program:
                    out.append((name, run))
    return out


@pytest.fixture(scope="session")
def golden_workdir(tmp_path_factory):
    """Golden inputs copied to a temp dir with the k-mer files gunzipped (the reference reads
    them with std::ifstream, classify.cpp:31, so they must be plain text)."""
    base = tmp_path_factory.mktemp("golden")
    for name in sorted(os.listdir(GOLDEN)):
        src = os.path.join(GOLDEN, name)
        if not os.path.isdir(src):
            continue
        dst = base / name
        shutil.copytree(src, dst)
        for fn in ("hap0.mer", "hap1.mer", "reads.fa"):
            gzp = dst / (fn + ".gz")
            if gzp.exists():
                with gzip.open(gzp, "rb") as f, open(dst / fn, "wb") as g:
                    shutil.copyfileobj(f, g)
    return base


def load_case(name):
    return json.load(open(os.path.join(GOLDEN, name, "case.json")))


def run_s00_case(exe, golden_workdir, tmp_path, case, run, extra_args=()):
    """Run a build_unshared_kmers.sh replacement (the oracle's or the product's) on a stage-00 golden case in a scratch
    directory and compare every product the reference script left behind (.mer files as sorted line sets)."""
    import subprocess
    meta = load_case(case)["runs"][run]
    work = tmp_path / ("%s_%s" % (case, run))
    os.makedirs(tmp_path, exist_ok=True)
    shutil.copytree(golden_workdir / case, work)
    res = subprocess.run([exe] + meta["argv"] + list(extra_args), cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, (res.stdout.decode()[-1000:], res.stderr.decode()[-2000:])
    for prod, rec in meta["products"].items():
        got = open(work / prod, "rb").read()
        if rec["sorted"]:
            got = b"".join(sorted(got.splitlines(keepends=True)))
        assert got == open(work / rec["expected"], "rb").read(), (case, run, prod)
    return res

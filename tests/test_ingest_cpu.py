"""Host-only test of the CLI's ingest building blocks (hast_amd/csrc/ingest.h): compiled with ThreadSanitizer."""
import os
import subprocess

from tests.conftest import ROOT


def test_barcode_dict_and_pool_under_tsan(tmp_path):
    exe = tmp_path / "test_ingest"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-o", str(exe),
                    os.path.join(ROOT, "tests", "native", "test_ingest.cpp"), "-lz"], check=True)
    r = subprocess.run([str(exe), "8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"ThreadSanitizer" not in r.stderr
    assert r.stdout.startswith(b"ok ")

"""Host-only test of the fingerprint filter's placement arithmetic (hast_amd/csrc/hast_common.h): the sliding selection
the classify kernel performs over a read must name the same block as the per-key function the build kernel uses."""
import os
import subprocess

from tests.conftest import ROOT


def test_filter_placement_host_model(tmp_path):
    exe = tmp_path / "test_filter_geom"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-o", str(exe),
                    os.path.join(ROOT, "tests", "native", "test_filter_geom.cpp")], check=True)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, (r.stdout.decode()[-500:], r.stderr.decode()[-2000:])
    assert r.stdout.startswith(b"ok ")

"""Host-only test of stage 00's record arithmetic (hast_amd/csrc/kc_common.h "records" / "PLACEMENT", the partitioned counting of
kc_kernels.hip): records cut as the emit kernel cuts them, placed by one m-mer hash, expanded as the LDS pass expands them -- against
the per-window arithmetic of the direct path, for every K and several minimizer lengths, on streams with every byte class."""
import os
import subprocess

from tests.conftest import ROOT


def test_records_of_minimizer_runs_host_model(tmp_path):
    exe = tmp_path / "test_kc_records"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-o", str(exe),
                    os.path.join(ROOT, "tests", "native", "test_kc_records.cpp")], check=True)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, (r.stdout.decode()[-500:], r.stderr.decode()[-2000:])
    assert r.stdout.startswith(b"ok ")

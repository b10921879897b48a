# the whole -m gpu suite and the smoke entry on the final tree
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4aj_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r4aj_pytest.log)"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2

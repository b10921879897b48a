# tools/probe/malloc_probe.py in a row of fresh processes: how long hipMalloc takes right behind a process that freed the same amount
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "4 6.4" "4 6.4" "4 6.4" "6 3.2" "6 3.2" "6 3.2" "6 6.4" "6 6.4" "2 19" "2 19" "4 6.4 touch" "4 6.4 touch" "4 6.4 touch" "6 3.2 touch" "6 3.2 touch" "6 3.2 touch" "1 100 touch" "1 100 touch" "1 100"; do python3 tools/probe/malloc_probe.py $cfg; done

# what streams and held memory cost a short process at its start and at its end (tools/probes/stream_cost.cpp)
# usage: gpurun -- 'bash tools/gpu/stream_cost.sh > gpurun_out/stream_cost.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
hipcc --offload-arch=gfx950 -O2 -o /tmp/stream_cost tools/probes/stream_cost.cpp || exit 1
now() { date +%s.%N; }
one() { local t0=$(now); /tmp/stream_cost "$@"; local t1=$(now); echo "      whole process $(python3 -c "print(round($t1-$t0,3))") s"; }
for rep in 1 2; do
  for n in 0 1 2 4 8 12 16; do one $n 0 0; done
  for n in 4 12; do one $n 1 0; done
  for mb in 1024 8192 65536; do one 4 0 $mb; done
done

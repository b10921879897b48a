# A/B of library builds on ONE box (box-to-box differences are larger than most kernel changes):
#     bash tools/gpu/ab.sh "<bench flags>" libA.so libB.so ...      (paths relative to the repo root; "-" = the tree's libhast.so)
# every library is benched twice, interleaved; prints Gbp/s, kernel ms and the hit totals (which must agree)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
FLAGS=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset HAST_LIB; else export HAST_LIB=$PWD/$lib; fi
    timeout 900 python bench.py --cpu-seconds 0 --steps 10 $FLAGS > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err
    python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$lib [$FLAGS]', round(d['value']/1e9,1), 'Gbp/s kernel', round(d['roofline']['kernel_ms_avg'],3), 'ms (min', round(d['roofline']['kernel_ms_min'],3), ') hits', d['hits']['c0'], d['hits']['c1'])"
  done
done

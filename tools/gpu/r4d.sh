# round 4, device inflate: the intermittent failure of a cold-cache run (stderr kept), kernel times of the device route
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
D=$(mktemp -d /tmp/hast_e2e.XXXXXX)
tools/gen_fastq $D 10000000 5000000 100000 21 150 64 0 || exit 1
ARGS="--hap0 $D/hap0.mer --hap1 $D/hap1.mer --weight0 1.04"
now() { date +%s.%N; }
run() { local name=$1; shift; local t0=$(now); "$@" > $D/out.$name 2> $D/err.$name; local rc=$?; local t1=$(now)
  echo "$name rc=$rc $(python3 -c "print(round($t1-$t0,3))") s md5=$(md5sum < $D/out.$name | cut -c1-12) $(grep -h __stats_read_phase__ $D/err.$name | cut -d" " -f2- | cut -c1-150)"; grep -h "__stats_phases__\|__stats_gz__" $D/err.$name | cut -c1-420; if [ $rc != 0 ]; then tail -5 $D/err.$name; fi; }
(gzip -6 -c $D/r1.fq > $D/r1.fq.gz & gzip -6 -c $D/r2.fq > $D/r2.fq.gz & wait)
evict() { python3 -c "
import os,sys
for p in sys.argv[1:]:
    fd=os.open(p,os.O_RDONLY); os.posix_fadvise(fd,0,0,os.POSIX_FADV_DONTNEED); os.close(fd)" "$@"; }
for rep in 1 2 3 4; do
  evict $D/r1.fq.gz $D/r2.fq.gz $D/hap0.mer $D/hap1.mer
  run gz6_device_cold$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats
done
for rep in 1 2 3; do run gz6_device_warm$rep hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats; done
cd /tmp && HAST_TEARDOWN=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r4d_prof -o gz -- $GRAFT_REPO_ROOT/hast_amd/classify $ARGS --read $D/r1.fq.gz --read $D/r2.fq.gz -t 32 --stats > /dev/null 2> $GRAFT_REPO_ROOT/$O/r4d_prof.err; cd $GRAFT_REPO_ROOT
ls $O/r4d_prof | head
rm -rf $D

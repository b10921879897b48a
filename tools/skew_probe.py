import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import hast_amd
from hast_amd import KmerCounter
rng = np.random.default_rng(1)
n, L = 400_000, 100
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
reads = acgt[rng.integers(0, 4, size=(n, L + 1))]
reads[:, L] = 10
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
sk = rng.random(n) < frac                      # skewed reads: random flank + poly-A core + random flank
mode = sys.argv[2] if len(sys.argv) > 2 else "core40"
if mode == "core40":
    reads[sk, 30:70] = ord('A')            # long homopolymer: one hot k-mer + ~10 distinct keys per read on the A^16 minimizer
else:
    reads[sk, 42:58] = ord('A')            # exactly A^16: 6 windows per read with the A^16 minimizer, all distinct keys
data = reads.reshape(-1)
for rep in range(2):
    with KmerCounter(21, table_bytes=4 << 30) as kc:
        t0 = time.time(); kc.count(0, data); kc.sync(); dt = time.time() - t0
        st = kc.stats()
print(mode, "skew frac %.3f: %.3f s, %.1f Mbp/s, distinct %d" % (frac, dt, n * L / dt / 1e6, st["distinct"][0]))

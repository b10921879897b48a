import numpy as np
rng = np.random.default_rng(5)
N = 400000
seq = rng.integers(0, 4, N, dtype=np.uint64)
def kmers(k):
    # forward values v[i] = bases i..i+k-1 (first base most significant), and revcomp (comp = code^2 in A0C1T2G3? use generic: comp = 3-code for ACGT order here)
    v = np.zeros(N-k+1, dtype=np.uint64); r = np.zeros(N-k+1, dtype=np.uint64)
    for j in range(k):
        v = (v << np.uint64(2)) | seq[j:N-k+1+j]
        r = r | ((np.uint64(3) - seq[j:N-k+1+j]) << np.uint64(2*j))
    return v, r
def h32(x): return ((x * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)).astype(np.uint64)
def canon_hash(k):
    v, r = kmers(k)
    return h32(np.minimum(v, r))
K, W = 21, 6
m = K - W + 1
mh = canon_hash(m)                      # per position canonical m-mer hash
nwin = N - K + 1
# plain random minimizer: min over W m-mers
plain = np.min(np.stack([mh[j:j+nwin] for j in range(W)]), axis=0)
dens_plain = (plain[1:] != plain[:-1]).mean()
print("plain minimizer (w=%d, m=%d): density %.4f -> lines per 130 windows %.1f" % (W, m, dens_plain, 1 + 129*dens_plain))
for t in (4, 10):
    ell = W + m - t
    if ell % W: print("t=%d: ell=%d not multiple of w" % (t, ell)); continue
    th = canon_hash(t)
    T = np.stack([th[j:j+nwin] for j in range(ell)])           # [ell][nwin]
    tmin = T.min(axis=0)
    M = np.stack([mh[j:j+nwin] for j in range(W)])              # [W][nwin]
    best = np.full(nwin, np.uint64(0xFFFFFFFFFFFFFFFF))
    for x in range(ell):
        cand = T[x] == tmin
        sel = M[x % W]
        best = np.where(cand, np.minimum(best, sel), best)
    d = (best[1:] != best[:-1]).mean()
    print("symmetric mod-minimizer t=%d ell=%d: density %.4f -> lines per 130 windows %.1f" % (t, ell, d, 1 + 129*d))
    # check symmetry: compute on reverse-complement sequence and compare mirrored

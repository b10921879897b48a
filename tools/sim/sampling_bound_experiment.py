"""Round-4 bounded experiment: can the probe kernel ask for fewer 128-B blocks per read?
The kernel sits at the random-request ceiling, so requests per read are the only lever: one request per run of consecutive
windows that sample the same m-mer.  Simulated here: runs per 150-bp read (130 windows at K = 21) of forward sampling schemes
(the filter files both strands, so no canonical forms), for the geometries a 128-B block could host, against the lower bound of
ANY forward scheme, ceil((W+m)/W)/(W+m) sampled positions per window.  Printed as a table; DESIGN.md section 8 quotes it."""
import numpy as np

rng = np.random.default_rng(7)
K, L, N_READS = 21, 150, 20000
NW = L - K + 1
reads = rng.integers(0, 4, (N_READS, L), dtype=np.uint64)


def mers(k):
    v = np.zeros((N_READS, L - k + 1), dtype=np.uint64)
    for j in range(k):
        v = (v << np.uint64(2)) | reads[:, j:L - k + 1 + j]
    return v


def h(x, salt=0):
    x = (x + np.uint64(salt)) * np.uint64(0x9E3779B97F4A7C15)
    x ^= x >> np.uint64(29)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    return x >> np.uint64(20)


def runs_per_read(sampled_pos):
    """sampled_pos[r][w] = absolute position of the m-mer window w of read r samples"""
    return 1 + (sampled_pos[:, 1:] != sampled_pos[:, :-1]).sum(axis=1).mean()


def modmin(m, t, order=None):
    """mod-minimizer: smallest t-mer (leftmost on ties) among the window's first kp - t + 1 positions, at x; sample the m-mer at x mod W"""
    W = K - m + 1
    nt = K - t + 1
    th = h(mers(t)) if order is None else order(t)
    th = (th << np.uint64(12)) | np.arange(L - t + 1, dtype=np.uint64)[None, :]          # leftmost on ties
    win = np.stack([th[:, j:j + NW] for j in range(nt)])                               # [nt][reads][windows]
    x = win.min(axis=0) & np.uint64(0xFFF)                                              # absolute position of the smallest t-mer
    rel = x - np.arange(NW, dtype=np.uint64)[None, :]
    return runs_per_read(np.arange(NW, dtype=np.uint64)[None, :] + rel % np.uint64(W))


def plain_min(m):
    W = K - m + 1
    mh = (h(mers(m)) << np.uint64(12)) | np.arange(L - m + 1, dtype=np.uint64)[None, :]
    win = np.stack([mh[:, j:j + NW] for j in range(W)])
    return runs_per_read(win.min(axis=0) & np.uint64(0xFFF))


def oc_order(s):
    """open-closed order on t-mers (Groot Koerkamp, Liu, Pibiri 2025): open syncmers first (smallest s-mer in the middle), then
    closed ones (at either end), then the rest; hash inside a class"""
    def f(t):
        tm = mers(t)
        ns = t - s + 1
        sm = np.stack([h((tm >> np.uint64(2 * (t - s - j))) & np.uint64(4 ** s - 1), 99) for j in range(ns)])
        at = sm.argmin(axis=0)
        cls = np.where(at == (ns - 1) // 2, 0, np.where((at == 0) | (at == ns - 1), 1, 2)).astype(np.uint64)
        return (cls << np.uint64(40)) | (h(tm) & np.uint64((1 << 40) - 1))
    return f


def bound(m):
    W = K - m + 1
    return NW * (-(-(W + m) // W)) / (W + m)


rows = []
for m in (14, 13, 12):
    W = K - m + 1
    rows.append(("random minimizer", m, W, "-", plain_min(m)))
    for t in range(2, m + 1):
        if (t - m) % W == 0:
            rows.append(("mod-minimizer", m, W, t, modmin(m, t)))
            if 4 <= t <= 8:
                rows.append(("mod-minimizer, open-closed t-mer order (s=%d)" % max(2, t - 3), m, W, t, modmin(m, t, oc_order(max(2, t - 3)))))
    for t in (5, 6, 7):
        if (t - m) % W:
            rows.append(("mod-minimizer (t not congruent to m mod W)", m, W, t, modmin(m, t)))
print("%-58s %3s %3s %3s  %8s  %8s  %s" % ("scheme", "m", "W", "t", "runs/read", "bound", "strings per 128-B block at 400M keys (two strands)"))
for name, m, W, t, r in rows:
    print("%-58s %3d %3d %3s  %8.2f  %8.2f  %.1f of %s" % (name, m, W, t, r, bound(m), 8e8 / 4 ** m,
                                                          "64 16-bit entries (exact codes fit: 2(K-m)+log2 W <= 17)" if 2 * (K - m) + np.log2(W) <= 17 else
                                                          "32 entries (an exact code needs %d bits > 16)" % int(np.ceil(2 * (K - m) + np.log2(W)) + 2)))

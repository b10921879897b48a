import numpy as np
rng = np.random.default_rng(3)
def revcomp(x, k):
    x = x ^ np.uint64(0xAAAAAAAAAAAAAAAA)
    x = ((x & np.uint64(0x3333333333333333)) << np.uint64(2)) | ((x >> np.uint64(2)) & np.uint64(0x3333333333333333))
    x = ((x & np.uint64(0x0F0F0F0F0F0F0F0F)) << np.uint64(4)) | ((x >> np.uint64(4)) & np.uint64(0x0F0F0F0F0F0F0F0F))
    x = x.byteswap()
    return x >> np.uint64(64-2*k)
def mmer_hash(cm): return ((cm * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)).astype(np.uint32)
def minh(keys,k,m):
    best=np.full(keys.shape,0xFFFFFFFF,dtype=np.uint32); mask=np.uint64((1<<(2*m))-1)
    for j in range(k-m+1):
        mm=(keys>>np.uint64(2*(k-m-j)))&mask
        best=np.minimum(best, mmer_hash(np.minimum(mm,revcomp(mm,m))))
    return best
def bucket(mh,nb): return (((mh*np.uint32(0x9E3779B1)).astype(np.uint64)*np.uint64(nb))>>np.uint64(32)).astype(np.int64)
def sim(label, K, m, N):
    raw=rng.integers(0,1<<(2*K),N,dtype=np.uint64); keys=np.unique(np.minimum(raw,revcomp(raw,K)))
    q=rng.integers(0,1<<(2*K),1_000_000,dtype=np.uint64); q=np.minimum(q,revcomp(q,K))
    kh=minh(keys,K,m); qh=minh(q,K,m)
    for slots in (8,16):
        for lf in (0.5,0.35,0.25,0.18):
            nb=int(len(keys)/lf/slots)+1
            cnt=np.bincount(bucket(kh,nb),minlength=nb)
            fill=np.empty(nb,dtype=np.int64)
            for _ in range(2):
                carry=0
                for i in range(nb):
                    t=cnt[i]+carry; fill[i]=min(t,slots); carry=t-fill[i]
            full=fill>=slots
            print(f"{label} slots={slots} lf={lf} P(home full)={full[bucket(qh,nb)].mean():.4f}")
sim("real(m=16,w=6)", 17, 12, int(0.19*4**12/2))
sim("real(m=15,w=7)", 18, 12, int(0.75*4**12/2))
sim("real(m=14,w=8)", 18, 11, int(3*4**11/2))

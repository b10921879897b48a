"""How long does hipMalloc take in a process that starts right after another one freed the same amount?  (round 4: the device inflate's
symbol arenas and stage 00's table + record buffers showed seconds of 'table' / 'open' time in the SECOND of two runs back to back.)
usage: malloc_probe.py <n_buffers> <GB each> [touch]"""
import ctypes, sys, time
hip = ctypes.CDLL("libamdhip64.so")
n, gb = int(sys.argv[1]), float(sys.argv[2])
touch = len(sys.argv) > 3
t0 = time.perf_counter()
hip.hipSetDevice(0)
p0 = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p0), ctypes.c_size_t(1 << 20))
t1 = time.perf_counter()
ts = []
ptrs = []
for i in range(n):
    p = ctypes.c_void_p()
    a = time.perf_counter()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(int(gb * (1 << 30))))
    ts.append(time.perf_counter() - a)
    ptrs.append(p)
    assert rc == 0, rc
t2 = time.perf_counter()
if touch:
    for p in ptrs:
        hip.hipMemset(p, 0, ctypes.c_size_t(int(gb * (1 << 30))))
    hip.hipDeviceSynchronize()
t3 = time.perf_counter()
print("context %.3f s; %d x %.1f GB: %s = %.3f s; memset %.3f s" % (t1 - t0, n, gb, " ".join("%.3f" % x for x in ts), t2 - t1, t3 - t2), flush=True)
import os
os._exit(0)

import ctypes as C, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from strainscan_amd import _lib
L = _lib.lib(); _lib.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
lvl = sys.argv[2] if len(sys.argv) > 2 else "-6"
rs = np.random.RandomState(1)
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
p = os.path.join(d, "gi_big_%d.fq" % os.getpid())
a = np.empty((n, 307), np.uint8)
a[:, 0:2] = np.frombuffer(b"@r", np.uint8); a[:, 2] = 10
a[:, 3:153] = np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, size=(n, 150))]
a[:, 153] = 10; a[:, 154] = ord("+"); a[:, 155] = 10
q = np.clip(38 - np.abs(rs.normal(0, 4, size=(n, 150))).astype(np.int64) - (np.arange(150) // 30), 2, 40) + 33
a[:, 156:306] = q.astype(np.uint8); a[:, 306] = 10
a.tofile(p)
t0 = time.time(); subprocess.check_call(["gzip", "-k", "-f", lvl, p]); print("gzip", lvl, round(time.time() - t0, 1), "s", os.path.getsize(p + ".gz") / 1e6, "MB ->", os.path.getsize(p) / 1e6, "MB", flush=True)
want = a.tobytes()
for rep in range(3):
    t = C.c_void_p(); m = C.c_uint64()
    t0 = time.time(); rc = L.ss_gz_inflate_gpu(os.fsencode(p + ".gz"), C.byref(t), C.byref(m)); dt = time.time() - t0
    ok = None
    if rc == 0:
        ok = C.string_at(t, m.value) == want; L.ss_gz_free(t)
    print("gpu inflate rc", rc, "equal", ok, "%.3f s" % dt, "%.1f M reads/s" % (n / dt / 1e6), flush=True)
t = C.c_void_p(); m = C.c_uint64()
t0 = time.time(); rc = L.ss_gz_inflate(os.fsencode(p + ".gz"), 2, 0, C.byref(t), C.byref(m)); dt = time.time() - t0
print("host threaded inflate rc", rc, "%.3f s" % dt)
os.remove(p); os.remove(p + ".gz")

import ctypes as C, gzip, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
rs = np.random.RandomState(1)
a = np.empty((n, 307), np.uint8)
a[:, 0:2] = np.frombuffer(b"@r", np.uint8); a[:, 2] = 10
a[:, 3:153] = np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, size=(n, 150))]
a[:, 153] = 10; a[:, 154] = ord("+"); a[:, 155] = 10
q = np.clip(38 - np.abs(rs.normal(0, 4, size=(n, 150))).astype(np.int64) - (np.arange(150) // 30), 2, 40) + 33
a[:, 156:306] = q.astype(np.uint8); a[:, 306] = 10
want = a.tobytes()
p = "/tmp/gi_prof_%d.gz" % os.getpid()
open(p, "wb").write(gzip.compress(want, 6))
from strainscan_amd import _lib
L = _lib.lib(); _lib.require_gpu()
for rep in range(2):
    t = C.c_void_p(); m = C.c_uint64()
    t0 = time.time(); rc = L.ss_gz_inflate_gpu(os.fsencode(p), C.byref(t), C.byref(m)); dt = time.time() - t0
    ok = None
    if rc == 0:
        ok = C.string_at(t, m.value) == want; L.ss_gz_free(t)
    print("gpu inflate rc", rc, "equal", ok, "%.3f s" % dt, flush=True)
os.remove(p)

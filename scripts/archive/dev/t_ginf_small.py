"""Device inflater on one small FASTQ .gz: rc, equality, first mismatch"""
import ctypes as C, gzip, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SS_INGEST_TRACE"] = "1"
os.environ["SS_GZ_SKIPCRC"] = "1"
from strainscan_amd import _lib
L = _lib.lib(); _lib.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
lvl = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rs = np.random.RandomState(1)
a = np.empty((n, 307), np.uint8)
a[:, 0:2] = np.frombuffer(b"@r", np.uint8); a[:, 2] = 10
a[:, 3:153] = np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, size=(n, 150))]
a[:, 153] = 10; a[:, 154] = ord("+"); a[:, 155] = 10
q = np.clip(38 - np.abs(rs.normal(0, 4, size=(n, 150))).astype(np.int64) - (np.arange(150) // 30), 2, 40) + 33
a[:, 156:306] = q.astype(np.uint8); a[:, 306] = 10
want = a.tobytes()
p = "/tmp/gi_small_%d.gz" % os.getpid()
open(p, "wb").write(gzip.compress(want, lvl))
t = C.c_void_p(); m = C.c_uint64()
rc = L.ss_gz_inflate_gpu(os.fsencode(p), C.byref(t), C.byref(m))
if rc == 0:
    got = C.string_at(t, m.value); L.ss_gz_free(t)
    g = np.frombuffer(got, np.uint8); w = np.frombuffer(want, np.uint8)
    if len(g) != len(w): print("LIB", os.environ.get("SS_LIB"), "length", len(g), len(w))
    else:
        bad = np.flatnonzero(g != w)
        print("LIB", os.environ.get("SS_LIB"), "rc 0 mismatches", bad.size, "first", bad[:8].tolist())
        if bad.size:
            i = int(bad[0]); print("  got ", got[max(0, i - 20):i + 20]); print("  want", want[max(0, i - 20):i + 20])
else:
    print("LIB", os.environ.get("SS_LIB"), "rc", rc)
os.remove(p)

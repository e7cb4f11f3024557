"""Device inflater: phase times (SS_INGEST_TRACE on stderr) across chunk sizes on one FASTQ .gz"""
import ctypes as C, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SS_INGEST_TRACE"] = "1"
from strainscan_amd import _lib
L = _lib.lib(); _lib.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
lvl = sys.argv[2] if len(sys.argv) > 2 else "-6"
chunks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [16384, 24576, 32768, 45000, 65536, 98304]
rs = np.random.RandomState(1)
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
p = os.path.join(d, "gi_sw_%d.fq" % os.getpid())
a = np.empty((n, 307), np.uint8)
a[:, 0:2] = np.frombuffer(b"@r", np.uint8); a[:, 2] = 10
a[:, 3:153] = np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, size=(n, 150))]
a[:, 153] = 10; a[:, 154] = ord("+"); a[:, 155] = 10
q = np.clip(38 - np.abs(rs.normal(0, 4, size=(n, 150))).astype(np.int64) - (np.arange(150) // 30), 2, 40) + 33
a[:, 156:306] = q.astype(np.uint8); a[:, 306] = 10
a.tofile(p)
subprocess.check_call(["gzip", "-k", "-f", lvl, p])
print("gz", os.path.getsize(p + ".gz") / 1e6, "MB ->", os.path.getsize(p) / 1e6, "MB", flush=True)
want = a.tobytes()
for ch in chunks:
    os.environ["SS_GZ_CHUNK"] = str(ch)
    for rep in range(2):
        t = C.c_void_p(); m = C.c_uint64()
        sys.stderr.write("== chunk %d rep %d\n" % (ch, rep)); sys.stderr.flush()
        t0 = time.time(); rc = L.ss_gz_inflate_gpu(os.fsencode(p + ".gz"), C.byref(t), C.byref(m)); dt = time.time() - t0
        ok = None
        if rc == 0:
            ok = C.string_at(t, m.value) == want; L.ss_gz_free(t)
        print("chunk", ch, "rc", rc, "equal", ok, "%.3f s" % dt, flush=True)
os.remove(p); os.remove(p + ".gz")

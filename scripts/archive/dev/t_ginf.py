import ctypes as C, gzip, os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from strainscan_amd import _lib
L = _lib.lib()
_lib.require_gpu()
def fastq(n, seed):
    rs = np.random.RandomState(seed)
    lut = np.frombuffer(b"ACGT", np.uint8)
    q = np.frombuffer(b"FFFFFFFF:FFF,FF#", np.uint8)
    out = []
    for i in range(n):
        out.append(b"@SRR1234567.%d %d/1\n" % (i, i) + lut[rs.randint(0, 4, 150)].tobytes() + b"\n+\n" + q[rs.randint(0, 16, 150)].tobytes() + b"\n")
    return b"".join(out)
def gpu_inflate(path):
    t = C.c_void_p(); n = C.c_uint64()
    t0 = time.time()
    rc = L.ss_gz_inflate_gpu(os.fsencode(path), C.byref(t), C.byref(n))
    dt = time.time() - t0
    if rc != 0: return rc, None, dt
    s = C.string_at(t, n.value); L.ss_gz_free(t)
    return 0, s, dt
cases = []
txt = fastq(40000, 1)
for lvl in (1, 6, 9):
    cases.append(("fastq_l%d" % lvl, gzip.compress(txt, lvl), txt))
rep = (b"ACGTACGTTTGA" * 1000 + b"\n") * 300
cases.append(("repetitive", gzip.compress(rep, 6), rep))
rnd = np.random.RandomState(3).randint(0, 256, 3_000_000).astype(np.uint8).tobytes()
cases.append(("random", gzip.compress(rnd, 6), rnd))
small = b"hello world\n" * 10
cases.append(("tiny", gzip.compress(small, 6), small))
os.makedirs('/tmp/gi', exist_ok=True)
for name, gz, want in cases:
    p = '/tmp/gi/%s.gz' % name
    open(p, 'wb').write(gz)
    for chunk in ("65536", "16384", None):
        if chunk: os.environ["SS_GZ_CHUNK"] = chunk
        else: os.environ.pop("SS_GZ_CHUNK", None)
        rc, got, dt = gpu_inflate(p)
        print(name, len(gz), len(want), "chunk", chunk, "rc", rc, "ok" if (got == want) else "MISMATCH" if got is not None else "-", "%.3fs" % dt, flush=True)

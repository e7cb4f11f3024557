"""Device inflater vs the number of deflate blocks: the same FASTQ text compressed with zlib memLevel 8 / 7 / 6 / 5 (a block
holds 16 K / 8 K / 4 K / 2 K symbols) -- how the inflate kernel would run if a block could be entered at several places."""
import ctypes as C, os, sys, time, zlib, struct
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SS_INGEST_TRACE"] = "1"
from strainscan_amd import _lib
L = _lib.lib(); _lib.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rs = np.random.RandomState(1)
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
a = np.empty((n, 307), np.uint8)
a[:, 0:2] = np.frombuffer(b"@r", np.uint8); a[:, 2] = 10
a[:, 3:153] = np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, size=(n, 150))]
a[:, 153] = 10; a[:, 154] = ord("+"); a[:, 155] = 10
q = np.clip(38 - np.abs(rs.normal(0, 4, size=(n, 150))).astype(np.int64) - (np.arange(150) // 30), 2, 40) + 33
a[:, 156:306] = q.astype(np.uint8); a[:, 306] = 10
want = a.tobytes()
for mem, chunk in ((8, 32768), (7, 16384), (6, 8192), (5, 4096)):
    co = zlib.compressobj(6, zlib.DEFLATED, -15, mem)
    body = co.compress(want) + co.flush()
    gz = b"\x1f\x8b\x08\0\0\0\0\0\0\xff" + body + struct.pack("<II", zlib.crc32(want) & 0xFFFFFFFF, len(want) & 0xFFFFFFFF)
    p = os.path.join(d, "gi_bl_%d_%d.gz" % (os.getpid(), mem))
    open(p, "wb").write(gz)
    os.environ["SS_GZ_CHUNK"] = str(chunk)
    for rep in range(2):
        t = C.c_void_p(); m = C.c_uint64()
        sys.stderr.write("== memLevel %d (%.1f MB) chunk %d rep %d\n" % (mem, len(gz) / 1e6, chunk, rep)); sys.stderr.flush()
        t0 = time.time(); rc = L.ss_gz_inflate_gpu(os.fsencode(p), C.byref(t), C.byref(m)); dt = time.time() - t0
        ok = None
        if rc == 0:
            ok = C.string_at(t, m.value) == want; L.ss_gz_free(t)
        sys.stderr.write("memLevel %d chunk %d rc %d equal %s %.3f s\n" % (mem, chunk, rc, ok, dt)); sys.stderr.flush()
    os.remove(p)

"""A pair of large .fastq.gz files (many segments, real wrong entries): row counts through the device path equal those of
the plain text; the resident read set holds the same records."""
import hashlib, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SS_INGEST_TRACE"] = "1"
from strainscan_amd import _lib as L
L.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
lvl = sys.argv[2] if len(sys.argv) > 2 else "-1"
d = "/dev/shm"
rs0 = np.random.RandomState(3)
lut = np.frombuffer(b"ACGT", np.uint8)
rows = lut[rs0.randint(0, 4, (200000, 31))]
kfa = b"".join(b">1\n" + r.tobytes() + b"\n" for r in rows)
db = L.KmerDB.from_text(kfa, 31, True)
paths = []
for f in range(2):
    rs = np.random.RandomState(10 + f)
    p = os.path.join(d, "gz_big_%d_%d.fq" % (os.getpid(), f + 1))
    a = np.empty((n, 307), np.uint8)
    a[:, 0:2] = np.frombuffer(b"@r", np.uint8); a[:, 2] = 10
    a[:, 3:153] = lut[rs.randint(0, 4, size=(n, 150))]
    idx = rs.randint(0, n, n // 4); off = rs.randint(0, 119, n // 4); src = rs.randint(0, rows.shape[0], n // 4)
    for j in range(31): a[idx, 3 + off + j] = rows[src, j]
    a[:, 153] = 10; a[:, 154] = ord("+"); a[:, 155] = 10
    q = np.clip(38 - np.abs(rs.normal(0, 4, size=(n, 150))).astype(np.int64) - (np.arange(150) // 30), 2, 40) + 33
    a[:, 156:306] = q.astype(np.uint8); a[:, 306] = 10
    a.tofile(p); del a, q
    paths.append(p)
db.reset(); nrec, _ = db.scan_files(paths); want = db.counts_rows().copy()
print("plain: records", nrec, "hits", int(want.sum()), flush=True)
pr = [subprocess.Popen(["gzip", "-f", lvl, p]) for p in paths]
[q.wait() for q in pr]
gz = [p + ".gz" for p in paths]
print("gz MB", [round(os.path.getsize(p) / 1e6) for p in gz], flush=True)
for mode in ("1", "0"):
    os.environ["SS_GZ_GPU"] = mode
    db.reset(); t0 = time.time(); nrec2, _ = db.scan_files(gz); dt = time.time() - t0
    print("SS_GZ_GPU=%s scan_files: %.3f s, records %d, counts equal %s" % (mode, dt, nrec2, bool(np.array_equal(db.counts_rows(), want))), flush=True)
    t0 = time.time(); rset = L.ReadSet(gz, 0, 1); dt = time.time() - t0
    db.reset(); rset.scan_into(db); L.check(L.lib().ss_device_sync(), "sync")
    print("SS_GZ_GPU=%s read set: %.3f s, records %d, counts equal %s" % (mode, dt, rset.info()["n_records"], bool(np.array_equal(db.counts_rows(), want))), flush=True)
    rset.close()
for p in gz: os.remove(p)

"""Resident read set from a pair of .fastq.gz files: host inflaters vs the device inflater (SS_GZ_GPU=1)"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SS_INGEST_TRACE"] = "1"
from strainscan_amd import _lib as L
L.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
lvl = sys.argv[2] if len(sys.argv) > 2 else "-6"
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
paths = []
for f in range(2):
    rs = np.random.RandomState(10 + f)
    p = os.path.join(d, "gz_e2e_%d_%d.fq" % (os.getpid(), f + 1))
    a = np.empty((n, 307), np.uint8)
    a[:, 0:2] = np.frombuffer(b"@r", np.uint8); a[:, 2] = 10
    a[:, 3:153] = np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, size=(n, 150))]
    a[:, 153] = 10; a[:, 154] = ord("+"); a[:, 155] = 10
    q = np.clip(38 - np.abs(rs.normal(0, 4, size=(n, 150))).astype(np.int64) - (np.arange(150) // 30), 2, 40) + 33
    a[:, 156:306] = q.astype(np.uint8); a[:, 306] = 10
    a.tofile(p)
    subprocess.check_call(["gzip", "-f", lvl, p])
    paths.append(p + ".gz")
print("files", [round(os.path.getsize(p) / 1e6, 1) for p in paths], "MB", flush=True)
for mode in ("0", "1", "0", "1"):
    os.environ["SS_GZ_GPU"] = mode
    sys.stderr.write("==== SS_GZ_GPU=%s\n" % mode); sys.stderr.flush()
    t0 = time.time()
    rs_ = L.ReadSet(paths, 0, 1)
    dt = time.time() - t0
    info = rs_.info()
    print("SS_GZ_GPU=%s: read set of %d records in %.3f s = %.1f M reads/s" % (mode, info["n_records"], dt, info["n_records"] / dt / 1e6), flush=True)
    rs_.close()
for p in paths: os.remove(p)

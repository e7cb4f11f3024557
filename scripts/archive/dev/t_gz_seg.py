"""Resident read set from a pair of .fastq.gz files for several segment sizes of the device inflater"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from strainscan_amd import _lib as L
L.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
d = "/dev/shm"
paths = []
for f in range(2):
    rs = np.random.RandomState(10 + f)
    p = os.path.join(d, "gz_seg_%d_%d.fq" % (os.getpid(), f + 1))
    a = np.empty((n, 307), np.uint8)
    a[:, 0:2] = np.frombuffer(b"@r", np.uint8); a[:, 2] = 10
    a[:, 3:153] = np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, size=(n, 150))]
    a[:, 153] = 10; a[:, 154] = ord("+"); a[:, 155] = 10
    q = np.clip(38 - np.abs(rs.normal(0, 4, size=(n, 150))).astype(np.int64) - (np.arange(150) // 30), 2, 40) + 33
    a[:, 156:306] = q.astype(np.uint8); a[:, 306] = 10
    a.tofile(p); del a, q
    paths.append(p)
pr = [subprocess.Popen(["gzip", "-f", "-1", p]) for p in paths]
[q.wait() for q in pr]
paths = [p + ".gz" for p in paths]
print("gz MB", [round(os.path.getsize(p) / 1e6) for p in paths], flush=True)
for seg_mb in (128, 128):
    os.environ["SS_GZ_SEG_KB"] = str(seg_mb * 1024)
    ts = []
    for rep in range(3):
        t0 = time.time(); rs_ = L.ReadSet(paths, 0, 1); ts.append(time.time() - t0); rs_.close()
    print("segment %4d MB: %s  best %.3f s = %.1f M reads/s" % (seg_mb, " ".join("%.3f" % t for t in ts), min(ts), 2 * n / min(ts) / 1e6), flush=True)
for p in paths: os.remove(p)

"""Resident read set from a pair of .fastq.gz files on the device, over SS_GZ_SPLIT_KB (blocks entered every so many KB of
deflate data, subsync_kernel; 0 = at their starts only): GNU gzip -6 (blocks of ~57 KB) and zlib level 6 (~30 KB)."""
import gzip, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from strainscan_amd import _lib as L
L.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
splits = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 6, 8, 12, 16, 24]
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
sets = {"gnu": [], "zlib": []}
for f in range(2):
    rs = np.random.RandomState(10 + f)
    p = os.path.join(d, "gz_sp_%d_%d.fq" % (os.getpid(), f + 1))
    a = np.empty((n, 307), np.uint8)
    a[:, 0:2] = np.frombuffer(b"@r", np.uint8); a[:, 2] = 10
    a[:, 3:153] = np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, size=(n, 150))]
    a[:, 153] = 10; a[:, 154] = ord("+"); a[:, 155] = 10
    q = np.clip(38 - np.abs(rs.normal(0, 4, size=(n, 150))).astype(np.int64) - (np.arange(150) // 30), 2, 40) + 33
    a[:, 156:306] = q.astype(np.uint8); a[:, 306] = 10
    a.tofile(p)
    pz = p + ".zlib.gz"
    open(pz, "wb").write(gzip.compress(a.tobytes(), 6))
    subprocess.check_call(["gzip", "-f", "-6", p])
    sets["gnu"].append(p + ".gz"); sets["zlib"].append(pz)
only = sys.argv[3] if len(sys.argv) > 3 else ""
for name, paths in sets.items():
    if only and name != only:
        continue
    for sp in splits:
        os.environ["SS_GZ_SPLIT_KB"] = str(sp)
        ts = []
        for rep in range(4):
            t0 = time.time()
            rs_ = L.ReadSet(paths, 0, 1)
            ts.append(time.time() - t0)
            nrec = rs_.info()["n_records"]
            rs_.close()
        print("%-5s split %2d KB: %d records, best %.4f s (%.1f M reads/s), all %s" % (name, sp, nrec, min(ts[1:]), nrec / min(ts[1:]) / 1e6, ["%.3f" % t for t in ts]), flush=True)
for ps in sets.values():
    for p in ps: os.remove(p)

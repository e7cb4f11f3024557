#!/usr/bin/env python3
"""One .fastq.gz inflated whole: threaded inflater vs libdeflate vs zlib (python gzip), same bytes."""
import gzip, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strainscan_amd import _lib
from tests.test_abi_and_host import _fastq_like
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
p = os.path.join(d, "ss_pgz_%d.fq" % os.getpid())
rs = np.random.RandomState(3)
with open(p, "wb") as f:
    for _ in range(0, n, 200000):
        f.write(_fastq_like(rs, 200000))
try:
    for level in (1, 6):
        subprocess.check_call("gzip -%d -c %s > %s.gz" % (level, p, p), shell=True)
        size = os.path.getsize(p + ".gz")
        t0 = time.perf_counter(); want = subprocess.run(["zcat", p + ".gz"], capture_output=True).stdout; tz = time.perf_counter() - t0
        for mode, threads, name in ((2, 0, "libdeflate"), (1, 4, "pgz x4"), (1, 16, "pgz x16"), (1, 32, "pgz x32"), (0, 0, "auto")):
            t0 = time.perf_counter(); got = _lib.gz_inflate(p + ".gz", threads, mode); t1 = time.perf_counter()
            print("level %d %-10s %.3f s %s  (%.0f MB/s of text)" % (level, name, t1 - t0, "same" if got == want else ("DECLINED" if got is None else "DIFFERENT"), len(want) / 1e6 / (t1 - t0)))
        print("level %d zcat %.2f s, %d -> %d bytes" % (level, tz, size, len(want)))
finally:
    for q in (p, p + ".gz"):
        if os.path.exists(q): os.unlink(q)

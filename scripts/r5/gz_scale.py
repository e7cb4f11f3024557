#!/usr/bin/env python3
"""A .fastq.gz pair of N reads -> resident read set, with SS_INGEST_TRACE's stage lines: where a LARGE pair spends its time
(the bench's pair is 2 x 500 K reads).   gz_scale.py [n_reads = 20000000] [gzip level = 1] [loads = 3]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from strainscan_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
lvl = sys.argv[2] if len(sys.argv) > 2 else "1"
loads = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
keep = os.environ.get("SS_GZ_KEEP_DIR")              # the pair stays there (SS_GZ_REUSE=1: and is not made again)
d = keep or os.path.join("/dev/shm", "ss_gzs_%d" % os.getpid())
reuse = bool(keep and os.environ.get("SS_GZ_REUSE") and all(os.path.exists(os.path.join(d, "s_%d.fq.gz" % i)) for i in (1, 2)))
os.makedirs(d, exist_ok=True)
try:
    fq = [os.path.join(d, "s_%d.fq" % i) for i in (1, 2)]
    if not reuse:
        spec = bench.make_db(torch, dev, 23, seed=1)
        r = bench.make_reads(torch, dev, spec, n, seed=2, hit_frac=0.05)
        half = n // 2
        bench.write_fastq(r[: half * 151], half, fq[0], noisy_quality_seed=5)
        bench.write_fastq(r[half * 151:], n - half, fq[1], noisy_quality_seed=6)
        del r
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        gz = [p + ".gz" for p in fq]
        if os.environ.get("SS_GZ_INPROCESS"):           # (under rocprofv3: no child processes)
            import threading, zlib

            def comp(src, dst):
                c = zlib.compressobj(int(lvl), zlib.DEFLATED, 31)
                with open(src, "rb") as f, open(dst, "wb") as o:
                    while True:
                        b = f.read(8 << 20)
                        if not b:
                            break
                        o.write(c.compress(b))
                    o.write(c.flush())
                os.unlink(src)
            th = [threading.Thread(target=comp, args=(a, b)) for a, b in zip(fq, gz)]
            [t.start() for t in th]
            [t.join() for t in th]
        else:
            pr = [subprocess.Popen(["gzip", "-" + lvl, p]) for p in fq]
            assert all(q.wait() == 0 for q in pr)
        print("gzip -%s: %.1f s, %d MB" % (lvl, time.perf_counter() - t0, sum(os.path.getsize(p) for p in gz) >> 20), flush=True)
    gz = [p + ".gz" for p in fq]
    if os.environ.get("SS_GZ_FRESH"):                # a FRESH process per load, as the CLI is one: traces of its first (only) load
        code = ("import sys, time; sys.path.insert(0, %r); t0 = time.perf_counter(); from strainscan_amd import _lib; _lib.warm_up(gz=2); t1 = time.perf_counter();"
                "rs = _lib.ReadSet(%r); _lib.check(_lib.lib().ss_device_sync(), 'sync'); t2 = time.perf_counter();"
                "print('fresh process: import + warm_up %%.3f s, load %%.3f s = %%.1f M reads/s' %% (t1 - t0, t2 - t1, %d / (t2 - t1) / 1e6), flush=True); import os; os._exit(0)" % (ROOT, gz, n))
        for i in range(loads):
            r_ = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
            print(r_.stdout.strip(), flush=True)
            if i == loads - 1:
                print(r_.stderr[-6000:], flush=True)
    _lib.warm_up(gz=2)
    for i in range(loads):
        t0 = time.perf_counter()
        rs = _lib.ReadSet(gz)
        _lib.check(_lib.lib().ss_device_sync(), "sync")
        dt = time.perf_counter() - t0
        info = rs.info()
        rs.close()
        print("load %d: %.1f ms = %.1f M reads/s (%d records)" % (i, dt * 1e3, n / dt / 1e6, info["n_records"]), flush=True)
finally:
    import shutil
    if not keep:
        shutil.rmtree(d, ignore_errors=True)

#!/usr/bin/env python3
"""Where the file -> HBM time goes: ReadSet load (resident blocks) vs streaming scan_files, cold and warm workers."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from scripts.bench_cli import write_fastq


def main():
    import torch
    from strainscan_amd import _lib
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
    dev = torch.device("cuda", 0)
    spec = bench.make_db(torch, dev, 823, seed=20231013)
    db = _lib.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True)
    r = bench.make_reads(torch, dev, spec, n_reads, seed=2, hit_frac=0.05)
    d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    fq = [os.path.join(d, "ss_ing_%d_%d.fq" % (os.getpid(), i)) for i in range(2)]
    half = n_reads // 2
    write_fastq(r[: half * 151], half, fq[0]); write_fastq(r[half * 151:], n_reads - half, fq[1])
    del r
    out = dict(n_reads=n_reads)
    try:
        for name in ("readset_load", "readset_load2", "readset_load3"):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); rs = _lib.ReadSet(fq, 0, 1); t1 = time.perf_counter()
            db.reset(); rs.scan_into(db); torch.cuda.synchronize(); t2 = time.perf_counter()
            db.reset(); rs.scan_into(db); torch.cuda.synchronize(); t3 = time.perf_counter()
            hits = int(db.counts_rows().astype(np.int64).sum())
            out[name] = dict(load_s=round(t1 - t0, 4), scan_s=round(t2 - t1, 4), scan2_s=round(t3 - t2, 4), hits=hits,
                             m_reads_per_s_load=round(n_reads / (t1 - t0) / 1e6, 1))
            rs.close()
        for name in ("scan_files", "scan_files2", "scan_files3"):
            db.reset(); t0 = time.perf_counter(); db.scan_files(fq); t1 = time.perf_counter()
            out[name] = dict(s=round(t1 - t0, 4), m_reads_per_s=round(n_reads / (t1 - t0) / 1e6, 1),
                             hits=int(db.counts_rows().astype(np.int64).sum()))
    finally:
        for p in fq: os.unlink(p)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

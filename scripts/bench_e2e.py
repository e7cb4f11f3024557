#!/usr/bin/env python3
"""End-to-end file -> counts rate (FASTQ text on the host -> parse -> PCIe -> scan), the
PCIe-inclusive number DESIGN.md quotes next to bench.py's HBM-resident `value`."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from strainscan_amd import _lib
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    spec = bench.make_db(torch, dev, 823, seed=20231013)
    db = _lib.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True)
    reads = bench.make_reads(torch, dev, spec, n_reads, seed=2, hit_frac=0.05).view(n_reads, 151)[:, :150].cpu().numpy()
    rec = np.empty((n_reads, 307), np.uint8)
    rec[:, 0:2] = np.frombuffer(b"@r", np.uint8)
    rec[:, 2] = 10
    rec[:, 3:153] = reads
    rec[:, 153] = 10
    rec[:, 154] = ord("+")
    rec[:, 155] = 10
    rec[:, 156:306] = ord("I")
    rec[:, 306] = 10
    d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path = os.path.join(d, "ss_e2e_%d.fq" % os.getpid())
    rec.tofile(path)
    size = os.path.getsize(path)
    out = {}
    try:
        for mode in ("parallel", "sequential"):
            if mode == "sequential":
                os.environ["SS_INGEST"] = "sequential"
            db.reset()
            db.scan_files([path])          # warm (page cache, pinned buffers)
            ts = []
            for _ in range(3):
                db.reset()
                t0 = time.perf_counter()
                nrec, nb = db.scan_files([path])
                ts.append(time.perf_counter() - t0)
            hits = int(db.counts_rows().astype(np.int64).sum())
            out[mode] = dict(s=min(ts), m_reads_per_s=round(n_reads / min(ts) / 1e6, 2),
                             fastq_gb_per_s=round(size / min(ts) / 1e9, 2), records=nrec, hits=hits)
            os.environ.pop("SS_INGEST", None)
    finally:
        os.unlink(path)
    assert out["parallel"]["hits"] == out["sequential"]["hits"]
    print(json.dumps(dict(n_reads=n_reads, fastq_bytes=size, host_threads=os.cpu_count(), **out)))


if __name__ == "__main__":
    main()

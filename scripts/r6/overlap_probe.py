#!/usr/bin/env python3
"""How well does a file-order tree scan overlap with the binning of the same reads on another stream?  (The lazy / asynchronous
binning that VERDICT round 5 item 4 proposes as the alternative to faster binning.)  -> JSON: ms alone and together."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from strainscan_amd import _lib
    dev = torch.device("cuda", 0)
    spec = bench.make_db(torch, dev, 823, seed=20231013, shape="sampled", hit_frac=0.05)
    n_reads = 20_000_000
    reads = bench.make_reads(torch, dev, spec, n_reads, seed=2, hit_frac=0.05)
    db = _lib.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True)
    side = torch.cuda.Stream(device=dev)
    out = {}

    def scan():
        db.scan_flat_dev(reads.data_ptr(), reads.numel(), side.cuda_stream)

    def binning(box):
        box.append(_lib.ReadSet.from_flat_dev(reads.data_ptr(), reads.numel(), order=True))

    for name in ("scan_alone", "bin_alone", "together", "scan_alone", "bin_alone", "together", "together"):
        torch.cuda.synchronize()
        db.reset(side.cuda_stream)
        torch.cuda.synchronize()
        box = []
        t0 = time.perf_counter()
        th = None
        if name in ("bin_alone", "together"):
            th = threading.Thread(target=binning, args=(box,))
            th.start()
        if name in ("scan_alone", "together"):
            scan()
        if th:
            th.join()
        torch.cuda.synchronize()
        out.setdefault(name, []).append(round((time.perf_counter() - t0) * 1e3, 3))
        for r in box:
            r.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

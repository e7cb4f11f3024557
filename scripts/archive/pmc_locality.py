#!/usr/bin/env python3
"""Three launches of the scan kernel over 20 M reads in file order (mode `file`) or binned (mode `bin`, SS_ORDER_BITS),
for rocprofv3 --pmc passes (scripts/gpu_pmc_loc.sh).  SS_MINI_XCD=0/1 selects the workgroup -> tile mapping."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "file"
    shape = sys.argv[2] if len(sys.argv) > 2 else "sampled"
    import torch
    from strainscan_amd import _lib
    dev = torch.device("cuda", 0)
    spec = bench.make_db(torch, dev, 823, seed=20231013, shape=shape, hit_frac=0.05)
    db = _lib.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True)
    reads = bench.make_reads(torch, dev, spec, 20_000_000, seed=2, hit_frac=0.05)
    stream = torch.cuda.current_stream().cuda_stream
    rs = _lib.ReadSet.from_flat_dev(reads.data_ptr(), reads.numel(), order=True) if mode == "bin" else None
    for _ in range(3):
        db.reset(stream)
        if rs is None:
            db.scan_flat_dev(reads.data_ptr(), reads.numel(), stream)
        else:
            rs.scan_into(db, stream)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()

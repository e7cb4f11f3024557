#!/usr/bin/env python3
"""Trip counts of scan_mini_kernel's loops per tile (a -DSS_COMB_STATS build: SS_LIB=build_tmp/lib_cstats.so) for bench.py's
workload in file order and binned: runs queued, runs looked up, found runs, lookup rounds, candidate rounds."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, torch
from strainscan_amd import _lib
shape = sys.argv[1] if len(sys.argv) > 1 else "sampled"
dev = torch.device("cuda", 0)
spec = bench.make_db(torch, dev, 823, seed=20231013, shape=shape, hit_frac=0.05)
db = _lib.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True)
reads = bench.make_reads(torch, dev, spec, 20_000_000, seed=2, hit_frac=0.05)
st = torch.cuda.current_stream().cuda_stream
rs = _lib.ReadSet.from_flat_dev(reads.data_ptr(), reads.numel(), order=True)
out = {}
for mode in ("file", "binned"):
    s16 = (C.c_ulonglong * 16)()
    _lib.lib().ss_debug_comb_stats(s16, 1)
    db.reset(st)
    if mode == "file":
        db.scan_flat_dev(reads.data_ptr(), reads.numel(), st)
    else:
        rs.scan_into(db, st)
    torch.cuda.synchronize()
    _lib.lib().ss_debug_comb_stats(s16, 1)
    v = list(s16)
    t = v[8]
    out[mode] = dict(tiles=t, runs_per_tile=round(v[9] / t, 2), looked_up_per_tile=round(v[10] / t, 2), found_per_tile=round(v[11] / t, 2),
                     lookup_rounds_per_tile=round(v[12] / t, 3), candidate_rounds_per_tile=round(v[13] / t, 3))
print(json.dumps({shape: out}))

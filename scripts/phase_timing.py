#!/usr/bin/env python3
"""Per-phase cycle counts of scan_mini_kernel (needs a -DSS_TIMING build: SS_LIB=build_tmp/libss_timing.so)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
from strainscan_amd import _lib
hf = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
dev = torch.device("cuda", 0)
spec = bench.make_db(torch, dev, 823, seed=20231013)
db = _lib.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True)
reads = bench.make_reads(torch, dev, spec, 20_000_000, seed=2, hit_frac=hf)
L = C.CDLL(_lib.LIB_PATH)
out = (C.c_ulonglong * 8)()
st = torch.cuda.current_stream().cuda_stream
db.scan_flat_dev(reads.data_ptr(), reads.numel(), st); torch.cuda.synchronize()
L.ss_debug_timing(out, 1)
db.reset(st)
db.scan_flat_dev(reads.data_ptr(), reads.numel(), st); torch.cuda.synchronize()
L.ss_debug_timing(out, 1)
v = np.array(list(out), float)
names = ["0 load/encode+barrier", "1a hash+barrier", "1b minimizer+queue+barrier", "2 dir lookup+barrier", "3 items", "end barrier", "", ""]
tot = v.sum()
for n, x in zip(names, v):
    if n: print("%-28s %6.1f %%  (%.0f cycles per tile per block)" % (n, 100 * x / tot, x / (reads.numel() / 4080)))

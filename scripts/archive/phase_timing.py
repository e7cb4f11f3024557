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
out = (C.c_ulonglong * 32)()
st = torch.cuda.current_stream().cuda_stream
db.scan_flat_dev(reads.data_ptr(), reads.numel(), st); torch.cuda.synchronize()
L.ss_debug_timing(out, 1)
db.reset(st)
db.scan_flat_dev(reads.data_ptr(), reads.numel(), st); torch.cuda.synchronize()
L.ss_debug_timing(out, 1)
v = np.array(list(out), float)
names = {0: "0 load/encode", 1: "1a m-mer keys", 6: "1b minimizers + run starts", 2: "1b run walk -> q1",
         7: "(lambda setup)", 3: "2 bloom + directory", 4: "3 candidates + overflow", 5: "end barrier"}
tot = v.sum()
tiles = reads.numel() / (62 * 16)
for i in (0, 1, 6, 2, 7, 3, 4, 5):
    print("%-28s %6.1f %%  (%.0f cycles per tile)" % (names[i], 100 * v[i] / tot, v[i] / tiles))

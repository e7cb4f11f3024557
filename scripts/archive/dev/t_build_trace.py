import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import torch, bench
from strainscan_amd import _lib
dev = torch.device("cuda", 0)
spec = bench.make_db(torch, dev, 823, seed=20231013, shape=sys.argv[1] if len(sys.argv) > 1 else "sampled", hit_frac=0.05)
for _ in range(2):
    t0 = time.perf_counter()
    db = _lib.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True)
    print("build %.3f s" % (time.perf_counter() - t0), file=sys.stderr)
    db.close()

#!/usr/bin/env python3
"""cProfile of detect_core at a typical cluster size (K = 300 k, S = 40)."""
import contextlib, cProfile, io, os, pstats, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.bench_l2 import make_case
from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 40
X, O, ids, y = make_case(K, S, {3 % S: 30.0, 57 % S: 11.0, 120 % S: 5.0})
npp = float(np.median(y[y != 0]) * 1000)
run = lambda: m.detect_core(X, O, ids, y.copy(), 31, 0, npp, npp, 0.9, [1], 0, 40, 0, 0)
with contextlib.redirect_stdout(io.StringIO()):
    run(); run()
    pr = cProfile.Profile(); pr.runcall(run)
pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(16)

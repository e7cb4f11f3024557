#!/usr/bin/env python3
"""Four 5 M x 300 clusters solved at once (bench.py l2_solve.four_clusters), with the phases of every solve.
   l2_four.py [threads = 4] [rows = 5000000] [strains = 300]"""
import contextlib, io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
if os.environ.get('L2_SWITCH'):
    sys.setswitchinterval(float(os.environ['L2_SWITCH']))
import scipy.sparse as sp
from concurrent.futures import ThreadPoolExecutor
from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m, l2 as L2
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Kc = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
S = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda", 0)
NSEG = 64
rs_ = np.random.RandomState(5)
pres = rs_.random_sample((S, NSEG)) < 0.35
seg = rs_.randint(0, NSEG, size=Kc)
lam = np.zeros(Kc)
for s_i, d in {3 % S: 30.0, 57 % S: 11.0, 120 % S: 5.0}.items():
    lam += pres[s_i, seg] * d
W = ((Kc + 31) // 32 + 3) & ~3
seg_d, pres_d = torch.from_numpy(seg).to(dev), torch.from_numpy(pres).to(dev)
wts = (1 << torch.arange(32, device=dev, dtype=torch.int64))
planes = np.zeros(S * W, np.uint32)
pad = W * 32 - Kc
for s_i in range(S):
    b = torch.cat([pres_d[s_i][seg_d], torch.zeros(pad, dtype=torch.bool, device=dev)]).view(W, 32).to(torch.int64)
    planes[s_i * W:(s_i + 1) * W] = (b * wts).sum(1).cpu().numpy().astype(np.uint32)
om = sp.csr_matrix(np.ones((Kc, 1), np.int8))
ids = ["S%03d" % i for i in range(S)]
ys = []
for f in (1.0, 0.8, 1.3, 0.6):
    v = rs_.poisson(lam * f).astype(np.int64)
    v[v == 1] = 0
    ys.append(v)
npp = [float(np.median(v[v != 0]) * 1000) for v in ys]
imgs = [L2.ClusterImage.from_planes(planes, Kc, S) for _ in range(4)]

def solve(i):
    tr = {}
    t0 = time.perf_counter()
    m.detect_core(None, om, ids, ys[i].copy(), 31, 0, npp[i], npp[i], 0.9, [1], 0, 40, 0, 0, trace=tr, img=imgs[i])
    return round((time.perf_counter() - t0) * 1e3, 1), {k: round(v, 1) for k, v in tr["timing_ms"].items()}

for rep in range(4):
    with ThreadPoolExecutor(max_workers=T) as pool:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):          # (once, around the threads: the redirection is process-wide)
            out = list(pool.map(solve, range(4)))
        wall = (time.perf_counter() - t0) * 1e3
    print("rep %d: wall %.1f ms" % (rep, wall))
    for o in out:
        print("   ", o)

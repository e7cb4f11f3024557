#!/usr/bin/env python3
"""all_strains_re.npz -> bit planes: the device route (members inflated / uploaded on the device, round 6) against np.load's.
    npz_route.py [K = 1000000] [S = 300]     -> one JSON line: seconds per route, deflated (what the reference's builder writes:
                                               sp.save_npz's default) and stored"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    import scipy.sparse as sp
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
    from strainscan_amd import l2 as L2
    from strainscan_amd import _lib
    rs = np.random.RandomState(1)
    seg = rs.randint(0, 64, size=K)                               # 64 segments, a strain carries a segment with p = 0.35
    pres = rs.random_sample((64, S)) < 0.35
    rows, cols = np.nonzero(pres[seg])
    X = sp.csr_matrix((np.ones(rows.size, np.int8), (rows, cols)), shape=(K, S))
    out = dict(K=K, S=S, nnz=int(X.nnz))
    with tempfile.TemporaryDirectory() as td:
        for name, comp in (("deflated", True), ("stored", False)):
            p = os.path.join(td, name + ".npz")
            t0 = time.perf_counter()
            sp.save_npz(p, X, compressed=comp)
            out[name] = dict(file_mb=round(os.path.getsize(p) / 1e6, 1), write_s=round(time.perf_counter() - t0, 2))
            ref = None
            for route in ("np.load", "device", "device_again"):
                _lib.lib().ss_device_sync()
                t0 = time.perf_counter()
                img = L2.ClusterImage(m._load_npz_csr(p)) if route == "np.load" else m._cluster_image_from_npz(p)
                _lib.lib().ss_device_sync()
                out[name][route + "_s"] = round(time.perf_counter() - t0, 3)
                if img is None:
                    out[name][route + "_s"] = None
                    continue
                pl = img.planes()
                if ref is None:
                    ref = pl
                out[name][route + "_equal"] = bool(np.array_equal(pl, ref))
                img.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

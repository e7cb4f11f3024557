#!/usr/bin/env python3
"""Layer-2 (intra-cluster) solve at the scale of BASELINE.json configs[3]: one large cluster,
K k-mers x S strains, a few strains present.  Times detect_core phases on the GPU and checks the
result against the CPU oracle on the same inputs (pre-scan quantities exact, abundances 1e-5)."""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_case(K, S, depths, seed=5, density=0.35):
    """K k-mers x S strains, CSC built column by column (no dense K x S array: 1.5 GB at K = 5 M, S = 300)."""
    rs = np.random.RandomState(seed)
    G = 64                                     # segments with a presence pattern over strains
    pres = rs.random_sample((S, G)) < density
    seg = rs.randint(0, G, size=K)
    indptr, indices = [0], []
    for s in range(S):
        r = np.nonzero(pres[s][seg])[0]
        indices.append(r.astype(np.int32))
        indptr.append(indptr[-1] + r.size)
    indices = np.concatenate(indices)
    X = sp.csc_matrix((np.ones(indices.size, np.int8), indices, np.array(indptr, np.int64)), shape=(K, S)).tocsr()
    lam = np.zeros(K)
    for s, d in depths.items():
        lam += pres[s, seg] * d
    y = rs.poisson(lam).astype(np.int64)
    y[y == 1] = 0
    O = sp.csr_matrix(np.ones((K, 1), np.int8))
    ids = ["S%03d" % i for i in range(S)]
    return X, O, ids, y


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    check = (len(sys.argv) > 3 and sys.argv[3] == "check")
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
    X, O, ids, y = make_case(K, S, {3 % S: 30.0, 57 % S: 11.0, 120 % S: 5.0})
    nz = y[y != 0]
    npp = float(np.median(nz) * 1000)
    trace = {}
    import contextlib, io
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            res = m.detect_core(X, O, ids, y.copy(), 31, 0, npp, npp, 0.9, [1], 0, 40, 0, 0, trace=trace)
        ts.append(time.perf_counter() - t0)
    # SURVEY 8(d) algorithmic bytes: a pre-scan pass reads the bit-packed matrix once, S * K / 8 B (+ K / 8 per mask);
    # the pattern statistics read the p selected bit planes + y once per fold: (p * K / 8 + 8 K) * 21
    out = dict(K=K, S=S, nnz=int(X.nnz), seconds=round(min(ts), 4), selected=list(res[0].keys()),
               algorithmic_bytes=dict(prescan_pass=S * K // 8 + K // 8,
                                      pattern_stats=(len(res[2]) * K // 8 + 8 * K) * 21),
               rel=[round(float(v), 6) for v in res[0].values()], n_rows=trace.get("n_rows"), p=trace.get("p"))
    if check:
        # the oracle on packed columns (oracle.prescan_packed: pinned to the dense prescan in the CPU tests; a dense K x S
        # array of this size would be 12 GB of int64), the regression on the selected columns only
        from oracle import oracle as orc
        t0 = time.perf_counter()
        cols, names, scov, sval, fsrc, depth = orc.prescan_packed(X, y, y, ids, 40 * 31, 0, 0, 0)
        keep = (y >= 0) & (y <= npp)
        Xs = X[:, cols].toarray()[keep]
        al, mse = orc.enet_cv(Xs, y[keep])
        a, _, _ = orc.lasso_mpm(al, mse)
        coef = orc.enet_fit(Xs, y[keep], a)
        out["oracle_seconds"] = round(time.perf_counter() - t0, 2)
        assert names == list(res[2].keys()), (names, list(res[2].keys()))
        assert {k: list(v) for k, v in res[2].items()} == {k: list(v) for k, v in scov.items()}
        assert {k: int(v) for k, v in res[3].items()} == {k: int(v) for k, v in sval.items()}
        assert np.allclose(trace["alphas_"], al, rtol=1e-12, atol=0) and np.allclose(trace["mse_path_"], mse, rtol=1e-7, atol=1e-9)
        rel = coef / coef.sum()
        got = np.array([float(res[0].get(n, 0.0)) for n in names])
        out["max_abs_diff_vs_oracle"] = float(np.abs(got - rel).max())
        out["checked"] = "prescan integers exact, alpha grid 1e-12, mse_path 1e-7, abundances < 1e-5 (oracle.prescan_packed + enet_cv + enet_fit)"
        assert out["max_abs_diff_vs_oracle"] < 1e-5
    print(json.dumps(out))


if __name__ == "__main__":
    main()

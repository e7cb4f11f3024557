"""ctypes/numpy face of the CPU oracle.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from strainscan_amd/.  Parity status: pinned against
reference-generated golden vectors (tests/golden/, tests/test_oracle_golden.py).

The integer/byte kernels live in ss_oracle.c; the intra-cluster pre-scan (dense numpy, small
cases only) and the ElasticNetCV driver are restated here in numpy, each function citing the
reference lines it follows.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libss_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libss_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        u64, i64, i32, dbl = C.c_uint64, C.c_int64, C.c_int, C.c_double
        vp = C.c_void_p
        L.orc_jellyfish_count.argtypes = [C.c_char_p, u64, i32, C.POINTER(C.c_char_p), C.POINTER(u64),
                                          i32, i32, u64, vp, vp]
        L.orc_jellyfish_count.restype = i32
        L.orc_count_flat.argtypes = [vp, u64, i32, vp, u64, vp, i32]
        L.orc_count_flat.restype = i32
        L.orc_match_node.argtypes = [vp, vp, u64, vp, u64, i64, C.POINTER(i64), C.POINTER(i64),
                                     C.POINTER(i64), C.POINTER(i64), C.POINTER(dbl)]
        L.orc_match_node.restype = i32
        L.orc_enet_cd_gram.argtypes = [vp, dbl, dbl, vp, vp, dbl, i32, i32, dbl, i32, C.POINTER(dbl)]
        L.orc_enet_cd_gram.restype = i32
        L.orc_enet_cd.argtypes = [vp, dbl, dbl, vp, vp, i64, i32, i32, dbl, i32, C.POINTER(dbl)]
        L.orc_enet_cd.restype = i32
        L.orc_revcomp.argtypes = [C.c_char_p, C.c_char_p, u64]
        L.orc_revcomp.restype = None
        L.orc_omp_threads.restype = i32
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------------------------
# a1 + a2: jellyfish count --if / dump -c + the Python tail (identify.py:73-103)
# ---------------------------------------------------------------------------------------------
def jellyfish_count(kmer_fa: bytes, reads, k=31, upper=True):
    """-> (counts uint32[n_rows], valid uint8[n_rows]); raises KeyError like identify.py:101."""
    if isinstance(reads, (bytes, bytearray)):
        reads = [reads]
    nl = kmer_fa.count(b"\n") + (1 if kmer_fa and not kmer_fa.endswith(b"\n") else 0)
    n_rows = nl // 2
    counts = np.zeros(n_rows, np.uint32)
    valid = np.zeros(n_rows, np.uint8)
    arr = (C.c_char_p * len(reads))(*[bytes(r) for r in reads])
    lens = (C.c_uint64 * len(reads))(*[len(r) for r in reads])
    rc = lib().orc_jellyfish_count(kmer_fa, len(kmer_fa), k, arr, lens, len(reads), int(upper),
                                   n_rows, _ptr(counts), _ptr(valid))
    if rc == -2:
        raise KeyError("dumped k-mer not in kmer_index_dict (identify.py:101)")
    if rc:
        raise RuntimeError("orc_jellyfish_count rc=%d" % rc)
    return counts, valid


def count_flat(db_keys, k, bases, threads=1):
    db_keys = np.ascontiguousarray(db_keys, np.uint64)
    b = np.frombuffer(bases, np.uint8) if isinstance(bases, (bytes, bytearray)) else np.ascontiguousarray(bases, np.uint8)
    counts = np.zeros(len(db_keys), np.uint32)
    rc = lib().orc_count_flat(_ptr(db_keys), len(db_keys), k, _ptr(b), b.size, _ptr(counts), threads)
    if rc:
        raise RuntimeError("orc_count_flat rc=%d" % rc)
    return counts


def encode_kmer(s, k=None):
    """2-bit key, A=0 C=1 G=2 T=3, first base most significant (oracle-side convention)."""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    v = 0
    for ch in s.upper():
        v = (v << 2) | code[ch]
    return v


# ---------------------------------------------------------------------------------------------
# a3: match_node + del_outlier (identify.py:106-127)
# ---------------------------------------------------------------------------------------------
def match_node(counts, valid, idx, min_valid=0):
    """-> dict(length, n_pos, n_kept, sum_kept, median)."""
    counts = np.ascontiguousarray(counts, np.uint32)
    valid = np.ascontiguousarray(valid, np.uint8)
    idx = np.ascontiguousarray(idx, np.int64)
    ln, npos, nk, sk = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    med = C.c_double()
    rc = lib().orc_match_node(_ptr(counts), _ptr(valid), counts.size, _ptr(idx), idx.size, min_valid,
                              C.byref(ln), C.byref(npos), C.byref(nk), C.byref(sk), C.byref(med))
    if rc:
        raise RuntimeError("orc_match_node rc=%d" % rc)
    return dict(length=ln.value, n_pos=npos.value, n_kept=nk.value, sum_kept=sk.value, median=med.value)


# ---------------------------------------------------------------------------------------------
# numpy.percentile(..., interpolation='nearest') as numpy 1.x computes it
# (used at identify_strains_L2_Enet_Pscan_new_sp.py:114-115,153-154)
# ---------------------------------------------------------------------------------------------
def percentile_nearest_index(n, q):
    """Index into the sorted array: round-half-even of q/100*(n-1)  [SURVEY 7, probed]."""
    pos = (q / 100.0) * (n - 1)
    return int(np.around(pos))


def percentile_nearest(a, q):
    s = np.sort(np.asarray(a))
    return s[percentile_nearest_index(s.size, q)]


# ---------------------------------------------------------------------------------------------
# a10-a12: Pre_Scan pieces (identify_strains_L2_Enet_Pscan_new_sp.py:33-49,94-175,228-373)
# dense numpy; small cases only
# ---------------------------------------------------------------------------------------------
def stat_cov(ix, iy):  # :33-43
    total = int(np.count_nonzero(ix))
    ic = ix.astype(np.int64) * iy
    valid = int(np.count_nonzero(ic > 1) + np.count_nonzero(ic < 0))
    return [float(valid / total) if total else 0, valid, total]


def optimize_dominat_y(pX, yy):  # :136-175
    res = []
    for c in range(pX.shape[1]):
        da = pX[:, c].astype(np.int64) * yy
        nz = da[da != 0]
        if nz.size < 1 or nz.sum() == 0:
            res.append(0)
            continue
        lo = percentile_nearest(nz, 5)
        hi = percentile_nearest(nz, 95)
        t = yy.copy()
        t[t < lo] = 0
        t[t > hi] = 0
        res.append(int(np.dot(pX[:, c].astype(np.int64), t)))
    res = np.array(res)
    return int(np.where(res == res.max())[0][0])


def get_avg_depth(dom, pX, yy):  # :110-120
    do = pX[:, dom].astype(np.int64) * yy
    do[do == 1] = 0
    nz = do[do != 0]
    f25 = percentile_nearest(nz, 25)
    f75 = percentile_nearest(nz, 75)
    nz = nz.copy()
    nz[nz < f25] = 0
    nz[nz > f75] = 0
    fin = nz[nz != 0]
    return float(np.mean(fin))


def prescan(pX, py, py_u, sid, cutoff, l2, pmode=0, emode=0):
    """Pre_Scan (:228-373).  pX dense K x S 0/1, py/py_u int64[K].
    -> (out_columns, out_strain, strain_cov, strain_val, final_src, dominat_avg_depth)"""
    pX = np.asarray(pX).astype(np.int64)
    py = np.asarray(py, np.int64)
    py_u = np.asarray(py_u, np.int64)
    K, S = pX.shape
    strain_cov, strain_val, strain_remainc, final_src = {}, {}, {}, {}
    pXt = pX.T
    cov_arr = np.array([stat_cov(pX[:, i], py)[0] for i in range(S)], dtype=float)
    default_cov = 0 if (pmode == 1 or emode == 1) else 0.7
    if cov_arr.max() > default_cov:
        keep = (cov_arr > default_cov).astype(np.int64)
        cov_arr = keep.astype(float)
        pXt_tem = (pXt.T * keep).T
    else:
        pXt_tem = pXt
        if cov_arr.max() < 0.01:
            l2 = 2
    yy = py_u if py_u.sum() > 0 else py
    if l2 == 2:
        dom = int(np.where(cov_arr == cov_arr.max())[0][0])
    else:
        dom = optimize_dominat_y(pX, yy)
    depth = get_avg_depth(dom, pX, yy)
    out_columns, out_strain = [dom], [sid[dom]]
    strain_cov[sid[dom]] = stat_cov(pX[:, dom], py)
    strain_val[sid[dom]] = strain_cov[sid[dom]][1]
    final_src[sid[dom]] = strain_cov[sid[dom]][0]
    used = pXt[dom].copy()
    # get_remainc (:94-108): once, always with py_u
    npXt = 2 * used + pXt_tem
    npXt[npXt > 1] = 0
    for i in range(S):
        if i == dom:
            continue
        all_k = int(npXt[i].sum())
        chk = int(np.count_nonzero(npXt[i] * py_u > 1))
        strain_remainc[i] = 0 if all_k == 0 else chk / all_k
    for _ in range(15):
        npXt = 2 * used + pXt_tem
        npXt[npXt > 1] = 0
        checks = [int(np.count_nonzero(npXt[i] * yy > 1)) for i in range(S)]
        # sorted(..., reverse=True) is stable: first index among equal maxima (:131-133)
        cand = int(np.argmax(checks))
        check = checks[cand]
        rc, cc = (0, 5000) if emode == 1 else (0.2, cutoff)
        if check >= cc:
            if strain_remainc.get(cand, strain_remainc.get(sid[cand])) > rc:
                out_columns.append(cand)
                out_strain.append(sid[cand])
                strain_cov[sid[cand]] = stat_cov(pX[:, cand], py)
                strain_val[sid[cand]] = check
                final_src[sid[cand]] = strain_remainc[cand]
            used = used + pXt[cand]
            used[used > 1] = 1
        else:
            break
    return out_columns, out_strain, strain_cov, strain_val, final_src, depth


def prescan_packed(X, py, py_u, sid, cutoff, l2, pmode=0, emode=0):
    """prescan() for inputs of BASELINE configs[3] size (K of millions x S of hundreds), where its dense int64
    K x S temporaries (8 K S bytes each) no longer fit a test's time: the same quantities from bit-packed columns.
    X is 0/1 (all_strains_re.npz holds int8 ones: Recls_withR_new.py:110-112), so `X[:, i] * v > 1` is `X[:, i] and
    v > 1`, a column AND a row mask, and the counts are popcounts.  Percentiles still see the actual values.
    Pinned to prescan() itself on the golden cases and on random ones (tests/test_oracle_golden.py).
    X: scipy sparse (any format) or dense, K x S."""
    import scipy.sparse as sp
    Xc = sp.csc_matrix(X)
    Xc.sum_duplicates()
    K, S = Xc.shape
    py = np.asarray(py, np.int64)
    py_u = np.asarray(py_u, np.int64)
    cols = [Xc.indices[Xc.indptr[i]:Xc.indptr[i + 1]][Xc.data[Xc.indptr[i]:Xc.indptr[i + 1]] != 0] for i in range(S)]

    def pack(rows_or_mask, is_mask=False):
        if is_mask:
            m = rows_or_mask
        else:
            m = np.zeros(K, bool)
            m[rows_or_mask] = True
        return np.packbits(m)

    P = [pack(c) for c in cols]
    popc = lambda a: int(np.bitwise_count(a).sum())          # noqa: E731
    y_gt1 = pack((py > 1) | (py < 0), True)
    total = [len(c) for c in cols]
    valid = [popc(P[i] & y_gt1) for i in range(S)]
    cov_arr = np.array([(valid[i] / total[i]) if total[i] else 0 for i in range(S)], dtype=float)
    default_cov = 0 if (pmode == 1 or emode == 1) else 0.7
    zero = np.zeros_like(P[0])
    if cov_arr.max() > default_cov:
        keep = cov_arr > default_cov
        cov_arr = keep.astype(float)
        Pt = [P[i] if keep[i] else zero for i in range(S)]
    else:
        Pt = P
        if cov_arr.max() < 0.01:
            l2 = 2
    yy = py_u if py_u.sum() > 0 else py
    if l2 == 2:
        dom = int(np.where(cov_arr == cov_arr.max())[0][0])
    else:
        res = []
        for c in range(S):
            da = yy[cols[c]]
            nz = da[da != 0]
            if nz.size < 1 or nz.sum() == 0:
                res.append(0)
                continue
            lo = percentile_nearest(nz, 5)
            hi = percentile_nearest(nz, 95)
            t = da.copy()
            t[t < lo] = 0
            t[t > hi] = 0
            res.append(int(t.sum()))
        res = np.array(res)
        dom = int(np.where(res == res.max())[0][0])
    do = yy[cols[dom]].copy()
    do[do == 1] = 0
    nz = do[do != 0]
    f25 = percentile_nearest(nz, 25)
    f75 = percentile_nearest(nz, 75)
    nz = nz.copy()
    nz[nz < f25] = 0
    nz[nz > f75] = 0
    depth = float(np.mean(nz[nz != 0]))
    strain_cov, strain_val, strain_remainc, final_src = {}, {}, {}, {}
    out_columns, out_strain = [dom], [sid[dom]]
    sc = lambda i: [float(valid[i] / total[i]) if total[i] else 0, valid[i], total[i]]   # noqa: E731
    strain_cov[sid[dom]] = sc(dom)
    strain_val[sid[dom]] = strain_cov[sid[dom]][1]
    final_src[sid[dom]] = strain_cov[sid[dom]][0]
    used = P[dom].copy()
    yu_gt1 = pack(py_u > 1, True)
    yy_gt1 = pack(yy > 1, True)
    for i in range(S):
        if i == dom:
            continue
        free = Pt[i] & ~used
        all_k = popc(free)
        strain_remainc[i] = 0 if all_k == 0 else popc(free & yu_gt1) / all_k
    for _ in range(15):
        checks = [popc(Pt[i] & ~used & yy_gt1) for i in range(S)]
        cand = int(np.argmax(checks))
        check = checks[cand]
        rc, cc = (0, 5000) if emode == 1 else (0.2, cutoff)
        if check >= cc:
            if strain_remainc.get(cand, strain_remainc.get(sid[cand])) > rc:
                out_columns.append(cand)
                out_strain.append(sid[cand])
                strain_cov[sid[cand]] = sc(cand)
                strain_val[sid[cand]] = check
                final_src[sid[cand]] = strain_remainc[cand]
            used = used | P[cand]
        else:
            break
    return out_columns, out_strain, strain_cov, strain_val, final_src, depth


# ---------------------------------------------------------------------------------------------
# a14-a16: ElasticNetCV -> lasso_mpm -> ElasticNet (identify_strains...:14-31,433-456)
# restated per SURVEY Appendix C (scikit-learn 0.23/0.24 semantics)
# ---------------------------------------------------------------------------------------------
def alpha_grid(X, y, l1_ratio=0.5, eps=1e-3, n_alphas=50):
    Xy = X.T.astype(np.float64) @ y.astype(np.float64)
    alpha_max = np.sqrt(Xy ** 2).max() / (len(y) * l1_ratio)
    if alpha_max <= np.finfo(float).resolution:
        return np.full(n_alphas, np.finfo(float).resolution)
    # np.logspace(np.log10(alpha_max * eps), np.log10(alpha_max), num)[::-1] with log10 / pow taken from libm, as numpy 1.17.3 of the
    # reference's environment.yaml takes them: later numpy builds bring their own SIMD log10 / pow, an ulp away now and then, and when
    # the cross-validation picks alphas[0] that ulp decides between a coefficient of 0 and one of 1e-16 (tests/golden/fuzz_reference.py)
    return np.array([math.pow(10.0, float(v)) for v in np.linspace(math.log10(alpha_max * eps), math.log10(alpha_max), num=n_alphas)])[::-1]


def shuffle_split(n, n_splits=20, test_size=0.5, seed=0):
    rng = np.random.RandomState(seed)
    n_test = int(math.ceil(test_size * n))
    n_train = n - n_test  # sklearn: n_train = floor((1-test_size)*n) == n - ceil(n/2) for .5
    for _ in range(n_splits):
        perm = rng.permutation(n)
        yield perm[n_test:n_test + n_train], perm[:n_test]


def enet_cd_gram(w, l1, l2, Q, q, yy, max_iter=5000, tol=1e-4, positive=True):
    w = np.ascontiguousarray(w, np.float64)
    Q = np.ascontiguousarray(Q, np.float64)
    q = np.ascontiguousarray(q, np.float64)
    gap = C.c_double()
    it = lib().orc_enet_cd_gram(_ptr(w), l1, l2, _ptr(Q), _ptr(q), yy, len(w), max_iter, tol,
                                int(positive), C.byref(gap))
    return w, gap.value, it


def enet_cd(w, l1, l2, X, y, max_iter=5000, tol=1e-4, positive=True):
    w = np.ascontiguousarray(w, np.float64)
    Xf = np.asfortranarray(X, np.float64)
    y = np.ascontiguousarray(y, np.float64)
    gap = C.c_double()
    it = lib().orc_enet_cd(_ptr(w), l1, l2, Xf.ctypes.data_as(C.c_void_p), _ptr(y), Xf.shape[0],
                           Xf.shape[1], max_iter, tol, int(positive), C.byref(gap))
    return w, gap.value, it


def enet_cv(X, y, l1_ratio=0.5, n_alphas=50, n_splits=20, max_iter=5000, tol=1e-4):
    """-> (alphas[50], mse_path[50, 20]) as ElasticNetCV(...).fit exposes them."""
    X = np.asarray(X, np.float64)
    y = np.asarray(y, np.float64)
    n, p = X.shape
    alphas = alpha_grid(X, y, l1_ratio, 1e-3, n_alphas)
    mse = np.empty((n_alphas, n_splits))
    for f, (tr, te) in enumerate(shuffle_split(n, n_splits)):
        Xt, yt = X[tr], y[tr]
        w = np.zeros(p)
        if len(tr) > p:  # precompute='auto'
            Q = Xt.T @ Xt
            q = Xt.T @ yt
            yy = float(yt @ yt)
        Xe, ye = X[te], y[te]          # (the test rows are the same for every alpha: gathered once per fold)
        for a, alpha in enumerate(alphas):
            l1 = alpha * l1_ratio * len(tr)
            l2 = alpha * (1.0 - l1_ratio) * len(tr)
            if len(tr) > p:
                w, _, _ = enet_cd_gram(w, l1, l2, Q, q, yy, max_iter, tol)
            else:
                w, _, _ = enet_cd(w, l1, l2, Xt, yt, max_iter, tol)
            r = Xe @ w - ye
            mse[a, f] = np.mean(r ** 2)
    return alphas, mse


def lasso_mpm(alphas, mse_path):  # identify_strains...:14-31
    mean = np.mean(mse_path, axis=1)
    std = np.std(mse_path, axis=1)
    i0 = int(np.argmin(mean))
    lo, hi = mean[i0] - std[i0], mean[i0] + std[i0]
    pick = i0
    for i in range(i0 - 1, -1, -1):
        if lo <= mean[i] <= hi:
            pick = i
    return alphas[pick], mean[pick], std[pick]


def enet_fit(X, y, alpha, l1_ratio=0.5, max_iter=5000, tol=1e-4):
    X = np.asarray(X, np.float64)
    y = np.asarray(y, np.float64)
    n = len(y)
    w, _, _ = enet_cd(np.zeros(X.shape[1]), alpha * l1_ratio * n, alpha * (1 - l1_ratio) * n, X, y,
                      max_iter, tol)
    return w


def revcomp(s: bytes) -> bytes:
    out = C.create_string_buffer(len(s))
    lib().orc_revcomp(s, out, len(s))
    return out.raw


# ---------------------------------------------------------------------------------------------
# The serial layer-2 pipeline of the reference, end to end (small cases): detect_strains
# (identify_strains_L2_Enet_Pscan_new_sp.py:177-478), vote_strain_L2's report
# (Vote_Strain_L2_Lasso_new_sp.py:334-438) and merge_res (:116-170), from the pieces above.
# Pinned to the reference's own report files (tests/golden/e2e_reports.json, l2_batch.json) in
# tests/test_oracle_golden.py; the -m gpu tests compare the product's reports of other samples with it.
# ---------------------------------------------------------------------------------------------
def detect_strains(X, O, sid, y, ksize, npp25, npp75, npp_out, all_cls, l2, msn, pmode=0, emode=0):
    """X: dense K x S 0/1, O: dense K x n_clusters 0/1, y int64[K] (1s already zeroed).
    -> (res, res2, strain_cov, strain_val, final_src) like the reference."""
    X = np.asarray(X).astype(np.int64)
    y = np.asarray(y, np.int64)
    om = np.asarray(O).astype(np.int64)[:, [int(a - 1) for a in all_cls]]
    ln = om.sum(axis=1)
    ln[ln > 1] = 0                                                   # :191-197
    cols, names, scov, sval, fsrc, depth = prescan(X, y, y * ln, sid, msn * ksize, l2, pmode, emode)
    if len(cols) == 1:                                               # :379-382
        return {names[0]: 1}, {names[0]: depth}, scov, sval, fsrc
    keep = ~((y < npp25) | (y > npp75) | (y > npp_out))              # :402-415 (a NaN bound keeps every row)
    Xs, ys = X[keep][:, cols], y[keep]
    alphas, mse = enet_cv(Xs, ys)
    alpha, _, _ = lasso_mpm(alphas, mse)
    coef = np.atleast_1d(enet_fit(Xs, ys, alpha))
    if not np.sum(coef) == 0:                                        # :465-471
        return dict(zip(names, list(coef / np.sum(coef)))), dict(zip(names, list(coef))), scov, sval, fsrc
    return {}, {}, scov, sval, fsrc


STRAINVOTE_HEADER = ("Strain_ID\tStrain_Name\tCluster_ID\tRelative_Abundance_Inside_Cluster\tPredicted_Depth (Enet)\t"
                     "Predicted_Depth (Ab*cls_depth)\tCoverage\tCoverd/Total_kmr\tValid_kmr\tRemain_Coverage\tCV\tExist_Evidence\n")


def vote_cluster(counts, X, O, sid, ksize, cls_ab, cls, all_cls, l2, msn, pmode=0, emode=0):
    """Vote_...:384-438 given the cluster's k-mer counts ordered by k-mer id: y = counts with 1s zeroed (remove_1 :312-322),
    npp_outlier = 1000 x median of the non-zero counts (:403-414), detect_strains, the report text ('' when nothing came back)."""
    y = np.asarray(counts, np.int64).copy()
    y[y == 1] = 0
    nz = y[y != 0]
    with np.errstate(invalid="ignore"):
        npp_out = np.median(nz) * 1000 if nz.size else float("nan")
    res, res2, scov, sval, fsrc = detect_strains(X, O, sid, y, int(ksize), 0, npp_out, npp_out, all_cls, l2, msn, pmode, emode)
    if len(res) == 0:
        return ""
    nr = sorted(res.items(), key=lambda d: d[1], reverse=True)
    tdep = sum(res2[n[0]] for n in nr)
    out = [STRAINVOTE_HEADER]
    for c, n in enumerate(nr, 1):
        name = n[0]
        body = ("\t" + cls + "\t" + str(n[1]) + "\t" + str(res2[name]) + "\t" + str((res2[name] / tdep) * cls_ab) + "\t" +
                str(scov[name][0]) + "\t" + str(scov[name][1]) + "/" + str(scov[name][2]) + "\t" + str(sval[name]) + "\t" + str(fsrc[name]))
        if n[1] > 0.02 and scov[name][0] > 0.7:
            out.append(str(c) + "\t" + name + body + "\t*\n")
        elif emode == 1:
            out.append(str(c) + "\t" + name + " (With_ExtraRegion_covered)" + body + "\t\n")
        else:
            out.append(str(c) + "\t" + name + body + "\t\n")
    return "".join(out)


def merge_reports(res, reports):
    """merge_res (:116-170): res = layer 1's dict, reports = {cluster id: StrainVote.report text or ''} -> final_report.txt text."""
    dinfo, total = {}, 0.0
    for r in res:
        if not res[r]["strain"] == 0:
            s = res[r]["strain"]
            total += float(res[r]["s_ab"])
            dinfo[s] = dict(cid="C" + str(r), pde="NA", pda=float(res[r]["s_ab"]), cov=float(res[r]["cls_cov"]),
                            ct=str(res[r]["cls_covered_num"]) + "/" + str(res[r]["cls_total_num"]))
        else:
            text = reports.get(r, "")
            if not text:
                continue
            pda = pde = 0.0
            tem = []
            for line in text.split("\n")[1:]:
                if not line.strip():
                    break
                ele = line.strip().split("\t")
                pda += float(ele[5])
                pde += float(ele[4])
                dinfo[ele[1]] = dict(cid=ele[2], pde=str(ele[4]), pda=float(ele[5]), cov=str(ele[6]), ct=str(ele[7]))
                tem.append(ele[1])
            if len(tem) == 1:
                total += pde
                dinfo[tem[0]]["pda"] = float(dinfo[tem[0]]["pde"])
            else:
                total += pda
    fr = sorted(((s, d["pda"] / total) for s, d in dinfo.items()), key=lambda d: d[1], reverse=True)
    out = ["ID\tStrain_Name\tCluster_ID\tRelative_Abundance\tPredicted_Depth (Enet)\tPredicted_Depth (Ab*cls_depth)\tCoverage\tCoverd/Total_kmr\n"]
    for c, (s, ab) in enumerate(fr, 1):
        d = dinfo[s]
        out.append(str(c) + "\t" + s + "\t" + d["cid"] + "\t" + str(ab) + "\t" + str(d["pde"]) + "\t" + str(d["pda"]) + "\t" +
                   str(d["cov"]) + "\t" + d["ct"] + "\n")
    return "".join(out)


def vote_batch(db_dir, reads, res, ksize=31, l2=0, msn=40, pmode=0, emode=0):
    """vote_strain_L2_batch (:247-311) on a database directory written by tests/synth.py (small clusters: the matrices are made
    dense) -> {relative path: text} of the report files the reference would write.  Exceptions propagate like the reference's."""
    import pickle
    import scipy.sparse as sp
    multi = [r for r in res if res[r]["strain"] == 0]
    if not multi:                                                    # check_L1_res (:68-74) == 1: generate_single_report (:232-244) + exit()
        ranked = sorted({res[c]["strain"]: res[c]["cls_per"] for c in res}.items(), key=lambda d: d[1], reverse=True)
        by_name = {res[c]["strain"]: c for c in res}
        out = ["Strain_ID\tStrain_Name\tCluster_ID\tRelative_Abundance_Inside_Cluster\tPredicted_Depth\tCoverage\tCovered/Total_kmr\n"]
        for i, (name, _) in enumerate(ranked, 1):
            e = res[by_name[name]]
            out.append("%d\t%s\tC%s\t%s\t%s\t%s\t%s/%s\n" % (i, name, by_name[name], str(e["cls_per"]), str(e["cls_ab"]), str(e["cls_cov"]),
                                                             str(e["cls_covered_num"]), str(e["cls_total_num"])))
        return {"final_report.txt": "".join(out)}
    reports = {}
    for r in (list(res) if len(res) == 1 else multi):
        cd = os.path.join(db_dir, "Kmer_Sets_L2", "Kmer_Sets", "C" + str(r))
        kfa = open(os.path.join(cd, "all_kmer.fasta"), "rb").read()
        counts, _ = jellyfish_count(kfa, reads, k=int(ksize), upper=False)
        X = sp.load_npz(os.path.join(cd, "all_strains_re.npz")).toarray()
        O = sp.load_npz(os.path.join(cd, "overlap_matrix.npz")).toarray()
        sid = pickle.load(open(os.path.join(cd, "id2strain_re.pkl"), "rb"))
        reports[r] = vote_cluster(counts, X, O, sid, ksize, res[r]["cls_ab"], "C" + str(r), list(res.keys()), l2, msn, pmode, emode)
    files = {"C%s/StrainVote.report" % r: t for r, t in reports.items() if t}
    if len(res) == 1:
        if files:
            files["final_report.txt"] = next(iter(files.values()))                      # `cp` at :273
    else:
        files["final_report.txt"] = merge_reports(res, reports)
    return files

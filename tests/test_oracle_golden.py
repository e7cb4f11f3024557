"""The CPU oracle (oracle/) against golden vectors produced by the real reference.

These tests pin the oracle (prompt item 3): every oracle function used as a checker elsewhere
is first checked here against outputs of jellyfish 2.3.0 / the reference Python / sklearn 0.24.2
recorded by tests/golden/make_golden.py.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests import scenarios as sc
from tests import synth


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_f1_jellyfish_semantics(golden_dir):
    g = _load(golden_dir, "f1_counts.json")
    c = sc.f1_case()
    assert synth.sha256_of(c["kmer_fa"], *c["reads"]) == g["sha256"]
    counts, valid = orc.jellyfish_count(c["kmer_fa"], c["reads"], k=31, upper=True)
    want = {int(k): v for k, v in g["match_results"].items()}
    got = {int(i): int(counts[i]) for i in np.nonzero(valid)[0]}
    assert got == want
    # the raw dump: every distinct ACGT k-mer of kmer.fa with its count, zeros included
    dump = {a: b for a, b in g["jellyfish_dump"]}
    rows = [r for r in c["kmer_fa"].split(b"\n")[1::2]]
    for i in np.nonzero(valid)[0]:
        assert dump[rows[i].decode().upper()] == counts[i]
    assert len(dump) == int(valid.sum())
    # identify_low_mem/low_depth key the dict by the raw text: a lower-case row -> KeyError
    # (identify_low_mem.py:81): the lower-case row is then a different key and its upper-case
    # twin row keeps the count; without a twin the reference dies with KeyError (:88)
    assert g["low_mem_error"] is None
    counts, valid = orc.jellyfish_count(c["kmer_fa"], c["reads"], k=31, upper=False)
    assert {int(i): int(counts[i]) for i in np.nonzero(valid)[0]} == \
        {int(k): v for k, v in g["low_mem_match_results"].items()}
    c2 = sc.f1_case(lower_only=True)
    lo = g["lower_only"]
    assert synth.sha256_of(c2["kmer_fa"]) == lo["sha256"]
    assert lo["identify_error"] is None and lo["low_mem_error"] == "KeyError" and lo["low_depth_error"] == "KeyError"
    counts, valid = orc.jellyfish_count(c2["kmer_fa"], c["reads"], k=31, upper=True)
    assert {int(i): int(counts[i]) for i in np.nonzero(valid)[0]} == {int(k): v for k, v in lo["match_results"].items()}
    with pytest.raises(KeyError):
        orc.jellyfish_count(c2["kmer_fa"], c["reads"], k=31, upper=False)


def test_l1_counts_and_node_profiles(golden_dir, l1_dbs, l1_reads):
    g = _load(golden_dir, "l1_search.json")
    for sname, (dbn, _, _) in sc.L1_SAMPLES.items():
        info = l1_dbs[dbn]
        tdb = os.path.join(info["db_dir"], "Tree_database")
        kfa = open(os.path.join(tdb, "kmer.fa"), "rb").read()
        reads = l1_reads[sname][1]
        assert synth.sha256_of(kfa, reads) == g[sname]["sha256"], sname
        counts, valid = orc.jellyfish_count(kfa, [reads], k=31, upper=True)
        assert synth.sha256_of(counts.tobytes()) == g[sname]["counts_sha256"], sname
        assert int(valid.sum()) == g[sname]["n_valid"]
        assert int(counts.sum()) == g[sname]["counts_sum"]
        # flat-stream counter (the format the device consumes) agrees with the FASTQ walker
        flat = synth.flat_bases_from_fastx(reads)
        rows = kfa.split(b"\n")[1::2]
        ok = np.array([b"N" not in r for r in rows])                       # database G: rows with an N are no keys at all
        keys = np.array([orc.encode_kmer(r.decode()) for r, o in zip(rows, ok) if o], np.uint64)
        assert np.array_equal(orc.count_flat(keys, 31, flat, threads=2), counts[ok]) and not counts[~ok].any()
        # root line of the reference's stdout trace: abundance | cov  length  (identify.py:240)
        run = g[sname]["runs"][0]
        assert run["module"] == "identify" and run["cutoff"] == [0.1, 0.4, 1]
        root = info["tree"].root
        tr = [t for t in run["trace"] if t[0] == root][0]
        if tr[1:] == ["weak"]:                           # database E: a weak root is not matched at all (:252-261)
            assert len(info["row_of_node"][root]) < 1000
            continue
        st = orc.match_node(counts, valid, np.array(info["row_of_node"][root]))
        cov = st["n_kept"] / st["length"]
        ab = st["sum_kept"] / st["n_kept"] if (st["n_kept"] and cov >= 0.1) else 0.0
        assert st["length"] == tr[3]
        assert abs(cov - tr[2]) < 5e-7 and abs(ab - tr[1]) < 5e-7


def test_percentile_nearest(golden_dir):
    for case in _load(golden_dir, "percentile_nearest.json"):
        a = np.array(case["a"])
        for q, want in case["q"].items():
            assert orc.percentile_nearest(a, float(q)) == want


def test_shuffle_split(golden_dir):
    g = _load(golden_dir, "shuffle_split.json")
    for n, e in g.items():
        folds = list(orc.shuffle_split(int(n)))
        assert folds[0][0].tolist() == e["train0"] and folds[0][1].tolist() == e["test0"]
        assert folds[19][0][:50].tolist() == e["train19"] and folds[19][1][:50].tolist() == e["test19"]
        blob = [np.concatenate([tr, te]).astype(np.int64).tobytes() for tr, te in folds]
        assert synth.sha256_of(*blob) == e["sha256"]


@pytest.mark.parametrize("name", sc.L2_CASES)
def test_l2_prescan_and_enet(golden_dir, name):
    g = _load(golden_dir, "l2_detect.json")[name]
    arrs = np.load(os.path.join(golden_dir, "l2_enet_arrays.npz"))
    case = sc.l2_case(name)
    X = case["X"].toarray()
    O = case["O"].toarray()
    y = case["y"]
    assert synth.sha256_of(X.tobytes(), O.tobytes(), y.tobytes()) == g["sha256"]
    ln = O[:, [c - 1 for c in case["all_cls"]]].sum(axis=1)
    ln[ln > 1] = 0
    y_u = y * ln
    cols, names, scov, sval, fsrc, depth = orc.prescan(X, y, y_u, case["ids"], case["msn"] * case["ksize"],
                                                       case["l2"], case["pmode"], case["emode"])
    assert set(names) == set(g["strain_cov"].keys())
    for nm in names:
        assert scov[nm] == g["strain_cov"][nm]
        assert float(sval[nm]) == float(g["strain_val"][nm])
        assert abs(fsrc[nm] - g["final_src"][nm]) < 1e-12
    if len(cols) == 1:
        assert g["res"] == {names[0]: 1}
        assert abs(g["res2"][names[0]] - depth) < 1e-9
        return
    keep = (y >= case["npp25"]) & (y <= case["npp75"]) & (y <= case["npp_out"])
    Xs, ys = X[keep][:, cols], y[keep]
    assert Xs.shape == (g["n_rows"], g["p"])
    alphas, mse = orc.enet_cv(Xs, ys)
    assert np.allclose(alphas, arrs[name + "_alphas"], rtol=1e-12, atol=0)
    assert np.allclose(mse, arrs[name + "_mse_path"], rtol=1e-9, atol=1e-9)
    alpha, _, _ = orc.lasso_mpm(alphas, mse)
    assert abs(alpha - g["alpha"]) <= 1e-12 * max(1.0, abs(alpha))
    coef = orc.enet_fit(Xs, ys, alpha)
    assert np.allclose(coef, arrs[name + "_coef"], rtol=1e-9, atol=1e-9)
    if coef.sum() == 0:
        assert g["res"] == {}
    else:
        for nm, c in zip(names, coef / coef.sum()):
            assert abs(c - g["res"][nm]) < 1e-9


def test_binom_table_matches_scipy(golden_dir):
    # host logic uses scipy.stats.binom on the GPU box as well; the table pins its values
    import scipy.stats as st
    t = np.load(os.path.join(golden_dir, "binom_table.npz"))
    tab = t["table"]
    for x in range(0, 61, 7):
        for y in range(0, 61, 5):
            assert abs((1 - st.binom.sf(max(x, y), x + y, 0.995)) - tab[x, y]) < 1e-12
    for x, y, v in t["big"]:
        assert abs((1 - st.binom.sf(max(x, y), x + y, 0.995)) - v) < 1e-12


def test_revcomp(golden_dir):
    for s, want in _load(golden_dir, "revcomp.json"):
        assert orc.revcomp(s.encode()).decode() == want


def test_prescan_packed_equals_prescan():
    """oracle.prescan_packed (bit-packed columns, for configs[3]-size inputs) against oracle.prescan (the literal
    restatement, pinned above to the reference's golden outputs): every golden scenario and random clusters, all
    modes."""
    from scripts.bench_l2 import make_case
    for name in sc.L2_CASES:
        c = sc.l2_case(name)
        X, y = c["X"].toarray(), c["y"]
        ln = c["O"].toarray()[:, [i - 1 for i in c["all_cls"]]].sum(1)
        ln[ln > 1] = 0
        yu = y * ln
        a = orc.prescan(X, y, yu, c["ids"], 40 * 31, c["l2"], c["pmode"], c["emode"])
        b = orc.prescan_packed(c["X"], y, yu, c["ids"], 40 * 31, c["l2"], c["pmode"], c["emode"])
        assert a == b, name
    rs = np.random.RandomState(1)
    for t in range(4):
        K, S = int(rs.randint(2000, 12000)), int(rs.randint(3, 24))
        X, O, ids, y = make_case(K, S, {0: 25.0, 1 % S: 9.0, 2 % S: float(rs.choice([0, 4.0]))}, seed=t,
                                 density=float(rs.uniform(0.2, 0.6)))
        yu = y.copy()
        yu[rs.random_sample(K) < 0.2] = 0
        for l2, pm, em in ((0, 0, 0), (1, 0, 0), (0, 0, 1), (0, 1, 0)):
            assert orc.prescan(X.toarray(), y, yu, ids, 1240, l2, pm, em) == orc.prescan_packed(X, y, yu, ids, 1240, l2, pm, em)


def _cmp_report_text(got, want, float_cols, tol=1e-5):
    gl, wl = got.strip().split("\n"), want.strip().split("\n")
    assert len(gl) == len(wl) and gl[0] == wl[0], (got, want)
    for a, b in zip(gl[1:], wl[1:]):
        fa, fb = a.split("\t"), b.split("\t")
        assert len(fa) == len(fb), (a, b)
        for i, (x, y) in enumerate(zip(fa, fb)):
            if i in float_cols and x != y:
                assert abs(float(x) - float(y)) <= tol * max(1.0, abs(float(y))), (i, a, b)
            else:
                assert x == y, (i, a, b)


@pytest.mark.parametrize("name", list(sc.L2_BATCH_CASES))
def test_oracle_vote_batch_equals_reference_reports(name, golden_dir, tmp_path):
    """The oracle's serial layer-2 pipeline (oracle.vote_batch = jellyfish_count + remove_1 + detect_strains + report + merge_res,
    restated from Vote_Strain_L2_Lasso_new_sp.py:116-170, 247-311, 334-438) writes the reference's own report files: the six
    hand-made layer-1 results of tests/scenarios.py, including the exception of the cluster without reads."""
    g = _load(golden_dir, "l2_batch.json")
    dbb, reads = sc.l2_batch_inputs(str(tmp_path))
    assert synth.sha256_of(reads) == g["sha256"]
    res, l2, emode = sc.L2_BATCH_CASES[name]
    want = g["cases"][name]
    err, files = None, {}
    try:
        files = orc.vote_batch(dbb, [reads], {k: dict(v) for k, v in res.items()}, 31, l2, 40, 0, emode)
    except Exception as e:
        err = type(e).__name__
    assert err == want["error"]
    if err is None:
        assert sorted(files) == sorted(want["files"])
        for rel, text in want["files"].items():
            _cmp_report_text(files[rel], text, (3, 4, 5, 6) if rel == "final_report.txt" and name != "one_cluster" else (3, 4, 5, 6, 8, 9))

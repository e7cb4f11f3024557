"""Host logic of layer 1 (tree walk) against the reference's golden results, on CPU.

The counts come from the oracle (the checker); the code under test is strainscan_amd.cst --
the same code the GPU path runs, only its provider differs."""
import json
import os

import pytest

from tests import hostlogic as hl
from tests import scenarios as sc


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "l1_search.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("sname", list(sc.L1_SAMPLES))
def test_walk_matches_reference(sname, golden, l1_dbs, l1_reads):
    dbn = sc.L1_SAMPLES[sname][0]
    tdb = os.path.join(l1_dbs[dbn]["db_dir"], "Tree_database")
    provs = {True: hl.OracleProvider(tdb, [l1_reads[sname][1]], upper=True),
             False: hl.OracleProvider(tdb, [l1_reads[sname][1]], upper=False)}
    for run in golden[sname]["runs"]:
        low_mem = run["module"] == "identify_low_mem"
        res, err, text = hl.run_walk(provs[not low_mem], tdb, run["cutoff"], low_mem, sc.POISSON_SEED)
        tag = (sname, run["module"], run["cutoff"])
        assert err == run["error"], (tag, err, text[-400:])
        if err is None:
            hl.assert_result_equal(res, run["result"], tag)
        got_tr = hl.parse_trace(text)
        assert len(got_tr) == len(run["trace"]), (tag, got_tr, run["trace"])
        for g, w in zip(got_tr, run["trace"]):
            assert g[0] == w[0] and len(g) == len(w), (tag, g, w)
            if len(w) == 4:
                assert abs(g[1] - w[1]) < 2e-6 and abs(g[2] - w[2]) < 2e-6 and g[3] == w[3], (tag, g, w)


def test_children_order_reversed_file(l1_dbs):
    """Two-leaf tree: the root is the last line, so the file is read backwards and the children
    are created as [2, 1] (identify.py:32-33)."""
    from strainscan_amd.tree import read_tree_structure
    tree, gcf = read_tree_structure(os.path.join(l1_dbs["C"]["db_dir"], "Tree_database"))
    assert [n.identifier for n in tree.all_nodes()] == [3, 2, 1]
    assert [n.identifier for n in tree.children(3)] == [2, 1]
    assert {n.identifier: s for n, s in gcf.items()} == {1: "GCF_ONLY1"}
    tree, _ = read_tree_structure(os.path.join(l1_dbs["A"]["db_dir"], "Tree_database"))
    assert [n.identifier for n in tree.all_nodes()] == [7, 8, 9, 10, 11, 1, 2, 3, 4, 5, 6]
    assert [n.identifier for n in tree.children(8)] == [9, 3]
    assert tree.paths_to_leaves()[0] == [7, 8, 9, 1]


@pytest.mark.parametrize("sname", list(sc.L1_SAMPLES))
def test_low_depth_ranks(sname, golden, l1_dbs, l1_reads):
    from strainscan_amd import identify_low_depth as ld
    from strainscan_amd.tree import read_tree_structure
    dbn = sc.L1_SAMPLES[sname][0]
    tdb = os.path.join(l1_dbs[dbn]["db_dir"], "Tree_database")
    pv = hl.OracleProvider(tdb, [l1_reads[sname][1]], upper=False)
    tree, _ = read_tree_structure(tdb)
    frac = {}
    for n in tree.all_nodes():
        ln, nk, _ = pv.node_stat(n.identifier)
        frac[n.identifier] = -1 if ln < ld.MIN_VALID else nk / ln
    got = ld.rank_paths(tree, frac)
    want = golden[sname]["ranks"]
    assert want["error"] is None
    assert [a for a, _ in got] == [a for a, _ in want["result"]]
    for (a, b), (_, wb) in zip(got, want["result"]):
        assert abs(b - wb) <= 1e-12 * max(1.0, abs(wb))

"""Test-only helpers: run the product's HOST logic (tree walk, report writers) on counts that
come from the oracle instead of the GPU, so that `-m "not gpu"` covers it without a device."""
import contextlib
import io
import os

import numpy as np

from oracle import oracle as orc


class OracleProvider:
    """Same interface as strainscan_amd.cst.ImageProvider, fed by oracle counts."""

    def __init__(self, tdb, reads_list, upper=True, min_valid=0):
        kfa = open(os.path.join(tdb, "kmer.fa"), "rb").read()
        self.counts, self.valid = orc.jellyfish_count(kfa, reads_list, k=31, upper=upper)
        self.tdb = tdb
        self._rows = {}

    def node_rows(self, node_id):
        if node_id not in self._rows:
            with open(os.path.join(self.tdb, "kmers", str(node_id))) as f:
                self._rows[node_id] = np.array(f.readline().split(), dtype=np.int64)
        return self._rows[node_id]

    def rows_stat(self, rows):
        s = orc.match_node(self.counts, self.valid, np.asarray(rows, np.int64))
        return s["length"], s["n_kept"], s["sum_kept"]

    def node_stat(self, node_id):
        return self.rows_stat(self.node_rows(node_id))


def run_walk(provider, tdb, cutoff, low_mem, seed):
    from strainscan_amd import cst
    lines = []

    def out(*a):
        lines.append(" ".join(str(x) for x in a))
    np.random.seed(seed)
    err, res = None, None
    try:
        res = cst.Walk(provider, tdb, list(cutoff), cst.Params(low_mem=low_mem), out=out).run()
    except BaseException as e:  # noqa: B902 -- the reference's exceptions are part of the contract
        err = type(e).__name__
    return res, err, "\n".join(lines)


def parse_trace(text):
    import re
    rx = re.compile(r"^(\d+):\s+(-?[\d.]+(?:e[-+]?\d+)?|nan) \| (-?[\d.]+(?:e[-+]?\d+)?|nan)\s+(\d+)$")
    out = []
    for ln in text.splitlines():
        m = rx.match(ln.strip())
        if m:
            out.append([int(m.group(1)), float(m.group(2)), float(m.group(3)), int(m.group(4))])
        elif ln.strip().endswith("weak") and ":" in ln:
            out.append([int(ln.split(":")[0]), "weak"])
    return out


def assert_result_equal(got, want, tag=""):
    assert (got is None) == (want is None), tag
    if want is None:
        return
    assert sorted(int(k) for k in got) == sorted(int(k) for k in want), (tag, dict(got), want)
    for k, w in want.items():
        g = got[int(k)]
        for f in ("cls_ab", "cls_per", "cls_cov", "s_ab"):
            assert abs(float(g[f]) - float(w[f])) <= 1e-9 * max(1.0, abs(float(w[f]))), (tag, k, f, g[f], w[f])
        assert int(g["cls_total_num"]) == int(w["cls_total_num"]), (tag, k)
        assert int(g["cls_covered_num"]) == int(w["cls_covered_num"]), (tag, k)
        assert g["strain"] == w["strain"], (tag, k)

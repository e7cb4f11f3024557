"""Cluster-search-tree (CST) walk: the host logic of layer 1.

Restates library/identify.py:45-70,130-164,167-228,231-504 (and the thresholds of
library/identify_low_mem.py) from SURVEY.md Appendix A.  All per-k-mer work -- the scan, the
per-node `match_node`/`del_outlier` reductions, the reduced profile of a reconstructed node --
comes from the device through a *provider* (strainscan_amd.db.TreeImage); what is left here is
the sequential tree logic, a few hundred scalar operations per sample.

Behaviours of the reference that look like bugs are kept on purpose (each marked KEPT): the
result dict, the stdout trace and the exceptions are the contract.
"""
import os
from collections import defaultdict

import numpy as np

from .tree import read_tree_structure


class Params:
    """identify.py vs identify_low_mem.py (full diff: SURVEY.md 3.3)."""

    def __init__(self, low_mem=False):
        self.low_mem = low_mem
        self.weak = 500 if low_mem else 1000          # identify.py:53 / identify_low_mem.py:50
        self.strong = 1500 if low_mem else 3000       # :58,67 / :52,61
        self.leaf_never_weak = not low_mem            # identify.py:54-57
        self.ancestor_min = 500 if low_mem else 1000  # :156 / :143
        self.adjust_min = 500 if low_mem else 1000    # :181 / :168
        self.qualified = not low_mem                  # :349-350,473-487
        self.root_fail_returns = not low_mem          # :243-245


def _mean(n_kept, sum_kept):
    # np.mean of a list of ints: exact integer sum (< 2**53) divided once -> one rounding
    if n_kept == 0:
        return float("nan")
    return float(sum_kept) / float(n_kept)


def piecewise(cov_cutoff, cov, label, mean_profile):
    """identify.py:130-136 (mean_profile = np.mean(k_profile))."""
    if label in [1, "o1"]:
        cov_cutoff = cov_cutoff / 2
    if cov >= cov_cutoff:
        return mean_profile
    return 0


def get_node_label(db_dir, tree, prm):
    """identify.py:45-70: categories from node_length.txt (file lengths, not valid lengths)."""
    length = {}
    with open(os.path.join(db_dir, "node_length.txt"), "r") as f:
        for line in f:
            d = line.rstrip().split("\t")
            length[int(d[0])] = int(d[1])
    leaves = set(id(n) for n in tree.leaves())
    for node in tree.all_nodes():
        ln = length[node.identifier]
        if ln < prm.weak:
            node.data[0] = 1 if (prm.leaf_never_weak and id(node) in leaves) else 0
        elif ln < prm.strong:
            node.data[0] = 1
        else:
            node.data[0] = 2
    with open(os.path.join(db_dir, "reconstructed_nodes.txt"), "r") as f:
        for line in f:
            node = tree.get_node(int(line.rstrip()))
            if node.data[0] != 0:
                node.data[0] = "o1" if length[node.identifier] < prm.strong else "o2"
    return length



def binom_sf(k, n, p):
    """scipy.stats.binom.sf(k, n, p) (identify.py:356) for integer k, n >= 0, without importing scipy (0.2-0.3 s of
    imports in a process whose whole identification takes 0.13 s).  rv_discrete.sf is 1 below the support and 0 at and
    beyond n; in between P(X > k) = P(n - X < n - k) = sum_{j < n-k} C(n, j) q^j p^(n-j): n - k = min(x, y) positive
    terms, built in log space from the ratio of consecutive terms (no cancellation; ~1e-13 relative).  The sibling
    test only asks whether 1 - sf < 0.05, and no pair of depths comes within 1e-9 of that threshold
    (tests/test_cst_host.py checks values, decisions and margin against scipy)."""
    import math
    k, n = int(k), int(n)
    if k < 0:
        return 1.0
    if k >= n:
        return 0.0
    m = n - k
    q = 1.0 - p
    j = np.arange(m - 1, dtype=np.float64)
    steps = np.log((n - j) / (j + 1.0)) + math.log(q / p)            # log(pmf(j + 1) / pmf(j))
    logs = n * math.log(p) + np.concatenate(([0.0], np.cumsum(steps)))
    return min(1.0, float(np.exp(logs).sum()))

class Walk:
    def __init__(self, provider, db_dir, cutoff, prm, out=print):
        self.pv = provider
        self.db_dir = db_dir
        self.prm = prm
        self.out = out
        self.cov_cutoff, self.wa_cov_cutoff, self.ab_cutoff = cutoff[0], cutoff[1], cutoff[2]
        self.tree, self.GCF = read_tree_structure(db_dir)
        for n in self.tree.all_nodes():
            n.data = [-1, -1, -1, -1, -1]          # [category, access, covered, total, abundance]
        get_node_label(db_dir, self.tree, prm)
        self.leaves = self.tree.leaves()
        self._leafset = set(id(n) for n in self.leaves)
        self.length, self.cov, self.abundance = {}, {}, {}
        self.results, self.alternative, self.qualified_parents = [], [], []
        self.overlapping_info = defaultdict(dict)
        self.pending = [[self.tree.all_nodes()[0]]]

    # -- device-backed pieces ---------------------------------------------------------------
    def match_node(self, node):
        """identify.py:115-127 -> (length, len(k_profile), np.mean(k_profile))."""
        st = self.pv.node_stat(node.identifier)
        return st[0], st[1], _mean(st[1], st[2])

    def _profile(self, node, length, n_kept, mean):
        self.length[node] = length
        self.cov[node] = n_kept / length
        self.abundance[node] = piecewise(self.cov_cutoff, self.cov[node], node.data[0], mean)

    # -- identify.py:139-164 ----------------------------------------------------------------
    def get_uniq_path(self, node, path):
        path.append(node)
        parent = self.tree.parent(node.identifier)
        if parent is None or self.tree.siblings(node.identifier)[0].data[1] in [1, 2]:
            return
        self.get_uniq_path(parent, path)

    def get_ancestor_ab(self, node):
        path = []
        self.get_uniq_path(node, path)
        kmer_number = {}
        valid = 0
        for N in path:
            kmer_number[N] = self.length[N] * self.cov[N]
            valid += self.length[N]
        total = sum(list(kmer_number.values()))
        if valid >= self.prm.ancestor_min:
            ratio, ab = [], []
            for N in path:
                ratio.append(kmer_number[N] / total)
                ab.append(self.abundance[N])
            return sum([a * b for a, b in zip(ab, ratio)])
        return -1

    # -- identify.py:167-228 ----------------------------------------------------------------
    def adjust_profile(self, node):
        prm = self.prm
        d = self.pv.node_rows(node.identifier)            # FILE order: overlap positions index it
        overlap = defaultdict(list)
        delete = set()
        for i in self.results:
            oi = self.overlapping_info
            if i.identifier in oi and node.identifier in oi[i.identifier]:
                pos = oi[i.identifier][node.identifier]
                overlap[i.identifier] = set(int(d[k]) for k in pos)
                oi[i.identifier][node.identifier] = iter(())   # KEPT: a one-shot map() at :442
                delete = overlap[i.identifier] | delete
        dset = set(int(x) for x in d)
        if len(dset) - len(delete) >= prm.adjust_min:
            remain = np.fromiter(dset - delete, np.int64)
            st = self.pv.rows_stat(remain)                 # device: match + del_outlier on `remain`
            self.length[node] = st[0]
            self.cov[node] = st[1] / self.length[node]     # ZeroDivisionError like the reference
            self.abundance[node] = piecewise(self.cov_cutoff, self.cov[node], node.data[0], _mean(st[1], st[2]))
            return 1 if self.length[node] < prm.strong else 2
        # fewer than adjust_min private k-mers: subtract what the reported clusters explain,
        # sampled from Poisson(abundance) with numpy's global legacy RNG exactly like :203-218
        counts, valid = self.pv.counts, self.pv.valid
        rows = np.fromiter(dset, np.int64)
        rows = rows[valid[rows] == 1]
        temp_match = {int(r): int(counts[r]) for r in rows}
        x = {i: i.data[4] for i in self.results}
        for i, _ in sorted(x.items(), key=lambda kv: (kv[1], kv[0]), reverse=True):
            temp1 = {}
            if i.identifier in overlap:
                for k in overlap[i.identifier]:
                    if k in temp_match and temp_match[k] > 0:
                        temp1[k] = temp_match[k]
            sample = np.random.poisson(self.abundance[i], size=len(temp1))
            sample.sort()
            order = sorted(temp1.items(), key=lambda kv: (kv[1], kv[0]))
            for k in range(0, len(sample)):
                temp_match[order[k][0]] = order[k][1] - sample[k]
        prof = [v for v in temp_match.values() if v > 0]   # no outlier cut in this branch
        self.length[node] = len(temp_match)
        self.cov[node] = len(prof) / self.length[node]
        mean = float(np.mean(prof)) if prof else float("nan")
        self.abundance[node] = piecewise(self.cov_cutoff, self.cov[node], node.data[0], mean)
        return "o1" if self.length[node] < prm.strong else "o2"

    # -- identify.py:231-372 ----------------------------------------------------------------
    def search(self, res_temp):
        pending, tree, out = self.pending, self.tree, self.out
        length, cov, abundance = self.length, self.cov, self.abundance
        group = pending[0]
        out("__________________________________________________")
        if len(group) == 1 and group[0].data[0] != 0:       # strong root
            node = group[0]
            node.data[1] = 1
            ln, nk, mean = self.match_node(node)
            self._profile(node, ln, nk, mean)
            out("%d:    %f | %f    %d" % (node.identifier, abundance[node], cov[node], length[node]))
            if abundance[node] >= self.ab_cutoff:
                pending.append(tree.children(node.identifier))
            elif self.prm.root_fail_returns:
                del pending[0]
                return
            if pending[1] == []:       # KEPT: IndexError in low_mem when the root fails (:230-234)
                res_temp.append(group[0])
                del pending[0]
                del pending[0]
            else:
                del pending[0]
            return
        if len(group) == 1 and group[0].data[0] == 0:        # weak root
            node = group[0]
            node.data[1] = 1
            length[node] = 0
            cov[node] = 0
            abundance[node] = 0
            out("%d:    weak" % node.identifier)
            pending.append(tree.children(node.identifier))
            del pending[0]
            return
        out("parent node: %d ->" % tree.parent(group[0].identifier).identifier)
        if group[0].data[0] == 0 and group[0].data[1] == 0:  # KEPT: tests only group[0] (:264)
            out("%d:    weak\n%d:    weak" % (group[0].identifier, group[1].identifier))
            group[0].data[1] = 2
            group[1].data[1] = 2
            for node in group:
                abundance[node] = 0
                cov[node] = 0
                length[node] = 0
                pending.append(tree.children(node.identifier))
            del pending[0]                                   # KEPT: and falls through (:273)

        correction_label = 0
        group_label = []
        weak_label = 0
        for node in group:
            if node.data[0] == 0:
                weak_label = 1
        for node in group:
            if node.data[0] == 0:
                abundance[node] = 0
                cov[node] = 0
                length[node] = 0
                node.data[1] = 2
                pending.append(tree.children(node.identifier))
                out("%d:    weak" % node.identifier)
                group_label.append((node, 0))
                continue
            elif node.data[0] in [1, 2] or len(self.results) == 0:
                if node.data[0] == "o1":
                    node.data[0] = 1
                elif node.data[0] == "o2":
                    node.data[0] = 2
                group_label.append((node, node.data[0]))
                ln, nk, mean = self.match_node(node)
                length[node] = ln
                if ln == 0:
                    abundance[node] = 0
                    cov[node] = 0
                    pending.append(tree.children(node.identifier))
                    out("%d:    weak" % node.identifier)
                    group_label.append((node, 0))
                else:
                    cov[node] = nk / ln
                    abundance[node] = piecewise(self.cov_cutoff, cov[node], node.data[0], mean)
            else:
                node.data[0] = self.adjust_profile(node)
                group_label.append((node, node.data[0]))
                if weak_label == 0:
                    correction_label = 1
            if abundance[node] < self.ab_cutoff:
                abundance[node] = 0
            out("%d:    %f | %f    %d" % (node.identifier, abundance[node], cov[node], length[node]))

        if correction_label == 1:
            ancestor_ab = self.get_ancestor_ab(tree.parent(group[0].identifier))
            if ancestor_ab > self.ab_cutoff:
                l0, l1 = group_label[0][1], group_label[1][1]
                label = 0
                if set([l0, l1]) in [set(["o1", "o1"]), set(["o2", "o2"])]:
                    label = 1
                elif 0 in set([l0, l1]) or set([l0, l1]) == set(["o1", "o2"]):
                    label = 2
                    for i in group_label:
                        if i[1] == 0 or i[1] == "o1":
                            x = i[0]
                        else:
                            y = i[0]
                elif set([l0, l1]) in [set(["o1", 2]), set(["o2", 2])]:
                    label = 2
                    for i in group_label:
                        if i[1] == 2:
                            y = i[0]
                        else:
                            x = i[0]
                if label == 1:
                    n0, n1 = group_label[0][0], group_label[1][0]
                    for i in [n0, n1]:
                        abundance[i] = ancestor_ab * (abundance[i] / (abundance[n0] + abundance[n1]))
                elif label == 2:
                    abundance[x] = ancestor_ab - abundance[y]   # KEPT: x/y may be unbound (:343)

        # binomial sibling test (:346-372)
        ab_temp = {}
        for i in range(0, 2):
            ab_temp[group[i]] = round(abundance[group[i]])
            if self.prm.qualified and cov[group[i]] >= 0.95:
                self.qualified_parents.append(group[i])
        if list(ab_temp.values()) == [0, 0]:
            del pending[0]
            return
        tup = sorted(ab_temp.items(), key=lambda kv: (kv[1]))
        (a, b, x, y) = (tup[1][0], tup[0][0], tup[1][1], tup[0][1])
        ret = 1 - binom_sf(max([x, y]), x + y, 0.995)
        keep = (a, b) if ret < 0.05 else [a]
        for i in keep:
            i.data[1] = 2 if i.data[0] == 0 else 1
            if id(i) not in self._leafset:
                ch = tree.children(i.identifier)
                if ch not in pending:
                    pending.append(ch)
            else:
                res_temp.append(i)
        del pending[0]

    # -- identify.py:375-399 ----------------------------------------------------------------
    def res_node_proc(self, node, wa_cov_cutoff):
        path = []
        self.get_uniq_path(node, path)
        for j in path:
            node.data[2] += self.length[j] * self.cov[j]
            node.data[3] += self.length[j]
        node.data[2] = int(node.data[2])
        if node.data[2] / node.data[3] < wa_cov_cutoff:
            return 0
        ratio, ab = [], []
        for j in path:
            ratio.append(self.cov[j] * self.length[j] / node.data[2])
            ab.append(self.abundance[j])
        node.data[4] = sum([a * b for a, b in zip(ab, ratio)])
        if node.data[4] <= 1:
            return 0
        return 1

    def check_access(self, node):
        while node is not None:
            node.data[1] = 1
            node = self.tree.parent(node.identifier)

    def _load_overlaps(self, j):
        base = os.path.join(self.db_dir, "overlapping_info", str(j.identifier))
        if not os.path.exists(base):
            return
        with open(base, "r") as f1, open(base + "_supple", "r") as f2:
            lines = f1.readlines()
            for line in f2.readlines():
                d = line.rstrip().split(" ")
                self.out(j.identifier, int(d[0]), int(d[1]))
                self.overlapping_info[j.identifier][int(d[0])] = list(
                    map(int, lines[int(d[1])].rstrip().split(" ")))

    # -- identify.py:402-504 ----------------------------------------------------------------
    def run(self):
        tree = self.tree
        while len(self.pending) != 0:
            res_temp = []
            self.search(res_temp)
            for j in res_temp:
                label = self.res_node_proc(j, self.wa_cov_cutoff)
                self.alternative.append(j)
                if label == 1:
                    self.check_access(j)
                    self.results.append(j)
                    self._load_overlaps(j)
                else:
                    j.data[1] = 0
        results = self.results
        for i in tree.all_nodes():
            i.data[1] = 0
        for i in results:
            self.check_access(i)
            i.data[2] = 0
            i.data[3] = 0
        for j in results:
            self.res_node_proc(j, self.wa_cov_cutoff)
        total_ab = 0
        if len(results) > 0:
            for i in results:
                total_ab += i.data[4]
        elif len(self.alternative) != 0:
            results = []
            cov_list = {}
            for j in self.alternative:
                cov_list[j] = j.data[2] / j.data[3]       # stale first-pass values (started at -1)
            r = max(cov_list, key=cov_list.get)
            if cov_list[r] >= 0.1:
                self.check_access(r)
                label = self.res_node_proc(j, 0.1)         # KEPT: stale loop variable j, not r (:467)
                if label == 1:
                    results = [r]
                    total_ab = r.data[4]
        if self.prm.qualified and len(results) == 0 and self.qualified_parents != []:
            qp = self.qualified_parents[-1].identifier
            cov_tmp = {}
            for node in self.cov:
                if id(node) in self._leafset and tree.is_ancestor(qp, node.identifier):
                    cov_tmp[node] = self.cov[node]
            max_key = max(cov_tmp, key=cov_tmp.get)        # ValueError on empty, like the reference
            results = [max_key]
            self.check_access(max_key)
            max_key.data[2] = 0
            max_key.data[3] = 0
            self.res_node_proc(max_key, self.wa_cov_cutoff)
            total_ab = max_key.data[4]
        res = defaultdict(lambda: {})
        for i in results:
            e = res[i.identifier]
            e["cls_ab"] = i.data[4]
            e["cls_per"] = i.data[4] / total_ab
            e["cls_cov"] = i.data[2] / i.data[3]
            e["cls_total_num"] = i.data[3]
            e["cls_covered_num"] = i.data[2]
            e["strain"] = 0
            e["s_ab"] = 0
            if i in self.GCF:
                e["strain"] = self.GCF[i]
                e["s_ab"] = i.data[4]
        return res


class ImageProvider:
    """Adapter TreeImage -> what Walk needs."""

    def __init__(self, image):
        self.img = image
        self._st = image.node_stats()

    def node_stat(self, node_id):
        s = self._st[self.img.node_index[node_id]]
        return int(s["length"]), int(s["n_kept"]), int(s["sum_kept"])

    def rows_stat(self, rows):
        s = self.img.rows_stat(rows)
        return int(s["length"]), int(s["n_kept"]), int(s["sum_kept"])

    def node_rows(self, node_id):
        return self.img.node_rows[node_id]

    @property
    def counts(self):
        return self.img.counts

    @property
    def valid(self):
        return self.img.valid

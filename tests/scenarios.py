"""Scenario definitions shared by tests/golden/make_golden.py and the parity tests.

Each scenario is rebuilt byte-identically from its seed by tests/synth.py; the golden files
pin a sha256 of the generated inputs next to the reference's outputs.
"""
import os

import numpy as np

from . import synth

K = 31

# ------------------------------------------------------------------------------------------------
# F1: jellyfish semantics -- tiny hand-made kmer.fa + reads (k = 31)
# ------------------------------------------------------------------------------------------------
def f1_case(lower_only=False):
    rs = np.random.RandomState(101)
    g = synth.rand_seq(rs, 400)
    kms = [g[i:i + K] for i in range(0, 120, 7)]
    rows = []
    for km in kms:
        rows.append(km)
        rows.append(synth.revcomp(km))
    rows.append(kms[0][:10] + b"N" + kms[0][11:])      # N row: never dumped -> not valid
    rows.append(kms[1].lower())                          # lower-case row: valid in identify.py only
    rows.append(kms[2])                                  # duplicate of row 4: last index wins
    rows.append(synth.rand_seq(rs, K))                   # absent from reads: count 0, still valid
    rows.append(kms[3][:K - 1])                          # 30-mer: yields no k-mer
    if lower_only:                                       # lower-case row WITHOUT an upper-case twin:
        rows.append(g[200:200 + K].lower())              # KeyError in identify_low_mem/low_depth
    kfa = b"".join(b">1\n" + r + b"\n" for r in rows)
    reads_fq = (
        b"@r1\n" + g[:150] + b"\n+\n" + b"I" * 150 + b"\n"
        b"@r2 rev\n" + synth.revcomp(g[:150]) + b"\n+\n" + b"I" * 150 + b"\n"
        b"@r3 lower\n" + g[:150].lower() + b"\n+\n" + b"A" * 150 + b"\n"      # quality = 'A's
        b"@r4 withN\n" + g[:40] + b"N" + g[41:150] + b"\n+\n" + b"@" + b"I" * 149 + b"\n"
        b"@r5 multiline\n" + g[:75] + b"\n" + g[75:150] + b"\n+\n" + b"I" * 75 + b"\n" + b"I" * 75 + b"\n"
        b"@r6 short\n" + g[:30] + b"\n+\n" + b"I" * 30 + b"\n"
        b"\n"
        b"@r7 exact\n" + g[7:7 + K] + b"\n+\n" + b"+" + b"I" * (K - 1) + b"\n"
    )
    reads_fa = b">a desc\n" + g[:60] + b"\n" + g[60:140] + b"\n>b\n" + g[100:160].lower() + b"\n"
    return dict(kmer_fa=kfa, reads=[reads_fq, reads_fa], n_rows=len(rows))


# ------------------------------------------------------------------------------------------------
# F2/F3: L1 cluster-search-tree databases
# ------------------------------------------------------------------------------------------------
#            7
#        8        10
#      9   3     4   11(weak)
#     1 2           5  6
PARENT_T11 = {1: 9, 2: 9, 3: 8, 4: 10, 5: 11, 6: 11, 7: None, 8: 7, 9: 8, 10: 7, 11: 10}

L1_DBS = {
    # deterministic adjust_profile branch (enough non-overlapping k-mers remain)
    "A": dict(parent=PARENT_T11,
              sites={1: 2500, 2: 1800, 3: 1000, 4: 2200, 5: 1700, 6: 400, 7: 2000, 8: 1600, 9: 1200,
                     10: 1500, 11: 300},
              seed=11, singleton={2: "GCF_SINGLE2", 6: "GCF_SINGLE6"},
              clusters={1: ["GCF_A1", "GCF_A2", "GCF_A3"], 3: ["GCF_C1", "GCF_C2"], 4: ["GCF_D1", "GCF_D2"],
                        5: ["GCF_E1", "GCF_E2", "GCF_E3", "GCF_E4"]},
              reconstructed=[4, 2, 10], overlaps=[(3, 4, 0, 300), (3, 2, 100, 400), (1, 10, 0, 200)]),
    # Poisson branch of adjust_profile (fewer than 1000 k-mers survive the overlap removal)
    "B": dict(parent=PARENT_T11,
              sites={1: 2500, 2: 1800, 3: 1000, 4: 2200, 5: 1700, 6: 1600, 7: 2000, 8: 1600, 9: 1200,
                     10: 1500, 11: 1500},
              seed=12, singleton={2: "GCF_SINGLE2"},
              clusters={1: ["GCF_A1", "GCF_A2"], 3: ["GCF_C1", "GCF_C2"], 4: ["GCF_D1", "GCF_D2"],
                        5: ["GCF_E1", "GCF_E2"], 6: ["GCF_F1", "GCF_F2"]},
              reconstructed=[4, 2, 5], overlaps=[(3, 4, 0, 1900), (3, 2, 0, 1500), (3, 5, 100, 1600)]),
    # two-leaf tree: root is the LAST line of tree_structure.txt -> reversed() creation order
    "C": dict(parent={1: 3, 2: 3, 3: None}, sites={1: 1600, 2: 1700, 3: 1500}, seed=13,
              singleton={1: "GCF_ONLY1"}, clusters={2: ["GCF_T1", "GCF_T2"]}, reconstructed=[], overlaps=[]),
    # single-cluster database (Build_tree.py:283-375): tree_structure.txt is "<id>\t" and the tree is in tree.pkl
    # (a pickled treelib.Tree, identify.py:19-21); the one node is root and leaf at once
    "D": dict(parent={1: None}, sites={1: 2600}, seed=14, singleton={}, clusters={1: ["GCF_U1", "GCF_U2", "GCF_U3"]},
              reconstructed=[], overlaps=[], single_cluster=True),
    # round 5: the branches of the walk the samples above never reach.
    # E: weak ROOT (identify.py:252-261, 400 rows: weak under both modules' thresholds); node 9 is 'o1' and visited while
    #    `results` is empty ('o1' -> 1, :290-291); 11 is weak in identify.py only, so with leaf 4 reported the unique path
    #    of get_ancestor_ab(11) holds 0 valid k-mers -> -1 (:163-164), while identify_low_mem.py (11 is label 1, both
    #    children 'o2' there) takes the rescaling branch label == 1 (:320-321, 338-340); leaf 1 (400 rows) is weak in
    #    identify_low_mem.py only: its children list [] goes into `pending` and the next search() raises IndexError
    "E": dict(parent=PARENT_T11,
              sites={1: 200, 2: 1800, 3: 1000, 4: 2200, 5: 1200, 6: 1300, 7: 200, 8: 1600, 9: 1200, 10: 1500, 11: 300},
              seed=15, singleton={4: "GCF_E4ONLY"},
              clusters={1: ["GCF_A1", "GCF_A2"], 2: ["GCF_B1", "GCF_B2"], 3: ["GCF_C1", "GCF_C2"],
                        5: ["GCF_E1", "GCF_E2"], 6: ["GCF_F1", "GCF_F2"]},
              reconstructed=[5, 6, 9], overlaps=[(3, 5, 0, 900), (3, 6, 0, 1000)]),
    # F: both children of a STRONG parent reconstructed and left with < 1000 private k-mers once leaf 3 is reported:
    #    [1, 2] come back 'o1','o1' -> label 1 (rescaled to the ancestor's abundance), [5, 6] 'o2','o1' -> label 2 with
    #    x = the 'o1' node (:322-328, 341-342); identify_low_mem.py sees 'o2','o2' twice
    "F": dict(parent=PARENT_T11,
              sites={1: 1400, 2: 1200, 3: 1000, 4: 2200, 5: 2500, 6: 1200, 7: 2000, 8: 1600, 9: 1200, 10: 1500, 11: 1500},
              seed=16, singleton={},
              clusters={1: ["GCF_A1", "GCF_A2"], 2: ["GCF_B1", "GCF_B2"], 3: ["GCF_C1", "GCF_C2"], 4: ["GCF_D1", "GCF_D2"],
                        5: ["GCF_E1", "GCF_E2"], 6: ["GCF_F1", "GCF_F2"]},
              reconstructed=[1, 2, 5, 6],
              overlaps=[(3, 1, 0, 1000), (3, 2, 0, 900), (3, 5, 0, 2100), (3, 6, 0, 900)]),
    # G: nodes whose every kmer.fa row carries an N -- node_length.txt says 1200 / 1400 rows (label 1), no row is valid:
    #    length == 0 (:298-303).  Internal node 11 comes FIRST in its group ([11, 4]: creation order), so with 4
    #    reconstructed and leaf 3 reported the labels are [(11, 1), (11, 0), (4, 'o1')] -> `0 in set` -> label 2, x = 4
    #    ('o2' and y = 4 under identify_low_mem.py); leaf 1 pushes [] and the next search() raises IndexError
    "G": dict(parent=PARENT_T11,
              sites={1: 700, 2: 1800, 3: 1000, 4: 1400, 5: 1700, 6: 1600, 7: 2000, 8: 1600, 9: 1200, 10: 1500, 11: 600},
              seed=17, singleton={},
              clusters={1: ["GCF_A1", "GCF_A2"], 2: ["GCF_B1", "GCF_B2"], 3: ["GCF_C1", "GCF_C2"], 4: ["GCF_D1", "GCF_D2"],
                        5: ["GCF_E1", "GCF_E2"], 6: ["GCF_F1", "GCF_F2"]},
              reconstructed=[4], overlaps=[(3, 4, 0, 1100)], invalid_nodes=[1, 11]),
}

# samples: name -> (db, [(leaf or ('path', node) or ('random', length), depth)], read seed)
L1_SAMPLES = {
    "A_mix3": ("A", [(1, 20.0), (3, 8.0), (4, 5.0)], 201),
    "A_leaf5": ("A", [(5, 12.0)], 202),
    "A_leaf6_single": ("A", [(6, 15.0), (2, 6.0)], 203),
    "A_low": ("A", [(1, 0.6)], 204),
    "A_verylow": ("A", [(3, 0.15)], 205),
    "A_none": ("A", [(("random", 60000), 5.0)], 206),
    "A_novel": ("A", [(("path", 9), 15.0)], 207),
    "B_mix": ("B", [(3, 12.0), (4, 6.0), (2, 9.0), (5, 4.0)], 211),
    "C_two": ("C", [(1, 10.0), (2, 3.0)], 221),
    "D_one": ("D", [(1, 9.0)], 231),
    "D_low": ("D", [(1, 0.5)], 232),
    "D_none": ("D", [(("random", 40000), 5.0)], 233),
    # round 5 (see the databases E, F, G above)
    "E_mix": ("E", [(3, 12.0), (4, 6.0), (5, 5.0), (6, 4.0)], 241),
    "E_weakleaf": ("E", [(2, 8.0), (3, 5.0)], 242),
    "F_mix": ("F", [(3, 12.0), (1, 6.0), (2, 5.0), (5, 6.0), (6, 4.0)], 251),
    # a third / a quarter of every node on two leaves' paths at 10x: both leaves fail the weighted coverage cutoff 0.4,
    # nothing is reported, and the best `alternative` (leaf 4) is taken after res_node_proc ran a SECOND time on the
    # stale loop variable j = leaf 1 and succeeded (identify.py:459-470: label == 1 with j != r)
    "F_slices": ("F", [(("slice", 4, 0.0, 0.35), 10.0), (("slice", 1, 0.0, 0.25), 10.0)], 252),
    "G_mix": ("G", [(3, 12.0), (4, 6.0), (5, 5.0)], 261),
    "G_zero_leaf": ("G", [(3, 10.0), (2, 7.0)], 262),
}

CUTOFFS = [[0.1, 0.4, 1], [0.05, 0.05, 1], [0.01, 0.05, 1], [0.005, 0.01, 1]]   # StrainScan.py:196-216
POISSON_SEED = 4321

# Single search() steps from a hand-made state (identify.py:231-372 / identify_low_mem.py:218-354 called directly by
# make_golden.py).  The "both weak" branch (identify.py:264-273) tests `group[0].data[1] == 0`; access flags start at -1
# and only a rejected LEAF is ever set to 0, after its group has left `pending` for good (a group enters `pending` once:
# children are pushed when their parent is processed, every later push is guarded or belongs to this very branch) -- so
# identify_cluster() can never take it.  The function is still there and its fall-through is behaviour, so it
# is pinned at the function: category / access of some nodes overridden, `pending` given, one call, everything it touched
# recorded.  name -> (sample, module, cutoff, {node: [category, access]}, pending as node ids)
L1_STEPS = {
    # two weak siblings, a third group waiting: the branch pushes both children lists and drops the group, falls through,
    # the weak loop pushes the children AGAIN and the final `del pending[0]` removes the waiting group [11, 4]
    "both_weak_third_group": ("E_mix", "identify", [0.1, 0.4, 1], {9: [0, 0], 3: [0, -1]}, [[9, 3], [11, 4]]),
    # only group[0] is tested: the second node is strong, is matched and reported; nothing else waits, so the final
    # delete removes the first children list the branch itself had pushed
    "both_weak_second_strong": ("E_mix", "identify", [0.1, 0.4, 1], {11: [0, 0]}, [[11, 4]]),
    "both_weak_low_mem": ("E_mix", "identify_low_mem", [0.05, 0.05, 1], {5: [0, 0]}, [[5, 6], [9, 3], [11, 4]]),
}


def build_l1(name, root_dir):
    spec = L1_DBS[name]
    db_dir = os.path.join(root_dir, "DB_" + name)
    info = synth.build_l1_db(db_dir, spec["parent"], spec["sites"], spec["seed"], spec.get("singleton"),
                             spec.get("clusters"), spec.get("reconstructed", ()), spec.get("overlaps", ()),
                             single_cluster=spec.get("single_cluster", False), invalid_nodes=spec.get("invalid_nodes", ()))
    info["db_dir"] = db_dir
    return info


def sample_reads(info, sample_name):
    db, mix, seed = L1_SAMPLES[sample_name]
    rs = np.random.RandomState(seed + 5000)
    gd = []
    for src, depth in mix:
        if isinstance(src, tuple) and src[0] == "random":
            g = synth.rand_seq(rs, src[1])
        elif isinstance(src, tuple) and src[0] == "path":
            g = b"".join(info["node_seq"][i] for i in info["tree"].path(src[1]))
        elif isinstance(src, tuple) and src[0] == "slice":           # the same fraction of every node on the leaf's path
            g = b"N".join(info["node_seq"][i][int(src[2] * len(info["node_seq"][i])):int(src[3] * len(info["node_seq"][i]))]
                          for i in info["tree"].path(src[1]))
        else:
            g = info["leaf_genome"][src]
        gd.append((g, depth))
    return synth.simulate_reads(gd, seed)


# ------------------------------------------------------------------------------------------------
# F4/F5: intra-cluster (L2) cases
# ------------------------------------------------------------------------------------------------
def l2_case(name):
    """-> dict(X csr K x S int8, O csr K x C int8, ids, y int64[K], kwargs for detect_strains)."""
    import scipy.sparse as sp
    P4 = [[1, 1, 0, 0, 1], [1, 0, 1, 0, 0], [0, 1, 1, 1, 0], [0, 0, 0, 1, 1]]
    P5 = [[1, 1, 0, 0, 1, 0], [1, 0, 1, 0, 0, 1], [0, 1, 1, 1, 0, 0], [0, 0, 0, 1, 1, 1], [1, 1, 1, 0, 0, 0]]
    rsm = np.random.RandomState(77)
    P12 = (rsm.random_sample((12, 16)) < 0.45).astype(int).tolist()
    cases = {
        # name: (seed, sites per segment, presence rows, depth per strain, all_cls, l2, emode, n_cls, outliers)
        "one": (31, [700, 500, 600, 400], [[1, 1, 0, 0], [1, 0, 1, 0], [0, 1, 1, 1]], [14, 0, 0], [2], 0, 0, 4, 0),
        "two": (32, [900, 800, 700, 600, 500], P4, [30, 0, 11, 0], [2, 3], 0, 0, 4, 0),
        "two_out": (32, [900, 800, 700, 600, 500], P4, [30, 0, 11, 0], [2, 3], 0, 0, 4, 0.002),
        "three": (33, [900, 800, 700, 600, 500, 900], P5, [40, 12, 0, 6, 0], [1, 2], 0, 0, 3, 0),
        "three_l2": (33, [900, 800, 700, 600, 500, 900], P5, [3, 1.2, 0, 0.8, 0], [1, 2], 1, 0, 3, 0),
        "emode": (34, [900, 800, 700, 600, 500], P4, [25, 0, 9, 3], [2], 0, 1, 4, 0),
        "lowcov": (35, [900, 800, 700, 600], [[1, 1, 0, 0], [1, 0, 1, 0], [0, 1, 1, 1]], [0.3, 0, 0], [1], 1, 0, 2, 0),
        "many": (36, [400] * 16, P12, [22, 0, 0, 9, 0, 0, 0, 0, 4, 0, 0, 0], [3], 0, 0, 5, 0),
    }
    seed, segs, pres, depths, all_cls, l2, emode, n_cls, outl = cases[name]
    rs = np.random.RandomState(seed)
    pres = np.array(pres, bool)
    S, G = pres.shape
    seg_of_row = np.repeat(np.arange(G), [2 * n for n in segs])
    Kn = seg_of_row.size
    Xd = pres[:, seg_of_row].T.astype(np.int8)                       # K x S
    cid = all_cls[0]
    O = np.zeros((Kn, n_cls), np.int8)
    O[:, cid - 1] = 1
    if len(all_cls) > 1:                                            # some k-mers shared with 2nd cluster
        shared = rs.random_sample(Kn) < 0.15
        O[shared, all_cls[1] - 1] = 1
    lam = Xd.astype(np.float64) @ (np.array(depths, float) * 0.4)  # strand-specific k-mer depth
    y = rs.poisson(lam).astype(np.int64)
    if outl:
        y[rs.random_sample(Kn) < outl] += 5000                      # a few repeats/outliers
    y[y == 1] = 0                                                   # remove_1 (Vote_...:312-322)
    ids = ["GCF_%s_%d" % (name.upper(), i + 1) for i in range(S)]
    nz = y[y != 0]
    npp_out = float(np.median(nz) * 1000) if nz.size else 0.0
    return dict(X=sp.csr_matrix(Xd), O=sp.csr_matrix(O), ids=ids, y=y, ksize=31, npp25=0, npp75=npp_out,
                npp_out=npp_out, cls_cov=0.9, all_cls=all_cls, l2=l2, msn=40, pmode=0, emode=emode)


L2_CASES = ["one", "two", "two_out", "three", "three_l2", "emode", "lowcov", "many"]


# ------------------------------------------------------------------------------------------------
# vote_strain_L2_batch on hand-made layer-1 results: the branches of Vote_Strain_L2_Lasso_new_sp.py:247-311, 417-438
# the end-to-end samples leave out (one identified cluster -> `cp`; a strain below the evidence bar with and without
# -e; a cluster whose regression comes back all zero -> no report, skipped by merge_res; a cluster without reads ->
# the IndexError of np.percentile([]) at identify_strains_L2_Enet_Pscan_new_sp.py:114)
# ------------------------------------------------------------------------------------------------
def l2_batch_inputs(root):
    """-> (db_dir, FASTQ bytes).  Clusters 1-4 of 5 have k-mer sets; 1: two strains at 16x / 6x, 2: one strain at 0.6x,
    3: two strains at 30x / 11x plus 13 of its k-mers repeated 5000 times (reads of exactly 31 bases), 4: no reads."""
    db = os.path.join(root, "dbL2B")
    pres = [[1, 1, 0, 0, 1], [1, 0, 1, 0, 0], [0, 1, 1, 1, 0]]
    c1 = synth.build_l2_cluster(db, 1, 5, ["GCF_1_0", "GCF_1_1", "GCF_1_2"], [1500, 1200, 1000, 1400, 900], pres, seed=81)
    c2 = synth.build_l2_cluster(db, 2, 5, ["GCF_2_0", "GCF_2_1", "GCF_2_2"], [1500, 1200, 1000, 1400, 900], pres, seed=82)
    c3 = synth.build_l2_cluster(db, 3, 5, ["GCF_3_0", "GCF_3_1"], [1300, 1100, 900], [[1, 1, 0], [1, 0, 1]], seed=83)
    synth.build_l2_cluster(db, 4, 5, ["GCF_4_0", "GCF_4_1"], [1200, 1000, 800], [[1, 1, 0], [1, 0, 1]], seed=84)
    mix = [(c1["strain_extra"]["GCF_1_0"], 16.0), (c1["strain_extra"]["GCF_1_2"], 6.0), (c2["strain_extra"]["GCF_2_1"], 0.6),
           (c3["strain_extra"]["GCF_3_0"], 30.0), (c3["strain_extra"]["GCF_3_1"], 11.0)]
    reads = synth.simulate_reads(mix, 505)
    rs = np.random.RandomState(99)
    kms = list(c3["kid"])
    pick = [kms[i] for i in rs.choice(len(kms), 13, replace=False)]
    reads += b"".join(b"@o%d\n%s\n+\n%s\n" % (i, km.encode(), b"I" * 31) * 5000 for i, km in enumerate(pick))
    return db, reads


def _l1_entry(cid, strain=0):
    return dict(strain=strain, cls_ab=20.0 + cid, cls_cov=0.9, cls_per=0.25, s_ab=(4.0 if strain else 0),
                cls_covered_num=10, cls_total_num=12)


# name -> (layer-1 result dict, l2, emode)
L2_BATCH_CASES = {
    "one_cluster": ({1: _l1_entry(1)}, 0, 0),
    "mixed": ({1: _l1_entry(1), 2: _l1_entry(2), 5: _l1_entry(5, "GCF_single")}, 1, 0),
    "mixed_emode": ({1: _l1_entry(1), 2: _l1_entry(2), 5: _l1_entry(5, "GCF_single")}, 1, 1),
    "empty_res": ({1: _l1_entry(1), 3: _l1_entry(3)}, 0, 0),
    "no_reads": ({1: _l1_entry(1), 4: _l1_entry(4)}, 0, 0),
    "only_no_reads": ({4: _l1_entry(4)}, 0, 0),
}


# ------------------------------------------------------------------------------------------------
# `-k 25` (StrainScan.py:136, 266-271 hands ksize to the layer-2 scans: Vote_Strain_L2_Lasso_new_sp.py:359-371, and to the
# pre-scan's cutoff msn * k: identify_strains_L2_Enet_Pscan_new_sp.py:226): a cluster whose all_kmer.fasta holds 25-mers
# ------------------------------------------------------------------------------------------------
def l2_k25_inputs(root):
    """-> (db_dir, FASTQ bytes): clusters 1 and 2 of 3 with k = 25 k-mer sets; three strains of cluster 1 at 20x / 8x / 3x, one of cluster 2."""
    db = os.path.join(root, "dbL2K25")
    pres = [[1, 1, 0, 0, 1, 0], [1, 0, 1, 0, 0, 1], [0, 1, 1, 1, 0, 0], [1, 0, 0, 1, 1, 1]]
    c1 = synth.build_l2_cluster(db, 1, 3, ["GCF_K1_0", "GCF_K1_1", "GCF_K1_2", "GCF_K1_3"], [1500, 1200, 1000, 1400, 900, 1100], pres, seed=91, k=25)
    c2 = synth.build_l2_cluster(db, 2, 3, ["GCF_K2_0", "GCF_K2_1"], [1300, 1100, 900], [[1, 1, 0], [1, 0, 1]], seed=92, k=25)
    mix = [(c1["strain_extra"]["GCF_K1_0"], 20.0), (c1["strain_extra"]["GCF_K1_2"], 8.0), (c1["strain_extra"]["GCF_K1_3"], 3.0),
           (c2["strain_extra"]["GCF_K2_1"], 12.0)]
    return db, synth.simulate_reads(mix, 515)


L2_K25_RES = {1: _l1_entry(1), 2: _l1_entry(2)}

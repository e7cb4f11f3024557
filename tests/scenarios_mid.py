"""Mid-size scenarios (round 6): BASELINE.json configs[0]'s SHAPE run through the real reference.

configs[0] is "A. muciniphila 157-strain DB, Sim_Data.zip SE reads": 53 clusters / 157 strains (README.md:111 of the
reference), hence a 105-node cluster search tree.  The real database and reads are not in the mount, so
the shape is rebuilt from seeds -- tests/synth.py writes the reference's on-disk formats -- and
tests/golden/make_golden.py runs the reference's WHOLE flow on it (StrainScan.py:113-271: cutoff ladder,
identify_cluster / identify_low_mem, identify_ranks, vote_strain_L2_batch, report files).  What the small scenarios
of tests/scenarios.py (trees of <= 11 nodes, clusters of <= 12 strains) cannot reach and these do:

  * get_ancestor_ab / get_uniq_path chains several levels deep (identify.py:139-164), many sibling groups pending at
    once (:231-372), reconstructed leaves AND internal nodes with overlapping_info deep in the tree;
  * layer-2 clusters of 40-60 strains and 200-300 k k-mers with 7+ strains present: Pre_Scan runs most of its 15
    iterations and hands more than six columns to ElasticNetCV (identify_strains_L2_Enet_Pscan_new_sp.py:302-371,
    437-456), so the per-fold-group statistics kernel and a 9..16-column coordinate descent are held to sklearn.

Everything here is a pure function of its seeds (numpy.random.RandomState only); the golden files pin a sha256 of
the generated inputs next to the reference's outputs.
"""
import os

import numpy as np

from . import synth

K = 31
C_MID = 53            # clusters = leaves; 2*53 - 1 = 105 tree nodes
N_STRAINS = 157


# ------------------------------------------------------------------------------------------------
# the 105-node tree
# ------------------------------------------------------------------------------------------------
def mid_tree(C=C_MID, seed=601):
    """-> parent map {id: parent or None}.  Numbering as Build_tree.py leaves it (what synth.build_l1_db asks for):
    leaves 1..C, root C+1, every internal node's id larger than its parent's.  Splits are skewed so that the tree is
    unbalanced (depth >= 6 asserted by the caller)."""
    rs = np.random.RandomState(seed)
    parent = {}
    leaf_ids = list(rs.permutation(C) + 1)
    nxt = [C + 1]
    # breadth-first: (node id, number of leaves below)
    queue = [(C + 1, C)]
    parent[C + 1] = None
    while queue:
        nid, n = queue.pop(0)
        a = int(np.clip(round(n * rs.beta(0.9, 0.9)), 1, n - 1))
        for m in (a, n - a):
            if m == 1:
                parent[int(leaf_ids.pop())] = nid
            else:
                nxt[0] += 1
                parent[nxt[0]] = nid
                queue.append((nxt[0], m))
    return parent


def _depth(parent, i):
    d = 0
    while parent[i] is not None:
        i = parent[i]
        d += 1
    return d


def mid_spec():
    """The database of the mid-size scenario: dict(parent, sites, singleton, clusters, reconstructed, overlaps, l2).

    sites: 1000-3000 k-mers per node and orientation for most nodes (label 2 / 'o2'), 500-1000 for a fifth (label 1 /
    'o1'), 300-480 for a tenth (weak under identify.py's 1000-row bar, label 1 under identify_low_mem.py's 500).
    Clusters: 28 singleton leaves (4th column of tree_structure.txt) and 25 multi-strain ones, 157 strains in all.
    l2: the clusters that own a Kmer_Sets_L2 directory (the ones the samples draw from), cid -> (S, seed)."""
    parent = mid_tree()
    T = synth.Tree(parent)
    assert len(T.ids) == 2 * C_MID - 1 and max(_depth(parent, l) for l in T.leaves) >= 6
    rs = np.random.RandomState(602)
    sites = {}
    for i in T.ids:
        u = rs.random_sample()
        if i == T.root or u < 0.70:
            sites[i] = int(rs.randint(1000, 3000))
        elif u < 0.90:
            sites[i] = int(rs.randint(500, 1000))
        else:
            sites[i] = int(rs.randint(300, 480))
    # cluster sizes: the layer-2 clusters first, then 20 more multi-strain ones, then singletons
    l2_sizes = {7: 12, 19: 9, 23: 6, 31: 4, 44: 3}                     # leaf id -> strains
    multi = dict(l2_sizes)
    rest = [l for l in sorted(T.leaves) if l not in multi]
    rs.shuffle(rest)
    for l in rest[:20]:
        multi[int(l)] = 0
    left = N_STRAINS - sum(l2_sizes.values()) - (C_MID - len(multi))   # strains for the 20 other multi clusters
    extra = [2] * 20
    left -= 40
    while left > 0:
        extra[int(rs.randint(0, 20))] += 1
        left -= 1
    for l, n in zip([l for l in multi if multi[l] == 0], extra):
        multi[l] = n
    clusters = {l: ["GCF_M%02d_%02d" % (l, j + 1) for j in range(n)] for l, n in multi.items()}
    singleton = {int(l): "GCF_S%02d" % l for l in T.leaves if l not in multi}
    assert sum(len(v) for v in clusters.values()) + len(singleton) == N_STRAINS
    # sample leaves need strong paths: every node on the path of a layer-2 cluster or of the sampled singletons
    # gets at least 1000 sites
    samp_single = sorted(singleton)[:3]
    for l in list(l2_sizes) + samp_single:
        for i in T.path(l):
            if sites[i] < 1000:
                sites[i] = int(rs.randint(1000, 2200))
    # one weak node and one label-1 node back on sampled paths, so that the deep chains meet them
    p7 = T.path(7)
    if len(p7) > 3:
        sites[p7[2]] = 420
    p19 = T.path(19)
    if len(p19) > 3:
        sites[p19[-2]] = 700
    # reconstructed nodes with overlapping_info: leaves and internal nodes, deep in the tree
    recon, overlaps = [], []
    inter = [i for i in T.ids if T.children[i] and i != T.root]
    cand = [23, 31, samp_single[1], p7[-2], p19[1] if len(p19) > 2 else p19[0], T.path(44)[-2]]
    for nj in cand:
        if nj in recon or nj == T.root:
            continue
        recon.append(int(nj))
    src = [7, 19, samp_single[0]]
    for t, nj in enumerate(recon):
        li = src[t % len(src)]
        if nj in T.path(li):                                       # a node cannot overlap a leaf below itself
            li = src[(t + 1) % len(src)]
            if nj in T.path(li):
                continue
        b = int(sites[nj] * (0.8 if t % 2 == 0 else 0.35))
        overlaps.append((int(li), int(nj), 0, b))
    del inter
    return dict(parent=parent, sites=sites, singleton=singleton, clusters=clusters, reconstructed=recon,
                overlaps=overlaps, l2={l: (n, 700 + l) for l, n in l2_sizes.items()}, sampled_singletons=samp_single)


def build_mid(root_dir, memory_db=False):
    """Write DB_M (or DB_Mmem: the same bytes plus the Memory_DB marker, StrainScan_build.py:129-130) with its five
    layer-2 cluster directories.  -> info dict of synth.build_l1_db + l2 infos + spec."""
    spec = mid_spec()
    db_dir = os.path.join(root_dir, "DB_Mmem" if memory_db else "DB_M")
    info = synth.build_l1_db(db_dir, spec["parent"], spec["sites"], 603, spec["singleton"], spec["clusters"],
                             spec["reconstructed"], spec["overlaps"])
    info["db_dir"] = db_dir
    info["spec"] = spec
    info["l2"] = {}
    for cid, (S, seed) in spec["l2"].items():
        rs = np.random.RandomState(seed)
        G = 2 * S + 6
        pres = np.zeros((S, G), bool)
        pres[:, 0] = True                                            # a core every strain carries
        for s in range(S):
            pres[s, 1 + s] = True                                    # a private segment per strain
        pres[:, 1 + S:] = rs.random_sample((S, G - 1 - S)) < 0.4     # shared segments
        seg = [2500] + [int(rs.randint(700, 1100)) for _ in range(S)] + [int(rs.randint(300, 700)) for _ in range(G - 1 - S)]
        other = [c for c in spec["l2"] if c != cid]
        shared_with = {1 + S: [other[0]], 2 + S: [other[1], other[2]]}
        info["l2"][cid] = synth.build_l2_cluster(db_dir, cid, C_MID, spec["clusters"][cid], seg, pres, seed=seed + 1,
                                                 shared_with=shared_with)
    if memory_db:
        open(os.path.join(db_dir, "Memory_DB"), "w").close()
    return info


def _strain_genome(info, leaf, j):
    name = info["spec"]["clusters"][leaf][j]
    return info["leaf_genome"][leaf] + b"N" + info["l2"][leaf]["strain_extra"][name]


# sample name -> ([(source, depth)], read seed).  source: ('strain', leaf, j) | ('leaf', leaf) | ('random', n)
def mid_samples(info):
    s0, s1, s2 = info["spec"]["sampled_singletons"]
    return {
        # four clusters, seven strains: two strains in C7, three in C19, one in C23, a singleton
        "M_mix": ([(("strain", 7, 0), 24.0), (("strain", 7, 5), 9.0), (("strain", 19, 1), 16.0), (("strain", 19, 4), 7.0),
                   (("strain", 19, 8), 4.0), (("strain", 23, 2), 11.0), (("leaf", s0), 8.0)], 611),
        # the same kind of mix below 1x: the default ladder's first rung finds nothing
        "M_low": ([(("strain", 7, 3), 0.7), (("strain", 31, 1), 0.5), (("leaf", s1), 0.4)], 612),
        # a little deeper: still nothing on the first rung, and layer 2 has k-mers seen twice to work with
        "M_low2": ([(("strain", 7, 3), 1.15), (("strain", 31, 1), 1.0), (("strain", 31, 2), 0.6), (("leaf", s1), 0.9)], 615),
        # one cluster only, three of its strains (the `cp` branch of vote_strain_L2_batch) + unrelated reads
        "M_one": ([(("strain", 44, 0), 20.0), (("strain", 44, 2), 8.0), (("random", 30000), 4.0)], 613),
        # reconstructed leaves and their overlap partners together
        "M_recon": ([(("strain", 23, 0), 14.0), (("strain", 31, 0), 10.0), (("strain", 31, 3), 5.0), (("strain", 7, 9), 12.0),
                     (("leaf", s1), 9.0), (("leaf", s2), 6.0)], 614),
    }


def mid_reads(info, sname):
    mix, seed = mid_samples(info)[sname]
    rs = np.random.RandomState(seed + 5000)
    gd = []
    for src, depth in mix:
        if src[0] == "strain":
            g = _strain_genome(info, src[1], src[2])
        elif src[0] == "leaf":
            g = info["leaf_genome"][src[1]]
        else:
            g = synth.rand_seq(rs, src[1])
        gd.append((g, depth))
    return synth.simulate_reads(gd, seed)


# whole-flow runs of the reference's command line: name -> (sample, database, extra argv)
MID_FLOW = {
    "mix_default": ("M_mix", "DB_M", []),
    "mix_b1": ("M_mix", "DB_M", ["-b", "1"]),
    "mix_e1": ("M_mix", "DB_M", ["-e", "1"]),
    "mix_l1": ("M_mix", "DB_M", ["-l", "1"]),
    "mix_lowmem": ("M_mix", "DB_Mmem", []),
    "low_default": ("M_low", "DB_M", []),                   # layer 2 dies in np.percentile([]) (identify_strains...:114)
    "low2_default": ("M_low2", "DB_M", []),                 # first rung empty -> [0.05, 0.05, 1], l2 = 1
    "low2_l2_b1": ("M_low2", "DB_M", ["-l", "2", "-b", "1"]),
    "low2_lowmem_l1": ("M_low2", "DB_Mmem", ["-l", "1"]),
    "one_default": ("M_one", "DB_M", []),
    "recon_default": ("M_recon", "DB_M", []),
    "recon_lowmem": ("M_recon", "DB_Mmem", []),
}


# ------------------------------------------------------------------------------------------------
# large layer-2 clusters
# ------------------------------------------------------------------------------------------------
L2_BIG = {
    # name: S, core sites, private sites (lo, hi), shared segments, shared sites (lo, hi), P(shared segment in strain),
    #       {strain: depth}, all_cls, l2, emode, outlier fraction
    # nine strains present, distinct depths: Pre_Scan keeps going until nothing unused is left
    "big9": dict(seed=41, S=48, core=14000, priv=(1100, 1700), n_sh=64, sh=(700, 1500), p_sh=0.22,
                 depths={3: 42.0, 7: 27.0, 11: 19.0, 17: 13.0, 22: 9.5, 29: 7.0, 35: 5.5, 41: 4.5, 46: 3.5},
                 all_cls=[2, 5], l2=0, emode=0, outl=0.0004),
    # -e: no coverage filter, 5000-k-mer bar, every candidate kept (identify_strains...:247-250, 350-355)
    "big_e": dict(seed=42, S=56, core=12000, priv=(3300, 3900), n_sh=40, sh=(600, 1200), p_sh=0.3,
                  depths={0: 44.0, 5: 36.0, 9: 30.0, 14: 25.0, 20: 21.0, 27: 18.0, 33: 15.5, 40: 13.0, 47: 11.5, 52: 10.0,
                          55: 9.0, 3: 3.0},
                  all_cls=[4], l2=0, emode=1, outl=0.0),
    # seventeen strains present: all 15 iterations run and ElasticNetCV gets 16 columns (max_iter, :302)
    "big16": dict(seed=44, S=40, core=9000, priv=(1200, 1600), n_sh=50, sh=(600, 1200), p_sh=0.2,
                  depths={0: 33.0, 2: 29.0, 4: 26.0, 7: 23.0, 9: 20.5, 12: 18.0, 14: 16.0, 17: 14.5, 19: 13.0, 22: 11.5, 24: 10.5,
                          27: 9.5, 29: 8.5, 32: 7.5, 34: 7.0, 37: 6.5, 39: 6.0},
                  all_cls=[3, 6], l2=0, emode=0, outl=0.0003),
    # the second rung of the ladder (l2 = 1): low depth, many k-mers at 0 / 2 / 3
    "big_l2": dict(seed=43, S=40, core=16000, priv=(1500, 2100), n_sh=56, sh=(800, 1400), p_sh=0.25,
                   depths={1: 6.0, 6: 4.4, 12: 3.6, 18: 3.0, 25: 2.6, 31: 2.2, 37: 1.9},
                   all_cls=[1, 3, 6], l2=1, emode=0, outl=0.0002),
}


def l2_big_case(name):
    """-> the same dict as scenarios.l2_case: X (K x S CSR int8), O (K x n_cls CSR int8), ids, y, detect_strains kwargs.
    K = 2 * (core + sum private + sum shared) rows, ~230-300 k."""
    import scipy.sparse as sp
    c = L2_BIG[name]
    rs = np.random.RandomState(c["seed"])
    S = c["S"]
    segs = [c["core"]] + [int(rs.randint(*c["priv"])) for _ in range(S)] + [int(rs.randint(*c["sh"])) for _ in range(c["n_sh"])]
    G = len(segs)
    pres = np.zeros((S, G), bool)
    pres[:, 0] = True
    pres[np.arange(S), 1 + np.arange(S)] = True
    pres[:, 1 + S:] = rs.random_sample((S, c["n_sh"])) < c["p_sh"]
    seg_of_row = np.repeat(np.arange(G), [2 * n for n in segs])
    Kn = seg_of_row.size
    Xd = pres[:, seg_of_row].T.astype(np.int8)                       # K x S
    n_cls = 7
    O = np.zeros((Kn, n_cls), np.int8)
    O[:, c["all_cls"][0] - 1] = 1
    for oc in c["all_cls"][1:]:
        O[rs.random_sample(Kn) < 0.08, oc - 1] = 1
    O[rs.random_sample(Kn) < 0.05, 6] = 1                            # shared with a cluster that was NOT identified
    dep = np.zeros(S)
    for s, d in c["depths"].items():
        dep[s] = d
    lam = Xd.astype(np.float64) @ (dep * 0.4)
    y = rs.poisson(lam).astype(np.int64)
    y[(lam == 0) & (rs.random_sample(Kn) < 0.003)] = 2               # a little noise where nothing is present
    if c["outl"]:
        y[rs.random_sample(Kn) < c["outl"]] += 20000
    y[y == 1] = 0                                                    # remove_1 (Vote_...:312-322)
    ids = ["GCF_%s_%02d" % (name.upper(), i + 1) for i in range(S)]
    nz = y[y != 0]
    npp_out = float(np.median(nz) * 1000)
    return dict(X=sp.csr_matrix(Xd), O=sp.csr_matrix(O), ids=ids, y=y, ksize=31, npp25=0, npp75=npp_out,
                npp_out=npp_out, cls_cov=0.9, all_cls=c["all_cls"], l2=c["l2"], msn=40, pmode=0, emode=c["emode"])

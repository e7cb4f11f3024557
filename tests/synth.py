"""Seeded synthetic StrainScan databases and read sets in the reference's on-disk formats.

Shared by tests/golden/make_golden.py (which feeds them to the real reference in the build
container) and by the tests (which feed the very same bytes to the oracle and to the HIP path).
Only numpy.random.RandomState (frozen MT19937 stream) is used, so the bytes are identical
under every numpy version; golden files additionally pin a sha256 of what was generated.

On-disk formats follow SURVEY.md 8(f):
  Tree_database/  tree_structure.txt node_length.txt reconstructed_nodes.txt kmer.fa kmers/<id>
                  overlapping_info/<leaf>[_supple] hclsMap_95_recls.txt
                  (writer in the reference: library/Build_tree.py:494-526,648-698)
  Kmer_Sets_L2/Kmer_Sets/C<id>/  all_kmer.fasta all_kid.pkl all_strains_re.npz id2strain_re.pkl
                  overlap_matrix.npz
                  (library/Build_kmer_sets_unique_region_lasso_test_allinone_sp.py:397-410,
                   library/Recls_withR_new.py:110-115, library/Build_overlap_matrix_sp.py:89-98)
"""
import hashlib
import os
import pickle

import numpy as np

_COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
K = 31


def rand_seq(rs, n):
    return bytes(np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, size=n)])


def revcomp(s):
    return s.translate(_COMP)[::-1]


def sha256_of(*blobs):
    h = hashlib.sha256()
    for b in blobs:
        h.update(b if isinstance(b, (bytes, bytearray)) else bytes(b))
    return h.hexdigest()


class Tree:
    """parent map -> ids, children (ascending id), leaves, paths."""

    def __init__(self, parent):
        self.parent = dict(parent)
        self.ids = sorted(parent)
        self.children = {i: [] for i in self.ids}
        for i in self.ids:
            if parent[i] is not None:
                self.children[parent[i]].append(i)
        self.root = [i for i in self.ids if parent[i] is None][0]
        self.leaves = [i for i in self.ids if not self.children[i]]

    def path(self, leaf):
        p = []
        while leaf is not None:
            p.append(leaf)
            leaf = self.parent[leaf]
        return p[::-1]


def treelib_pickle(parent, protocol=None):
    """Bytes of `pickle.dump(tree)` for a treelib 1.6.1 Tree with the given parent map, written WITHOUT treelib
    (absent from the image): classes named treelib.tree.Tree / treelib.node.Node carrying 1.6.1's attribute
    layout (tests/golden/_standin/treelib mirrors the same layout, so the reference run under the stand-in
    reads these bytes too).  Build_tree.py:283-329 pickles a one-node tree for a single-cluster database."""
    import sys
    import types
    from collections import defaultdict
    mods = {n: types.ModuleType(n) for n in ("treelib", "treelib.tree", "treelib.node")}
    Node = type("Node", (object,), {"__module__": "treelib.node"})
    TreeC = type("Tree", (object,), {"__module__": "treelib.tree"})
    mods["treelib.node"].Node = Node
    mods["treelib.tree"].Tree = TreeC
    tid = "8a4d5e6e-5f7d-11ee-b3a1-0242ac110002"          # uuid1 string in treelib; any hashable works
    ids = sorted(parent, key=lambda i: (parent[i] is not None, i))
    nodes = {}
    for i in ids:
        n = Node()
        n.__dict__.update(_identifier=i, _tag=i, expanded=True, _predecessor={tid: parent[i]},
                          _successors=defaultdict(list), data=None, _initial_tree_id=tid)
        nodes[i] = n
    for i in ids:
        if parent[i] is not None:
            nodes[parent[i]]._successors[tid].append(i)
    t = TreeC()
    t.__dict__.update(_identifier=tid, _nodes=nodes, root=[i for i in ids if parent[i] is None][0])
    saved = {n: sys.modules.get(n) for n in mods}
    sys.modules.update(mods)
    try:
        return pickle.dumps(t, protocol) if protocol is not None else pickle.dumps(t)
    finally:
        for n, m in saved.items():
            if m is None:
                del sys.modules[n]
            else:
                sys.modules[n] = m


def build_l1_db(db_dir, parent, sites, seed, singleton=None, clusters=None, reconstructed=(),
                overlaps=(), extra_rows=(), single_cluster=False, invalid_nodes=()):
    """Write <db_dir>/Tree_database.  Returns dict(tree, node_seq, leaf_genome, row_of_node).

    parent        {id: parent or None}; leaves must be 1..C, the root C+1, internal ids so that
                  every parent id is smaller than its internal children (Build_tree.py numbering)
    sites         {id: n}: node owns a private random sequence of n+30 bases = n forward k-mers,
                  stored with their reverse complements as adjacent rows of kmer.fa (2n rows)
    singleton     {leaf: strain name} (4th column of tree_structure.txt)
    clusters      {leaf: [strain names]} for hclsMap_95_recls.txt
    reconstructed iterable of node ids
    overlaps      [(leaf_i, node_j, a, b)]: sites [a,b) of node_j also occur in leaf_i's genome
    extra_rows    [(position, text)] raw rows spliced into kmer.fa (not listed in any node)
    invalid_nodes node ids whose every kmer.fa row carries an 'N': jellyfish never dumps them, so the node's valid
                  length is 0 while node_length.txt still counts the rows (identify.py:298-303)
    """
    rs = np.random.RandomState(seed)
    singleton = singleton or {}
    clusters = clusters or {}
    T = Tree(parent)
    tdir = os.path.join(db_dir, "Tree_database")
    os.makedirs(os.path.join(tdir, "kmers"), exist_ok=True)
    os.makedirs(os.path.join(tdir, "overlapping_info"), exist_ok=True)
    node_seq = {i: rand_seq(rs, sites[i] + K - 1) for i in T.ids}
    rows = []
    row_of_node = {}
    for i in T.ids:
        s = node_seq[i]
        idx = []
        for p in range(sites[i]):
            km = s[p:p + K]
            if i in invalid_nodes:
                km = km[:15] + b"N" + km[16:]
            idx.append(len(rows)); rows.append(km)
            idx.append(len(rows)); rows.append(revcomp(km))
        row_of_node[i] = idx
    # splice extra rows (N rows, lower-case rows, duplicates) without disturbing node lists
    extra_rows = sorted(extra_rows)
    if extra_rows:
        shift = np.zeros(len(rows) + 1, np.int64)
        out = []
        e = 0
        for r, km in enumerate(rows):
            while e < len(extra_rows) and extra_rows[e][0] <= r:
                out.append(extra_rows[e][1]); e += 1
            shift[r] = len(out) - r
            out.append(km)
        while e < len(extra_rows):
            out.append(extra_rows[e][1]); e += 1
        for i in row_of_node:
            row_of_node[i] = [r + int(shift[r]) for r in row_of_node[i]]
        rows = out
    with open(os.path.join(tdir, "kmer.fa"), "wb") as f:
        f.write(b"".join(b">1\n" + km + b"\n" for km in rows))
    with open(os.path.join(tdir, "tree_structure.txt"), "w") as f:
        if single_cluster:                              # Build_tree.py:329-334: "<id>\t", no newline, + tree.pkl
            assert len(T.ids) == 1
            f.write("%d\t" % T.ids[0])
            with open(os.path.join(tdir, "tree.pkl"), "wb") as fp:
                fp.write(treelib_pickle(parent))
        else:
            for i in T.ids:
                par = "N" if parent[i] is None else str(parent[i])
                ch = "N" if not T.children[i] else " ".join(map(str, T.children[i]))
                f.write("%d\t%s\t%s\t%s\n" % (i, par, ch, singleton.get(i, "")))
    with open(os.path.join(tdir, "node_length.txt"), "w") as f:
        for i in T.ids:
            f.write("%d\t%d\n" % (i, len(row_of_node[i])))
    with open(os.path.join(tdir, "reconstructed_nodes.txt"), "w") as f:
        for i in reconstructed:
            f.write("%d\n" % i)
    for i in T.ids:
        with open(os.path.join(tdir, "kmers", str(i)), "w") as f:
            f.write("".join("%d " % r for r in row_of_node[i]))
    with open(os.path.join(tdir, "hclsMap_95_recls.txt"), "w") as f:
        for leaf in T.leaves:
            names = clusters.get(leaf) or [singleton.get(leaf, "S%d" % leaf)]
            f.write("%d\t%d\t%s\n" % (leaf, len(names), ",".join(names)))
    by_leaf = {}
    for (li, nj, a, b) in overlaps:
        by_leaf.setdefault(li, []).append((nj, a, b))
    for li, lst in by_leaf.items():
        with open(os.path.join(tdir, "overlapping_info", str(li)), "w") as f, \
                open(os.path.join(tdir, "overlapping_info", "%d_supple" % li), "w") as f1:
            count = -1
            for (nj, a, b) in lst:
                f.write("%d\n" % nj)
                f.write("".join("%d " % p for p in range(2 * a, 2 * b)))
                f.write("\n")
                count += 2
                f1.write("%d %d\n" % (nj, count))
    leaf_genome = {}
    for leaf in T.leaves:
        g = b"".join(node_seq[i] for i in T.path(leaf))
        for (nj, a, b) in by_leaf.get(leaf, []):
            g += node_seq[nj][a:b + K - 1]
        leaf_genome[leaf] = g
    return dict(tree=T, node_seq=node_seq, leaf_genome=leaf_genome, row_of_node=row_of_node,
                n_rows=len(rows))


def build_l2_cluster(db_dir, cid, n_clusters, strains, seg_sites, presence, seed,
                     shared_with=None, k=None):
    """Write <db_dir>/Kmer_Sets_L2/Kmer_Sets/C<cid>.  Returns dict(strain_extra, K, X).

    strains    list of strain names (columns after re-clustering)
    seg_sites  [n_g]: segment g owns n_g sites (forward + revcomp rows -> 2*n_g k-mers)
    presence   bool [S, G]: strain s carries segment g
    shared_with {g: [other cluster ids]} k-mers of segment g also belong to those clusters
               (extra 1s in overlap_matrix.npz)
    """
    import scipy.sparse as sp
    rs = np.random.RandomState(seed)
    cdir = os.path.join(db_dir, "Kmer_Sets_L2", "Kmer_Sets", "C%d" % cid)
    os.makedirs(cdir, exist_ok=True)
    presence = np.asarray(presence, bool)
    S, G = presence.shape
    K = globals()["K"] if k is None else int(k)       # (the k-mer sets of a database built with StrainScan_build.py -k)
    seg_seq = [rand_seq(rs, n + K - 1) for n in seg_sites]
    kid = {}
    rows_i, rows_j, orow, ocol = [], [], [], []
    for g, s in enumerate(seg_seq):
        for p in range(seg_sites[g]):
            for km in (s[p:p + K], revcomp(s[p:p + K])):
                km = km.decode()
                if km in kid:
                    continue
                kid[km] = len(kid) + 1
                r = kid[km] - 1
                for si in np.nonzero(presence[:, g])[0]:
                    rows_i.append(r); rows_j.append(int(si))
                orow.append(r); ocol.append(cid - 1)
                for oc in (shared_with or {}).get(g, []):
                    orow.append(r); ocol.append(oc - 1)
    Kn = len(kid)
    X = sp.csr_matrix((np.ones(len(rows_i), np.int8), (rows_i, rows_j)), shape=(Kn, S), dtype=np.int8)
    O = sp.csr_matrix((np.ones(len(orow), np.int8), (orow, ocol)), shape=(Kn, n_clusters), dtype=np.int8)
    sp.save_npz(os.path.join(cdir, "all_strains_re.npz"), X)
    sp.save_npz(os.path.join(cdir, "overlap_matrix.npz"), O)
    with open(os.path.join(cdir, "all_kid.pkl"), "wb") as f:
        pickle.dump(kid, f, 2)
    with open(os.path.join(cdir, "id2strain_re.pkl"), "wb") as f:
        pickle.dump(list(strains), f, 2)
    with open(os.path.join(cdir, "all_kmer.fasta"), "w") as f:
        for c, km in enumerate(kid, 1):
            f.write(">%d\n%s\n" % (c, km))
    strain_extra = {strains[si]: b"".join(seg_seq[g] + b"N" for g in range(G) if presence[si, g])
                    for si in range(S)}
    return dict(strain_extra=strain_extra, K=Kn, X=X, O=O, kid=kid)


def simulate_reads(genomes_depths, seed, read_len=150, err=0.005, n_frac=0.002, lower_frac=0.01,
                   fasta=False):
    """[(genome bytes, depth)] -> FASTQ (or FASTA) text.  Strand 50/50, substitution errors,
    a few reads carry an 'N', a few are lower-case, quality is 'I' (SURVEY 8d)."""
    rs = np.random.RandomState(seed)
    recs = []
    rid = 0
    lut = np.frombuffer(b"ACGT", np.uint8)
    for g, depth in genomes_depths:
        L = len(g)
        n = int(depth * L / read_len)
        ga = np.frombuffer(g, np.uint8)
        starts = rs.randint(0, max(1, L - read_len + 1), size=n)
        strand = rs.randint(0, 2, size=n)
        for s, st in zip(starts, strand):
            r = ga[s:s + read_len].copy()
            m = rs.random_sample(r.size) < err
            if m.any():
                r[m] = lut[rs.randint(0, 4, size=int(m.sum()))]
            if rs.random_sample() < n_frac:
                r[rs.randint(0, r.size)] = ord("N")
            b = r.tobytes()
            if st:
                b = revcomp(b)
            if rs.random_sample() < lower_frac:
                b = b.lower()
            rid += 1
            if fasta:
                recs.append(b">r%d\n%s\n" % (rid, b))
            else:
                recs.append(b"@r%d\n%s\n+\n%s\n" % (rid, b, b"I" * len(b)))
    order = rs.permutation(len(recs))
    return b"".join(recs[i] for i in order)


def flat_bases_from_fastx(text):
    """Sequence lines of a FASTA/FASTQ text, records joined by '\\n' (the device block format).
    Pure-python reference parser for tests (4-line FASTQ and multi-line FASTA/FASTQ)."""
    out = []
    lines = text.split(b"\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith(b">"):
            i += 1
            seq = []
            while i < len(lines) and not lines[i].startswith(b">"):
                seq.append(lines[i]); i += 1
            out.append(b"".join(seq))
        elif ln.startswith(b"@"):
            i += 1
            seq = []
            while i < len(lines) and not lines[i].startswith(b"+"):
                seq.append(lines[i]); i += 1
            s = b"".join(seq)
            out.append(s)
            i += 1
            q = 0
            while i < len(lines) and q < len(s):
                q += len(lines[i]); i += 1
        else:
            i += 1
    return b"\n".join(out) + b"\n"

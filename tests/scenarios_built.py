"""A Tree_database written by the REFERENCE'S OWN BUILDER (round 6).

Every other database of the suite is written by tests/synth.py -- this package's reading of library/Build_tree.py:494-526,
648-698.  Here the reference's `Build_tree.build_tree` itself runs (tests/golden/make_golden.py, in the build container) on a
seeded set of synthetic genomes: its hierarchy(), its node-unique k-mer sets, its random down-sampling of sets above the cap
(:590-591), its reconstruction of nodes below the floor with `overlapping_info` (:598-661), its file writers.  The builder
cannot travel (it is the reference's source, and it needs Bio / bidict / treelib, absent from the image: minimal stand-ins under
tests/golden/_standin/, used for that run only), so its OUTPUT is committed as a fixture -- tests/golden/built_tree_db.tar.gz --
next to what the reference's identify modules then found in reads of those genomes (built_l1.json).

What is NOT covered: the front of StrainScan_build.py (dashing's Jaccard matrix, R's hclust, select_rep) -- the similarity
matrix and the 95 % clusters are made here -- and the layer-2 k-mer set builder (sibeliaz)."""
import io
import os
import tarfile

import numpy as np

from . import synth

K = 31
# the phylogeny the genomes are drawn from: cluster (leaf) -> path of segment names root..leaf; strains per cluster
BUILT_PARENT = {"r": None, "a": "r", "b": "r", "a1": "a", "a2": "a", "b1": "b", "b2": "b", "a1x": "a1", "a1y": "a1", "b2x": "b2", "b2y": "b2",
                "a2x": "a2", "a2y": "a2"}
BUILT_LEAVES = {"a1x": 3, "a1y": 1, "a2x": 2, "a2y": 1, "b1": 2, "b2x": 1, "b2y": 3}        # leaf segment -> strains
BUILT_PARAMS = [0.8, 1000, 6000, 3000]        # alpha_ratio, minsize, maxsize (sets above it are down-sampled), max clusters for reconstruction


def built_genomes():
    """-> ({strain name: genome bytes}, {strain name: cluster id 1..7}, segment table).  A strain's genome = the segments on its
    leaf's path + a private tail; strains of one cluster differ by 0.3 % substitutions in the leaf segment (so that the cluster's
    core -- k-mers in >= 80 % of its genomes, Build_tree.py:124-129 -- is smaller than any one genome)."""
    rs = np.random.RandomState(901)
    seg = {}
    for name in BUILT_PARENT:
        n = int(rs.randint(1800, 5200))
        if name in ("a1y", "b2x"):
            n = 700                                   # two short leaf segments
        if name in ("a2", "b"):
            n = 350                                   # two internal nodes below the builder's floor of 1000 k-mers: reconstructed (:598-661)
        seg[name] = synth.rand_seq(rs, n)
    genomes, cluster_of = {}, {}
    lut = np.frombuffer(b"ACGT", np.uint8)
    for cid, (leaf, n_strains) in enumerate(BUILT_LEAVES.items(), 1):
        path = []
        x = leaf
        while x is not None:
            path.append(x)
            x = BUILT_PARENT[x]
        base = b"N".join(seg[s] for s in path[::-1][:-1])
        for s in range(n_strains):
            lf = np.frombuffer(seg[leaf], np.uint8).copy()
            if n_strains > 1:
                m = rs.random_sample(lf.size) < 0.003
                lf[m] = lut[rs.randint(0, 4, int(m.sum()))]
            name = "GCF_B%d_%d" % (cid, s + 1)
            genomes[name] = base + b"N" + lf.tobytes() + b"N" + synth.rand_seq(rs, 600)
            cluster_of[name] = cid
    return genomes, cluster_of, seg


def write_builder_inputs(root):
    """FASTA files, distance_matrix.txt (similarities, as Cluster.construct_matrix leaves them: Build_tree.py:256-270 reads the paths
    from the header line) and the 95 % cluster map (Build_tree.py:273-280: id, size, comma-separated names).  -> (matrix, map)."""
    genomes, cluster_of, _ = built_genomes()
    names = sorted(genomes)
    gdir = os.path.join(root, "genomes")
    os.makedirs(gdir, exist_ok=True)
    paths = {}
    for n in names:
        paths[n] = os.path.join(gdir, n + ".fna")
        with open(paths[n], "wb") as f:
            g = genomes[n]
            f.write(b">" + n.encode() + b" synthetic\n" + b"\n".join(g[i:i + 70] for i in range(0, len(g), 70)) + b"\n")
    # similarity = shared 31-mers / all 31-mers of the pair (what dashing estimates)
    km = {n: set(genomes[n][i:i + K] for i in range(len(genomes[n]) - K + 1) if b"N" not in genomes[n][i:i + K]) for n in names}
    mpath = os.path.join(root, "distance_matrix.txt")
    with open(mpath, "w") as f:
        f.write("\t" + "\t".join(paths[n] for n in names) + "\n")
        for a in names:
            f.write(a + "\t" + "\t".join("%.6f" % (1.0 if a == b else len(km[a] & km[b]) / len(km[a] | km[b])) for b in names) + "\n")
    cpath = os.path.join(root, "hclsMap_95_recls.txt")
    with open(cpath, "w") as f:
        for cid in sorted(set(cluster_of.values())):
            mem = [n for n in names if cluster_of[n] == cid]
            f.write("%d\t%d\t%s\n" % (cid, len(mem), ",".join(mem)))
    return mpath, cpath


# samples of the built database: name -> ([(strain name or ('random', n), depth)], seed)
BUILT_SAMPLES = {
    "T_mix": ([("GCF_B1_2", 14.0), ("GCF_B5_1", 8.0), ("GCF_B7_3", 5.0)], 921),
    "T_single": ([("GCF_B2_1", 12.0), ("GCF_B6_1", 6.0)], 922),
    "T_low": ([("GCF_B3_1", 0.8), ("GCF_B7_1", 0.5)], 923),
    "T_none": ([(("random", 20000), 6.0)], 924),
}


def built_reads(sname):
    genomes, _, _ = built_genomes()
    mix, seed = BUILT_SAMPLES[sname]
    rs = np.random.RandomState(seed + 7000)
    return synth.simulate_reads([(genomes[s] if isinstance(s, str) else synth.rand_seq(rs, s[1]), d) for s, d in mix], seed)


def pack_tree_database(tdir):
    """-> bytes of a .tar.gz of the directory, the same bytes for the same files (sorted names, no times, no owners)."""
    import gzip
    raw = io.BytesIO()
    with tarfile.open(fileobj=raw, mode="w", format=tarfile.USTAR_FORMAT) as tf:
        for dirpath, dirs, files in sorted(os.walk(tdir)):
            dirs.sort()
            for fn in sorted(files):
                p = os.path.join(dirpath, fn)
                ti = tarfile.TarInfo(os.path.join("Tree_database", os.path.relpath(p, tdir)))
                ti.size = os.path.getsize(p)
                ti.mtime = 0
                ti.mode = 0o644
                with open(p, "rb") as f:
                    tf.addfile(ti, f)
    out = io.BytesIO()
    with gzip.GzipFile(fileobj=out, mode="wb", mtime=0, compresslevel=9) as gz:
        gz.write(raw.getvalue())
    return out.getvalue()


def unpack_tree_database(blob_path, root):
    """The committed fixture -> <root>/DB_T/Tree_database; -> that directory."""
    dst = os.path.join(root, "DB_T")
    os.makedirs(dst, exist_ok=True)
    with tarfile.open(blob_path, "r:gz") as tf:
        for m in tf.getmembers():
            if not m.isfile() or m.name.startswith("/") or ".." in m.name.split("/"):
                raise ValueError("unexpected member %r in %s" % (m.name, blob_path))
        tf.extractall(dst)
    return os.path.join(dst, "Tree_database")

"""Seeded RANDOM scenarios for a campaign of the real reference against the oracle / the product (round 6).

tests/scenarios.py and tests/scenarios_mid.py are hand-made: each case aims at branches somebody thought of.  The
cases here are drawn from a seed -- tree shape, node sizes (strong / label-1 / weak / never dumped), reconstructed nodes
and their overlaps, sample mixes from 0.2x to 40x with partial genomes and foreign reads; layer-2 clusters of 2..24
strains with random presence patterns, depths, `-e`, low-depth mode, a small strain cap, outliers -- so that what
nobody thought of gets a chance too.  tests/golden/fuzz_reference.py runs the reference on seeds [a, b) here in the
build container and checks the oracle and the product's host logic against it (hundreds of seeds, nothing committed);
the seeds listed in FUZZ_L1_KEPT / FUZZ_L2_KEPT / FUZZ_FLOW_KEPT are committed as goldens (fuzz_l1.json, fuzz_l2.json,
fuzz_flow.json) and held by tests/test_fuzz_golden.py (CPU: oracle and host logic; -m gpu: the HIP path).  The third kind
(flow_*) is the reference's whole command line on a random database WITH layer-2 k-mer sets: flags (-b, -e, -l, -s, -k),
paired and gzip-compressed input, Memory_DB, clusters whose k-mer set is missing.

Everything is a pure function of the seed (numpy.random.RandomState only)."""
import os

import numpy as np

from . import synth

K = 31
FUZZ_L1_KEPT = [5, 28, 36, 42, 43, 50, 63, 73, 81, 88, 92, 125]
FUZZ_L2_KEPT = [2, 3, 11, 15, 28, 39, 62, 112, 147, 190, 377, 418, 420, 555, 676, 692, 10081, 10112, 20003, 20005, 20006, 20012, 20019]      # (20003...: no k-mer seen, all counts <= 2, one k-mer seen [IndexError / a result], a handful)
FUZZ_FLOW_KEPT = [0, 2, 28, 36, 63, 66, 1000, 1019, 1027, 1042, 1057, 1109, 1146, 1239, 2002, 2015, 2010, 2020, 2031, 2007, 3029, 3038]      # (2002...: an empty file, [2015: and the IndexError it ends in,] reads shorter than k, a single read, nothing but N, a handful of reads; 3029, 3038: a single-cluster database with its tree.pkl)
FUZZ_L1X_KEPT = [0, 2, 6, 23, 92]                                    # (kmer.fa with rows no node lists: build_l1x)
FUZZ_FMT_KEPT = [0, 4, 5, 8, 9, 13, 14, 16, 20, 21, 23, 29, 31, 33, 36, 38, 45, 51, 66, 84, 89, 92] + [2, 3, 149]      # (the last three: one per known deviation)


# ------------------------------------------------------------------------------------------------
# layer 1: a random cluster search tree and samples
# ------------------------------------------------------------------------------------------------
def l1_tree(C, rs):
    """Random binary tree with Build_tree.py's numbering (leaves 1..C, root C + 1, children's ids above their parent's)."""
    parent = {C + 1: None}
    leaf_ids = list(rs.permutation(C) + 1)
    nxt = C + 1
    queue = [(C + 1, C)]
    skew = float(rs.choice([0.5, 0.9, 3.0]))
    while queue:
        nid, n = queue.pop(0)
        a = int(np.clip(round(n * rs.beta(skew, skew)), 1, n - 1))
        for m in (a, n - a):
            if m == 1:
                parent[int(leaf_ids.pop())] = nid
            else:
                nxt += 1
                parent[nxt] = nid
                queue.append((nxt, m))
    return parent


def l1_spec(seed):
    rs = np.random.RandomState(100000 + seed)
    C = int(rs.choice([2, 3, 4, 6, 9, 14, 22, 30]))
    parent = l1_tree(C, rs)
    T = synth.Tree(parent)
    sites = {}
    p_small = float(rs.choice([0.0, 0.15, 0.4]))
    for i in T.ids:
        u = rs.random_sample()
        if u >= p_small:
            sites[i] = int(rs.randint(1000, 1800))           # label 2
        elif u >= p_small * 0.45:
            sites[i] = int(rs.randint(500, 1000))            # label 1 under identify.py (1000-row bar counts both strands)
        else:
            sites[i] = int(rs.randint(120, 499))             # weak
    multi_n = int(rs.randint(0, C + 1))
    multi = [int(x) for x in rs.permutation(T.leaves)[:multi_n]]
    clusters = {l: ["GCF_F%d_%02d_%02d" % (seed, l, j + 1) for j in range(int(rs.randint(2, 6)))] for l in multi}
    singleton = {int(l): "GCF_F%d_S%02d" % (seed, l) for l in T.leaves if l not in clusters}
    inner = [i for i in T.ids if i != T.root]
    recon, overlaps = [], []
    if len(inner) >= 3 and rs.random_sample() < 0.7:
        for nj in rs.permutation(inner)[:int(rs.randint(1, min(5, len(inner)) + 1))]:
            nj = int(nj)
            recon.append(nj)
            cand = [int(l) for l in T.leaves if nj not in T.path(l)]
            if cand and rs.random_sample() < 0.8:
                li = int(rs.choice(cand))
                a = int(rs.randint(0, sites[nj] // 3))
                b = int(rs.randint(a + 1, sites[nj] + 1))
                overlaps.append((li, nj, a, b))
    invalid = []
    if len(inner) >= 4 and rs.random_sample() < 0.15:
        invalid = [int(rs.choice(inner))]
    return dict(parent=parent, sites=sites, singleton=singleton, clusters=clusters, reconstructed=recon, overlaps=overlaps,
                invalid_nodes=invalid, db_seed=200000 + seed)


def build_l1(seed, root_dir):
    spec = l1_spec(seed)
    db_dir = os.path.join(root_dir, "DB_F%d" % seed)
    info = synth.build_l1_db(db_dir, spec["parent"], spec["sites"], spec["db_seed"], spec["singleton"], spec["clusters"],
                             spec["reconstructed"], spec["overlaps"], invalid_nodes=spec["invalid_nodes"])
    info["db_dir"] = db_dir
    info["spec"] = spec
    return info


def l1_reads(info, seed, which):
    """Sample `which` (0 or 1) of database `seed`: FASTQ bytes."""
    rs = np.random.RandomState(300000 + 10 * seed + which)
    T = info["tree"]
    n_src = int(rs.choice([1, 1, 2, 3, 5]))
    gd = []
    for l in rs.permutation(T.leaves)[:n_src]:
        g = info["leaf_genome"][int(l)]
        depth = float(np.exp(rs.uniform(np.log(0.2), np.log(40.0))))
        how = rs.random_sample()
        if how < 0.15:                                               # a fraction of every node on the path
            a, b = sorted(rs.uniform(0, 1, size=2))
            if b - a < 0.2:
                a, b = 0.0, 0.6
            g = b"N".join(info["node_seq"][i][int(a * len(info["node_seq"][i])):int(b * len(info["node_seq"][i]))] for i in T.path(int(l)))
        elif how < 0.25:                                             # the path without its leaf: a relative the tree does not hold
            g = b"".join(info["node_seq"][i] for i in T.path(int(l))[:-1]) or g
        gd.append((g, depth))
    if rs.random_sample() < 0.5:
        gd.append((synth.rand_seq(rs, int(rs.randint(2000, 30000))), float(rs.uniform(1, 10))))
    return synth.simulate_reads(gd, 400000 + 10 * seed + which, err=float(rs.choice([0.0, 0.005, 0.02])))


def l1_runs(seed):
    """Which (module, cutoff) pairs the reference is run with on both samples of this seed."""
    from . import scenarios as sc
    rs = np.random.RandomState(500000 + seed)
    cuts = [sc.CUTOFFS[int(i)] for i in rs.permutation(4)[:2]]
    return [(m, c) for c in cuts for m in ("identify", "identify_low_mem")]


# ------------------------------------------------------------------------------------------------
# layer 2: a random cluster for detect_strains
# ------------------------------------------------------------------------------------------------
def l2_case(seed):
    """-> the dict of tests/scenarios.py l2_case (X, O csr; ids; y; the keyword values of detect_strains)."""
    import scipy.sparse as sp
    rs = np.random.RandomState(600000 + seed)
    big = seed >= 10000                                              # seeds from 10000 on: 30-56 strains, 6-14 of them present (p > 6: the
    S = int(rs.choice([30, 40, 56])) if big else int(rs.choice([2, 3, 4, 6, 9, 13, 18, 24]))      # per-fold-group statistics and wide refits)
    G = S + int(rs.randint(1, (S // 2 if big else 2 * S) + 3))
    dens = float(rs.choice([0.2, 0.4, 0.6]))
    pres = rs.random_sample((S, G)) < dens
    if rs.random_sample() < 0.7:
        pres[:, 0] = True                                            # a core
    for s in range(S):                                               # most strains own a private segment
        if rs.random_sample() < 0.8 and 1 + s < G:
            pres[:, 1 + s] = False
            pres[s, 1 + s] = True
    if rs.random_sample() < 0.2 and S >= 3:                          # two strains with the same column pattern
        pres[1] = pres[0]
    segs = [int(rs.randint(80, 900)) for _ in range(G)]
    n_cls = int(rs.randint(2, 7))
    all_cls = [int(x) for x in (rs.permutation(n_cls)[:int(rs.choice([1, 1, 2, 3]))] + 1)][:n_cls]
    l2 = int(rs.random_sample() < 0.3)
    emode = int(rs.random_sample() < 0.3)
    msn = int(rs.choice([40, 40, 40, 3, 1]))
    n_pres = int(rs.randint(6, 15)) if big else int(rs.randint(1 if rs.random_sample() < 0.25 else 2, min(S, 8) + 1))
    depths = np.zeros(S)
    for s in rs.permutation(S)[:n_pres]:
        depths[s] = float(np.exp(rs.uniform(np.log(2.0), np.log(9.0)))) if l2 else float(np.exp(rs.uniform(np.log(9.0), np.log(90.0))))
    seg_of_row = np.repeat(np.arange(G), [2 * n for n in segs])
    Kn = seg_of_row.size
    Xd = pres[:, seg_of_row].T.astype(np.int8)
    O = np.zeros((Kn, n_cls), np.int8)
    O[:, all_cls[0] - 1] = 1
    for c in all_cls[1:]:
        O[rs.random_sample(Kn) < float(rs.choice([0.05, 0.3])), c - 1] = 1
    lam = Xd.astype(np.float64) @ (depths * 0.4)
    y = rs.poisson(lam).astype(np.int64)
    if rs.random_sample() < 0.3:
        y[rs.random_sample(Kn) < 0.003] += int(rs.randint(200, 5000))
    if rs.random_sample() < 0.3:                                     # k-mers seen in the sample that no present strain explains
        m = rs.random_sample(Kn) < 0.02
        y[m] += rs.poisson(3.0, size=int(m.sum()))
    if 20000 <= seed < 30000:                                        # seeds 20000..: counts at the edge of being a sample
        u = np.random.RandomState(620000 + seed).random_sample()
        if u < 0.2:
            y[:] = 0                                                 # no read hit the cluster at all
        elif u < 0.4:
            keep1 = np.random.RandomState(630000 + seed).randint(0, Kn)
            y[np.arange(Kn) != keep1] = 0                            # one k-mer seen
        elif u < 0.6:
            y[np.random.RandomState(640000 + seed).random_sample(Kn) < 0.995] = 0      # a handful of k-mers seen
        elif u < 0.8:
            y[:] = np.minimum(y, 2)                                  # everything seen at most twice
    y[y == 1] = 0                                                    # remove_1 (Vote_Strain_L2_Lasso_new_sp.py:312-322)
    ids = ["GCF_Z%d_%d" % (seed, i + 1) for i in range(S)]
    nz = y[y != 0]
    npp25, npp_out = 0.0, float(np.median(nz) * 1000) if nz.size else 0.0      # what the caller passes (Vote_Strain_L2_Lasso_new_sp.py:401-409)
    npp75 = npp_out
    if nz.size and rs.random_sample() < 0.25:                        # a narrower window: the signature allows it
        npp25, npp75, npp_out = float(np.percentile(nz, 10)), float(np.percentile(nz, 95)), float(np.percentile(nz, 99))
    cls_cov = float(rs.choice([0.9, 0.5, 0.05]))
    return dict(X=sp.csr_matrix(Xd), O=sp.csr_matrix(O), ids=ids, y=y, ksize=31, npp25=npp25, npp75=npp75, npp_out=npp_out,
                cls_cov=cls_cov, all_cls=all_cls, l2=l2, msn=msn, pmode=int(rs.random_sample() < 0.1), emode=emode)


# ------------------------------------------------------------------------------------------------
# the whole command line: a random database with layer-2 k-mer sets, a sample, flags
# ------------------------------------------------------------------------------------------------
FLOW_FLAGS = [[], [], [], ["-b", "1"], ["-e", "1"], ["-l", "1"], ["-l", "2"], ["-s", "3"], ["-l", "1", "-e", "1"], ["-s", "1", "-b", "1"]]


def flow_spec(seed):
    """l1_spec(7000 + seed) with at least one multi-strain cluster, plus which clusters own a Kmer_Sets_L2 directory."""
    spec = l1_spec(7000 + seed)
    rs = np.random.RandomState(700000 + seed)
    if 3000 <= seed < 4000:                                           # seeds 3000..: a single-cluster database (tree_structure.txt "<id>\t" + tree.pkl,
        n = int(rs.randint(600, 2600))                                # Build_tree.py:283-375; identify.py:19-21 unpickles the tree)
        spec = dict(parent={1: None}, sites={1: n}, singleton={}, clusters={1: ["GCF_F%d_01_%02d" % (7000 + seed, j + 1) for j in range(int(rs.randint(2, 6)))]},
                    reconstructed=[], overlaps=[], invalid_nodes=[], db_seed=200000 + 7000 + seed, single_cluster=True,
                    l2={1: 710000 + 100 * seed + 1} if rs.random_sample() < 0.85 else {}, memory_db=bool(rs.random_sample() < 0.25))
        return spec
    T = synth.Tree(spec["parent"])
    if len(spec["clusters"]) < 2:                                    # make two of the leaves multi-strain clusters
        for l in [int(x) for x in rs.permutation(T.leaves)[:2]]:
            if l in spec["singleton"]:
                del spec["singleton"][l]
                spec["clusters"][l] = ["GCF_F%d_%02d_%02d" % (7000 + seed, l, j + 1) for j in range(int(rs.randint(2, 6)))]
    # sampled leaves get strong paths often (else most samples end in "nothing found")
    l2c = [int(x) for x in rs.permutation(sorted(spec["clusters"]))[:int(rs.randint(1, 4))]]
    if rs.random_sample() < 0.75:
        for l in l2c:
            for i in T.path(l):
                if spec["sites"][i] < 1000:
                    spec["sites"][i] = int(rs.randint(1000, 1600))
    spec["l2"] = {l: 710000 + 100 * seed + l for l in l2c}
    spec["memory_db"] = bool(rs.random_sample() < 0.25)
    return spec


def build_flow(seed, root_dir):
    spec = flow_spec(seed)
    db_dir = os.path.join(root_dir, "DB_W%d" % seed)
    info = synth.build_l1_db(db_dir, spec["parent"], spec["sites"], spec["db_seed"], spec["singleton"], spec["clusters"],
                             spec["reconstructed"], spec["overlaps"], invalid_nodes=spec["invalid_nodes"], single_cluster=spec.get("single_cluster", False))
    info["db_dir"] = db_dir
    info["spec"] = spec
    info["l2"] = {}
    C = len(info["tree"].leaves)
    cids = sorted(spec["l2"])
    for cid in cids:
        rs = np.random.RandomState(spec["l2"][cid])
        strains = spec["clusters"][cid]
        S = len(strains)
        G = 2 * S + 3
        pres = np.zeros((S, G), bool)
        pres[:, 0] = True
        for s in range(S):
            pres[s, 1 + s] = True
        pres[:, 1 + S:] = rs.random_sample((S, G - 1 - S)) < 0.45
        seg = [int(rs.randint(1200, 2400))] + [int(rs.randint(450, 950)) for _ in range(S)] + [int(rs.randint(150, 600)) for _ in range(G - 1 - S)]
        other = [c for c in cids if c != cid]
        shared_with = {1 + S: [other[0]]} if other and rs.random_sample() < 0.6 else None
        info["l2"][cid] = synth.build_l2_cluster(db_dir, cid, C, strains, seg, pres, seed=spec["l2"][cid] + 1, shared_with=shared_with,
                                                 k=flow_variant(seed)["k"])
    if spec["memory_db"]:
        open(os.path.join(db_dir, "Memory_DB"), "w").close()
    return info


def flow_reads(info, seed):
    rs = np.random.RandomState(800000 + seed)
    spec = info["spec"]
    if 2000 <= seed < 3000:                                           # seeds 2000..2999: samples at the edge of being a sample at all
        u = np.random.RandomState(810000 + seed).random_sample()
        leaf = info["leaf_genome"][sorted(spec["l2"])[0]]
        if u < 0.15:
            return b""                                                # an empty file
        if u < 0.3:
            return b"".join(b"@s%d\n%s\n+\n%s\n" % (i, leaf[40 * i:40 * i + 30], b"I" * 30) for i in range(200))      # every read shorter than k
        if u < 0.45:
            return b"".join(b"@n%d\n%s\n+\n%s\n" % (i, b"N" * 150, b"I" * 150) for i in range(100))                      # nothing but N
        if u < 0.6:
            return b"@one\n%s\n+\n%s\n" % (leaf[:150], b"I" * 150)                                                      # a single read
        if u < 0.75:
            return synth.simulate_reads([(leaf, 0.05)], 900000 + seed)                                                     # a handful of reads
    low = rs.random_sample() < 0.3
    gd = []
    for cid in sorted(spec["l2"]):
        if rs.random_sample() < 0.85:
            strains = spec["clusters"][cid]
            for j in rs.permutation(len(strains))[:int(rs.randint(1, min(3, len(strains)) + 1))]:
                g = info["leaf_genome"][cid] + b"N" + info["l2"][cid]["strain_extra"][strains[int(j)]]
                depth = float(np.exp(rs.uniform(np.log(0.5), np.log(3.0)))) if low else float(np.exp(rs.uniform(np.log(4.0), np.log(40.0))))
                gd.append((g, depth))
    others = [l for l in info["tree"].leaves if l not in spec["l2"]]
    if others and rs.random_sample() < 0.5:
        gd.append((info["leaf_genome"][int(rs.choice(others))], float(rs.uniform(2, 15))))
    if rs.random_sample() < 0.3 or not gd:
        gd.append((synth.rand_seq(rs, int(rs.randint(5000, 40000))), float(rs.uniform(1, 6))))
    return synth.simulate_reads(gd, 900000 + seed)


def flow_argv(seed):
    rs = np.random.RandomState(950000 + seed)
    k = flow_variant(seed)["k"]
    return list(FLOW_FLAGS[int(rs.randint(0, len(FLOW_FLAGS)))]) + (["-k", str(k)] if k != 31 else [])


def flow_variant(seed):
    """Seeds from 1000 on also vary what the first campaign held fixed: the layer-2 k (`-k`, the database's k-mer sets built with
    it), paired input (`-j`), gzip-compressed input."""
    if seed < 1000:
        return dict(k=31, paired=False, gz=False)
    rs = np.random.RandomState(970000 + seed)
    return dict(k=int(rs.choice([31, 31, 25, 21, 27, 19])), paired=bool(rs.random_sample() < 0.4), gz=bool(rs.random_sample() < 0.4))


def flow_inputs(info, seed, root):
    """Write the sample's file(s).  -> (paths as the command line takes them: [fq] or [fq1, fq2], the FASTQ bytes per file)."""
    import gzip
    reads = flow_reads(info, seed)
    v = flow_variant(seed)
    parts = [reads]
    if v["paired"]:
        recs = reads.split(b"\n")[:-1]
        recs = [b"\n".join(recs[i:i + 4]) + b"\n" for i in range(0, len(recs), 4)]
        parts = [b"".join(recs[0::2]), b"".join(recs[1::2])]
    paths = []
    for i, blob in enumerate(parts):
        p = os.path.join(root, "w%d_%d.fq" % (seed, i + 1) + (".gz" if v["gz"] else ""))
        if v["gz"]:
            with open(p, "wb") as f, gzip.GzipFile(fileobj=f, mode="wb", compresslevel=1 + seed % 9, mtime=0) as z:
                z.write(blob)
        else:
            with open(p, "wb") as f:
                f.write(blob)
        paths.append(p)
    return paths, parts


def flow_known_deviation(seed, memory_db):
    """The one place where the product knowingly does not follow the reference: a Memory_DB database with
    gzip-compressed reads.  identify_low_mem.jellyfish_count (identify_low_mem.py:67-75) hands the .gz file to jellyfish as it is --
    identify.py:81-84 pipes it through zcat --, jellyfish finds no record in the compressed bytes, every node comes back empty and the
    walk dies at the root (ZeroDivisionError).  The product inflates the file for both modules."""
    return bool(memory_db) and flow_variant(seed)["gz"]


# ------------------------------------------------------------------------------------------------
# input formats: what jellyfish's reader accepts, rendered at random
# ------------------------------------------------------------------------------------------------
FMT_KINDS = ["fq4", "fq4_at", "fq_wrap", "fa1", "fa_wrap", "fq4_nonl", "fa_nonl", "fq4_crlf", "fa_crlf", "fq_blank_tail", "fa_lower_mix", "fq_plus_name"]


def fmt_render(recs, kind, rs):
    """[(name, sequence)] -> bytes in one of the shapes a FASTA / FASTQ file comes in."""
    out = []
    nl = b"\r\n" if kind.endswith("crlf") else b"\n"
    for name, seq in recs:
        if kind.startswith("fa"):
            if kind == "fa_wrap":
                w = int(rs.choice([1, 7, 60, 70, 149]))
                body = nl.join(seq[i:i + w] for i in range(0, len(seq), w)) if seq else b""
            else:
                body = seq
            if kind == "fa_lower_mix" and rs.random_sample() < 0.3:
                body = bytes(c + 32 if (65 <= c <= 90 and rs.random_sample() < 0.5) else c for c in body)
            out.append(b">" + name + nl + body + nl)
        else:
            qual = bytes(rs.randint(33, 74, size=len(seq)).astype(np.uint8)) if kind != "fq4" else b"I" * len(seq)
            if kind == "fq4_at" and len(seq):
                qual = b"@" + qual[1:] if rs.random_sample() < 0.5 else b"+" + qual[1:]
            plus = b"+" + name if kind == "fq_plus_name" else b"+"
            if kind == "fq_wrap":
                w = int(rs.choice([13, 60, 100]))
                sq = nl.join(seq[i:i + w] for i in range(0, len(seq), w))
                ql = nl.join(qual[i:i + w] for i in range(0, len(qual), w))
                out.append(b"@" + name + nl + sq + nl + plus + nl + ql + nl)
            else:
                out.append(b"@" + name + nl + seq + nl + plus + nl + qual + nl)
    blob = b"".join(out)
    if kind.endswith("nonl") and blob.endswith(nl):
        blob = blob[:-len(nl)]
    if kind == "fq_blank_tail":
        blob += nl * int(rs.randint(1, 4))
    return blob


def fmt_case(seed, root_dir):
    """A small database and a sample of one or two files, each in its own random shape, plain or gzip-compressed.
    -> (info, [paths], [bytes as written, uncompressed], [kinds])"""
    import gzip
    rs = np.random.RandomState(1200000 + seed)
    spec = l1_spec(9000 + seed % 40)                                  # forty small-to-mid databases shared by the seeds
    db_dir = os.path.join(root_dir, "DB_X%d" % seed)
    info = synth.build_l1_db(db_dir, spec["parent"], spec["sites"], spec["db_seed"], spec["singleton"], spec["clusters"],
                             spec["reconstructed"], spec["overlaps"], invalid_nodes=spec["invalid_nodes"])
    info["db_dir"] = db_dir
    T = info["tree"]
    gd = [(info["leaf_genome"][int(l)], float(rs.uniform(1, 6))) for l in rs.permutation(T.leaves)[:int(rs.randint(1, 3))]]
    gd.append((synth.rand_seq(rs, 3000), 2.0))
    fq = synth.simulate_reads(gd, 1300000 + seed, read_len=int(rs.choice([150, 150, 100, 251, 75])))
    lines = fq.split(b"\n")
    recs = [(lines[i][1:], lines[i + 1]) for i in range(0, len(lines) - 1, 4)]
    if rs.random_sample() < 0.3:                                      # a few records of odd lengths: empty, shorter than k, very long
        recs.insert(int(rs.randint(0, len(recs))), (b"empty", b""))
        recs.insert(int(rs.randint(0, len(recs))), (b"short", recs[0][1][:17]))
        recs.insert(int(rs.randint(0, len(recs))), (b"long", b"".join(r[1] for r in recs[:40])))
    n_files = 1 + int(rs.random_sample() < 0.4)
    parts = [recs] if n_files == 1 else [recs[0::2], recs[1::2]]
    paths, blobs, kinds = [], [], []
    for i, part in enumerate(parts):
        kind = str(rs.choice(FMT_KINDS))
        blob = fmt_render(part, kind, rs)
        gz = bool(rs.random_sample() < 0.3)
        p = os.path.join(root_dir, "x%d_%d.%s%s" % (seed, i + 1, "fa" if kind.startswith("fa") else "fq", ".gz" if gz else ""))
        if gz:
            with open(p, "wb") as f, gzip.GzipFile(fileobj=f, mode="wb", compresslevel=1 + seed % 9, mtime=0) as z:
                z.write(blob)
        else:
            with open(p, "wb") as f:
                f.write(blob)
        paths.append(p); blobs.append(blob); kinds.append(kind + ("+gz" if gz else ""))
    return info, paths, blobs, kinds


def fmt_known_deviation(kinds):
    """Inputs on which the reference's pipeline loses reads and the product does not:
    * a pair of files of which exactly ONE is gzip-compressed: identify.py:81-84 runs `zcat file1 file2 | jellyfish` as soon as either
      name ends in .gz, zcat refuses the plain file ("not in gzip format") and only the compressed one is counted;
    * a FASTQ file whose last line has no newline: jellyfish 2.3.0 takes the file for truncated and drops the buffer it was parsing
      (every read of a small file, the last few of a large one);
    * a pair of .gz files, FASTQ first and FASTA second: zcat makes ONE stream of them, jellyfish takes the stream's format from its
      first byte and the FASTA records behind the FASTQ ones are lost (as two plain files each is read in its own format)."""
    if len(kinds) == 2 and kinds[0].endswith("+gz") != kinds[1].endswith("+gz"):
        return "one of two files is .gz"
    if any(k.split("+")[0] == "fq4_nonl" for k in kinds):
        return "FASTQ without a final newline"
    if len(kinds) == 2 and all(k.endswith("+gz") for k in kinds) and kinds[0][:2] == "fq" and kinds[1][:2] == "fa":
        return "a .gz pair, FASTQ then FASTA"
    return None


# ------------------------------------------------------------------------------------------------
# layer 1 once more, on a kmer.fa with rows no node lists: duplicates, rows with an N, lower-case rows
# ------------------------------------------------------------------------------------------------
def build_l1x(seed, root_dir):
    """l1_spec(5000 + seed)'s database with 1-40 extra rows spliced into kmer.fa (synth.build_l1_db extra_rows): exact copies of node
    rows in front of or behind the original (kmer_index_dict keeps the LAST row of a k-mer, identify.py:91-95: a copy behind it takes
    the node's row out of every profile), rows with an N (jellyfish never reports them), lower-case copies of node rows, and lower-case
    rows of k-mers that occur in the reads but in no node (the junctions between two nodes' sequences in a leaf's genome):
    identify.py looks rows up in upper case, identify_low_mem.py as written -- its KeyError is part of the result."""
    import shutil
    spec = l1_spec(5000 + seed)
    rs = np.random.RandomState(1400000 + seed)
    db_dir = os.path.join(root_dir, "DB_Y%d" % seed)
    args = (spec["parent"], spec["sites"], spec["db_seed"], spec["singleton"], spec["clusters"], spec["reconstructed"], spec["overlaps"])
    first = synth.build_l1_db(db_dir, *args, invalid_nodes=spec["invalid_nodes"])
    rows = open(os.path.join(db_dir, "Tree_database", "kmer.fa"), "rb").read().split(b"\n")[1::2]
    shutil.rmtree(db_dir)
    T = first["tree"]
    junction = []
    for l in T.leaves:
        p = T.path(l)
        for a, b in zip(p[:-1], p[1:]):
            j = first["node_seq"][a][-(K - 1):] + first["node_seq"][b][:K - 1]
            junction += [j[i:i + K] for i in range(0, K - 1, 3)]
    extra = []
    lower_too = rs.random_sample() < 0.35                             # (a lower-case row ends identify_low_mem / identify_low_depth at once)
    for _ in range(int(rs.randint(1, 41))):
        pos = int(rs.randint(0, len(rows) + 1))
        u = rs.random_sample() * (1.0 if lower_too else 0.55)
        if u < 0.4:
            txt = rows[int(rs.randint(0, len(rows)))]
        elif u < 0.55:
            txt = bytearray(synth.rand_seq(rs, K)); txt[int(rs.randint(0, K))] = ord("N"); txt = bytes(txt)
        elif u < 0.8:
            txt = rows[int(rs.randint(0, len(rows)))].lower()
        else:
            txt = junction[int(rs.randint(0, len(junction)))].lower() if junction else rows[0].lower()
        extra.append((pos, txt))
    info = synth.build_l1_db(db_dir, *args, extra_rows=extra, invalid_nodes=spec["invalid_nodes"])
    info["db_dir"] = db_dir
    info["spec"] = spec
    return info

"""Layer 1 end to end on the GPU: strainscan_amd.identify / identify_low_mem /
identify_low_depth against the reference's golden results (real jellyfish + reference Python)."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from tests import hostlogic as hl
from tests import scenarios as sc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "l1_search.json")) as f:
        return json.load(f)


def _run(fn, *a):
    buf = io.StringIO()
    err = res = None
    with contextlib.redirect_stdout(buf):
        try:
            res = fn(*a)
        except BaseException as e:  # noqa: B902
            err = type(e).__name__
    return res, err, buf.getvalue()


@pytest.mark.parametrize("sname", list(sc.L1_SAMPLES))
def test_identify_cluster(sname, golden, l1_dbs, l1_reads):
    from strainscan_amd import identify, identify_low_mem, identify_low_depth
    dbn = sc.L1_SAMPLES[sname][0]
    tdb = os.path.join(l1_dbs[dbn]["db_dir"], "Tree_database")
    fq = l1_reads[sname][0]
    mods = {"identify": identify, "identify_low_mem": identify_low_mem}
    for run in golden[sname]["runs"]:
        np.random.seed(sc.POISSON_SEED)
        res, err, text = _run(mods[run["module"]].identify_cluster, (fq, ""), tdb, list(run["cutoff"]))
        tag = (sname, run["module"], run["cutoff"])
        assert err == run["error"], (tag, err, text[-300:])
        if err is None:
            hl.assert_result_equal(res, run["result"], tag)
        got_tr = hl.parse_trace(text)
        assert [g[0] for g in got_tr] == [w[0] for w in run["trace"]], tag
        for g, w in zip(got_tr, run["trace"]):               # abundance | coverage, valid length of every visited node: the device's
            assert len(g) == len(w), (tag, g, w)             # per-node statistics as the reference printed them (%f)
            if len(w) == 4:
                assert abs(g[1] - w[1]) < 2e-6 and abs(g[2] - w[2]) < 2e-6 and g[3] == w[3], (tag, g, w)
    res, err, _ = _run(identify_low_depth.identify_ranks, (fq, ""), tdb)
    want = golden[sname]["ranks"]
    assert err == want["error"]
    assert [a for a, _ in res] == [a for a, _ in want["result"]]
    for (a, b), (_, wb) in zip(res, want["result"]):
        assert abs(b - wb) <= 1e-12 * max(1.0, abs(wb))


@pytest.mark.parametrize("modname", ["identify", "identify_low_mem"])
@pytest.mark.parametrize("sname", ["A_mix3", "B_mix"])
def test_reference_helpers_under_their_own_names(modname, sname, golden, l1_dbs, l1_reads):
    """What a caller of the reference's modules finds besides identify_cluster (identify.py:73-136 and :45-70,
    identify_low_mem.py:67-127): jellyfish_count -> match_results (here a read-only view over the device's counts: the
    mapping protocol of the dict it replaces, zero counts included, rows with a non-ACGT k-mer absent), match_node,
    del_outlier, piecewise, get_node_label.  Against the real jellyfish's counts (sha256 in the golden file) and against
    the per-node lines the reference printed while it searched: a node that is not a reconstructed one prints
    piecewise(match_node(...)) with abundances below the cutoff set to 0 (identify.py:297-314)."""
    import importlib
    from strainscan_amd import identify
    from strainscan_amd.tree import read_tree_structure
    from tests import synth
    mod = importlib.import_module("strainscan_amd." + modname)
    dbn = sc.L1_SAMPLES[sname][0]
    info = l1_dbs[dbn]
    tdb = os.path.join(info["db_dir"], "Tree_database")
    fq = l1_reads[sname][0]
    mr = mod.jellyfish_count((fq, ""), tdb)
    g = golden[sname]
    if modname == "identify":                                  # (the golden counts are identify.py's: upper-cased keys)
        cnt = np.zeros(info["n_rows"], np.uint32)
        for k_, v_ in mr.items():
            cnt[k_] = v_
        assert synth.sha256_of(cnt.tobytes()) == g["counts_sha256"]
        assert len(mr) == g["n_valid"] and int(cnt.sum()) == g["counts_sum"]
    keys = list(mr)
    assert len(keys) == len(mr) == len(mr.keys()) and all(k in mr for k in keys[:50])
    assert -1 not in mr and info["n_rows"] not in mr and mr.get(info["n_rows"], "none") == "none"
    missing = sorted(set(range(info["n_rows"])) - set(keys))
    if missing:                                                # an N row of kmer.fa is no key (identify.py:93-95)
        with pytest.raises(KeyError):
            mr[missing[0]]
        assert mr.get(missing[0]) is None
    assert mr.get(keys[0]) == mr[keys[0]] and dict(zip(keys[:20], [mr[k] for k in keys[:20]])) == dict(list(mr.items())[:20])
    # the same string form of the input as the reference's callers use (StrainScan.py:182: a tuple; identify.py:75-76 joins it)
    assert identify._paths("a.fq  b.fq") == ["a.fq", "b.fq"] and identify._paths(("a.fq", "")) == ["a.fq"]
    tree, _ = read_tree_structure(tdb)
    for n in tree.all_nodes():
        n.data = [-1] * 5                                       # identify.py:406-407
    mod.get_node_label(tdb, tree)
    with open(os.path.join(tdb, "node_length.txt")) as f:
        flen = {int(a): int(b) for a, b in (ln.split() for ln in f if ln.strip())}
    weak, strong = (500, 1500) if modname == "identify_low_mem" else (1000, 3000)
    leaves = {n.identifier for n in tree.leaves()}
    for n in tree.all_nodes():
        lab, ln = n.data[0], flen[n.identifier]
        if isinstance(lab, str):
            assert lab == ("o1" if ln < strong else "o2")
        elif ln < weak:
            assert lab == (1 if (modname == "identify" and n.identifier in leaves) else 0), (n.identifier, lab, ln)
        else:
            assert lab == (1 if ln < strong else 2)
    checked = 0
    for run in g["runs"]:
        if run["module"] != modname or run["error"] is not None:
            continue
        cov_cut, _, ab_cut = run["cutoff"]
        for w in run["trace"]:
            lab = tree.get_node(w[0]).data[0]
            if len(w) != 4 or isinstance(lab, str) or lab == 0:
                continue
            length, prof = mod.match_node(mr, tdb, w[0], set(keys))
            assert length == w[3], (w, length)
            if length == 0:
                continue
            cov = len(prof) / length
            ab = mod.piecewise(cov_cut, cov, lab, prof)
            ab = 0 if ab < ab_cut else ab
            assert abs(cov - w[2]) < 2e-6 and abs(ab - w[1]) < 2e-6, (run["cutoff"], w, cov, ab)
            checked += 1
    assert checked >= 3
    # del_outlier: values at or above 100 x the median go (identify.py:106-112); a median of x.5 compares as a float
    assert mod.del_outlier([1, 1, 2, 2, 150]) == [1, 1, 2, 2, 150]          # median 2: cutoff 200
    assert mod.del_outlier([1, 1, 1, 2, 100, 250]) == [1, 1, 1, 2, 100]     # median 1.5: cutoff 150
    assert mod.del_outlier([1, 1, 100]) == [1, 1]                           # at the cutoff: dropped


def test_gz_and_pair_inputs(l1_dbs, l1_reads, tmp_path):
    """(fq1, fq2) pairs and .gz inputs (identify.py:75-87): counts add over the two files."""
    import gzip
    from strainscan_amd import identify
    tdb = os.path.join(l1_dbs["A"]["db_dir"], "Tree_database")
    fq, data = l1_reads["A_mix3"]
    recs = data.split(b"@r")[1:]
    half = len(recs) // 2
    p1, p2 = tmp_path / "a_1.fq", tmp_path / "a_2.fq.gz"
    p1.write_bytes(b"".join(b"@r" + r for r in recs[:half]))
    with gzip.open(p2, "wb") as f:
        f.write(b"".join(b"@r" + r for r in recs[half:]))
    os.environ["STRAINSCAN_QUIET"] = "1"
    try:
        whole = identify.jellyfish_count((fq, ""), tdb)
        c0 = whole.counts.copy()
        both = identify.jellyfish_count((str(p1), str(p2)), tdb)
        assert np.array_equal(both.counts, c0)
    finally:
        del os.environ["STRAINSCAN_QUIET"]


def test_cli_flags_b_and_l(golden, l1_dbs, l1_reads, tmp_path):
    """`strainscan -b 1 -l 2`: strain_prob.txt (StrainScan.py:98-111) carries identify_ranks' scores and
    the -l 2 cutoffs [0.005, 0.01, 1] reach the tree walk (StrainScan.py:213-217)."""
    from strainscan_amd import StrainScan
    info = l1_dbs["A"]
    out = tmp_path / "o"
    np.random.seed(sc.POISSON_SEED)
    with contextlib.redirect_stdout(io.StringIO()):
        try:
            StrainScan.main(["-i", l1_reads["A_low"][0], "-d", info["db_dir"], "-o", str(out), "-b", "1", "-l", "2"])
        except (SystemExit, FileNotFoundError, RuntimeError):
            pass        # no Kmer_Sets_L2 in this fixture database: layer 1 and the -b report are what is checked
    want = golden["A_low"]["ranks"]["result"]
    lines = (out / "strain_prob.txt").read_text().strip().split("\n")
    assert lines[0] == "Cluster_ID\tProbability\tNumber_of_strains\tStrains_in_the_cluster"
    assert len(lines) - 1 == len(want)
    for ln, (leaf, score) in zip(lines[1:], want):
        f = ln.split("\t")
        assert f[0] == "C%d" % leaf and abs(float(f[1]) - score) <= 1e-12 * max(1.0, abs(score))
    names = dict((l.split("\t")[0], l.rstrip("\n").split("\t")[2]) for l in
                 open(os.path.join(info["db_dir"], "Tree_database", "hclsMap_95_recls.txt")))
    assert lines[1].split("\t")[3] == names[str(want[0][0])]


@pytest.mark.gpu
def test_identify_cluster_from_files_cold_and_cached(tmp_path, monkeypatch):
    """A 23-leaf synthetic Tree_database written in the reference's on-disk format (scripts/bench_cli.py's writer)
    and a paired FASTQ sample of a 70/20/10 three-strain mix: identify_cluster finds exactly those three leaves with
    those proportions; the second call (tree cache + index image read back, new process state) returns the same
    dict; hit counts of the cached index equal those of the freshly built one."""
    import torch
    import bench
    from scripts import bench_cli
    from strainscan_amd import db as ssdb
    from strainscan_amd import identify
    monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / "cache"))
    monkeypatch.setenv("STRAINSCAN_QUIET", "1")
    dev = torch.device("cuda", 0)
    C, n_reads = 23, 600_000
    spec = bench.make_db(torch, dev, C, seed=20231013)
    tdir = bench_cli.write_db(torch, dev, spec, C, str(tmp_path / "db"))
    r = bench.make_reads(torch, dev, spec, n_reads, seed=2, hit_frac=0.05)
    half = n_reads // 2
    fq = [str(tmp_path / "s_1.fq"), str(tmp_path / "s_2.fq")]
    bench_cli.write_fastq(r[: half * 151], half, fq[0])
    bench_cli.write_fastq(r[half * 151:], n_reads - half, fq[1])
    del r
    ssdb.clear_cache()
    first = identify.identify_cluster((fq[0], fq[1]), tdir, [0.1, 0.4, 1])
    counts_first = ssdb.tree_image(tdir, True).counts.copy()
    assert len(first) == 3
    per = sorted((float(v["cls_per"]) for v in first.values()), reverse=True)
    assert abs(per[0] - 0.7) < 0.02 and abs(per[1] - 0.2) < 0.02 and abs(per[2] - 0.1) < 0.02
    assert all(v["strain"] == "strain_%d" % k and float(v["cls_cov"]) > 0.9 for k, v in first.items())
    ssdb.wait_cache_writes()
    assert {f[:5] for f in os.listdir(tmp_path / "cache")} == {"tree_", "index"}
    ssdb.clear_cache()
    second = identify.identify_cluster((fq[0], fq[1]), tdir, [0.1, 0.4, 1])
    assert dict(second) == dict(first)
    assert np.array_equal(ssdb.tree_image(tdir, True).counts, counts_first)
    ssdb.clear_cache()
    # damaged cache files are ignored, the image is built from the database again (and the files written again): another
    # magic, a cut file, a header whose sizes do not add up -- for the tree arrays; a cut file and a flipped header field for the index image
    cdir = tmp_path / "cache"
    for damage in ("magic", "cut", "header", "index_cut", "index_header"):
        ssdb.wait_cache_writes()
        name = [f for f in sorted(os.listdir(cdir)) if f.startswith("index" if damage.startswith("index") else "tree_")][0]
        raw = (cdir / name).read_bytes()
        if damage == "magic":
            bad = b"NOTATREE" + raw[8:]
        elif damage in ("cut", "index_cut"):
            bad = raw[: len(raw) * 2 // 3]
        elif damage == "header":
            bad = raw[:24] + (int.from_bytes(raw[24:32], "little") + 1).to_bytes(8, "little") + raw[32:]      # n_total + 1
        else:
            bad = raw[:16] + bytes(b ^ 0x5A for b in raw[16:24]) + raw[24:]
        (cdir / name).write_bytes(bad)
        if damage in ("magic", "cut", "header"):
            with pytest.raises(ValueError, match={"magic": "not a tree cache", "cut": "truncated", "header": "inconsistent"}[damage]):
                ssdb._read_tree_cache(str(cdir / name))
        ssdb.clear_cache()
        again = identify.identify_cluster((fq[0], fq[1]), tdir, [0.1, 0.4, 1])
        assert dict(again) == dict(first), damage
        assert np.array_equal(ssdb.tree_image(tdir, True).counts, counts_first), damage
        ssdb.wait_cache_writes()
        ssdb.clear_cache()
    # the same sample as a .fastq.gz pair: inflated and reduced to its sequence lines on the device (the default), and with
    # the host inflaters: same result, same counts
    import ctypes as C
    import subprocess
    from strainscan_amd import _lib
    for p in fq:
        subprocess.check_call(["gzip", "-1", "-k", p])
    gz = [p + ".gz" for p in fq]
    for mode, on_device in (("1", 2), ("0", 0)):
        monkeypatch.setenv("SS_GZ_GPU", mode)
        a, b = C.c_uint64(), C.c_uint64()
        _lib.lib().ss_gz_gpu_counters(C.byref(a), C.byref(b))
        h0 = a.value
        third = identify.identify_cluster((gz[0], gz[1]), tdir, [0.1, 0.4, 1])
        assert dict(third) == dict(first), mode
        assert np.array_equal(ssdb.tree_image(tdir, True).counts, counts_first), mode
        _lib.lib().ss_gz_gpu_counters(C.byref(a), C.byref(b))
        assert a.value - h0 == on_device, mode
        ssdb.clear_cache()


def _rank_paths_restated(parent_of, leaves, frac):
    """identify_low_depth.py:134-151 restated for the test: geometric mean over the root path of
    x = 1 if frac > 0.05 else log(180 frac + 1, 10); nodes with frac == -1 are skipped; zero scores dropped."""
    from math import log
    out = {}
    for leaf in leaves:
        path, n = [], leaf
        while n is not None:
            path.append(n)
            n = parent_of[n]
        path = path[::-1]
        N = len([i for i in path if frac[i] != -1])
        score = 1
        for i in path:
            if frac[i] == -1:
                continue
            x = 1 if frac[i] > 0.05 else log(180 * frac[i] + 1, 10)
            score = score * pow(x, 1 / N)
        if score != 0:
            out[leaf] = score
    return sorted(out.items(), key=lambda kv: kv[1], reverse=True)


@pytest.mark.parametrize("shape", ["sampled", "contiguous"])
def test_config4_low_depth_cli(shape, tmp_path, monkeypatch):
    """BASELINE configs[4]: low-depth mode (-b 1) on an M. tuberculosis-shaped tree (25 clusters = 49 nodes), a mixed
    sample of two strains 80/20 at 0.5x coverage, through the CLI entry (StrainScan.main, StrainScan.py:98-111,186-191).
    strain_prob.txt must carry the scores the reference's formula gives on the ORACLE's counts of the same files
    (identify_low_depth.py:104-156), to 1e-12, in the same order; the two true clusters lead the ranking."""
    import torch
    import bench
    from oracle import oracle as orc
    from scripts import bench_cli
    from strainscan_amd import StrainScan, db as ssdb
    monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / "cache"))
    dev = torch.device("cuda", 0)
    C = 25
    spec = bench.make_db(torch, dev, C, seed=4242, shape=shape, hit_frac=0.05)
    dbdir = str(tmp_path / "db")
    tdir = bench_cli.write_db(torch, dev, spec, C, dbdir)
    # genome of a leaf = its root path's stretches + filler (bench.make_reads); 0.5x of a ~5 path-nodes genome
    n_db = int(spec["sites"].mean() * 5)
    genome_len = int(n_db / 0.05)
    n_reads = int(0.5 * genome_len / 150 / 0.8)           # the dominant strain (80 %) at ~0.5x
    r = bench.make_reads(torch, dev, spec, n_reads, seed=9, hit_frac=0.05, mix=(0.8, 0.2))
    half = n_reads // 2
    fq = [str(tmp_path / "s_1.fq"), str(tmp_path / "s_2.fq")]
    bench.write_fastq(r[: half * 151], half, fq[0])
    bench.write_fastq(r[half * 151:], n_reads - half, fq[1])
    ssdb.clear_cache()
    out = tmp_path / "o"
    with contextlib.redirect_stdout(io.StringIO()):
        try:
            StrainScan.main(["-i", fq[0], "-j", fq[1], "-d", dbdir, "-o", str(out), "-b", "1"])
        except (SystemExit, FileNotFoundError, RuntimeError):
            pass        # no Kmer_Sets_L2 in this synthetic database: layer 1 and the -b report are what is checked
    lines = (out / "strain_prob.txt").read_text().strip().split("\n")
    assert lines[0] == "Cluster_ID\tProbability\tNumber_of_strains\tStrains_in_the_cluster"
    # expectation from the oracle's counts of the same two files
    kfa = open(os.path.join(tdir, "kmer.fa"), "rb").read()
    counts, valid = orc.jellyfish_count(kfa, [open(fq[0], "rb").read(), open(fq[1], "rb").read()], k=31, upper=False)
    ids = [bench.heap_to_id(h, C) for h in range(spec["n_nodes"])]
    parent_of = {bench.heap_to_id(h, C): (None if h == 0 else bench.heap_to_id((h - 1) // 2, C)) for h in range(spec["n_nodes"])}
    row_off = spec["row_off"].astype(np.int64)
    frac = {}
    for h, nid in enumerate(ids):
        o = orc.match_node(counts, valid, spec["rows"][int(row_off[h]):int(row_off[h + 1])].astype(np.int64))
        frac[nid] = -1 if o["length"] < 1000 else o["n_kept"] / o["length"]
    want = _rank_paths_restated(parent_of, list(range(1, C + 1)), frac)
    assert len(lines) - 1 == len(want) and len(want) >= 2
    got = [(int(ln.split("\t")[0][1:]), float(ln.split("\t")[1])) for ln in lines[1:]]
    assert sorted(a for a, _ in got) == sorted(a for a, _ in want)
    for (a, b), (wa, wb) in zip(got, want):
        assert abs(b - wb) <= 1e-12 * max(1.0, abs(wb))
        assert a == wa or abs(b - dict(want)[a]) <= 1e-12      # equal scores may swap
    # the strains the reads came from: make_reads draws its leaves with RandomState(seed)
    rs = np.random.RandomState(9)
    true_leaves = {bench.heap_to_id(int(h), C) for h in rs.choice(np.arange(spec["n_nodes"] // 2, spec["n_nodes"]), size=2, replace=False)}
    assert {a for a, _ in got[:2]} == true_leaves
    # `-b 1` on a COLD cache loaded the tree twice (identify_ranks, then identify_cluster: identify_low_depth.py:119,
    # identify.py:402): exactly one tree image and one index image, no temp file, and the mapped image holds the
    # arrays of a fresh parse
    ssdb.wait_cache_writes()
    files = sorted(os.listdir(tmp_path / "cache"))
    # (two index images: identify_ranks keys the table by the raw text, identify_cluster by its upper case)
    assert [f[:5] for f in files] == ["index", "index", "tree_"] and all(f.endswith(".bin") for f in files), files
    cached = ssdb._read_tree_cache(str(tmp_path / "cache" / files[2]))
    monkeypatch.setenv("SS_IMAGE_CACHE", "off")
    fresh = ssdb.load_tree(tdir)
    assert list(cached.ids) == list(fresh.ids)
    for name in ("keys", "flags", "rows", "offs", "urows", "uoffs"):
        assert np.array_equal(getattr(cached, name), getattr(fresh, name)), name
    ssdb.clear_cache()


@pytest.mark.parametrize("stname", list(sc.L1_STEPS))
def test_search_step_on_the_device(stname, golden_dir, l1_dbs, l1_reads):
    """The hand-made search() states of tests/scenarios.py L1_STEPS (the "both weak" branch, identify.py:264-273, which
    identify_cluster itself can never take) through the DEVICE provider: the tree image, the resident reads' scan and
    cst.ImageProvider instead of the oracle's counts -- against what the reference's search() did."""
    from strainscan_amd import cst, db as ssdb, identify, identify_low_mem
    with open(os.path.join(golden_dir, "l1_search_steps.json")) as f:
        want = json.load(f)[stname]
    sname, modname, cut, override, pend = sc.L1_STEPS[stname]
    mod = {"identify": identify, "identify_low_mem": identify_low_mem}[modname]
    tdb = os.path.join(l1_dbs[sc.L1_SAMPLES[sname][0]]["db_dir"], "Tree_database")
    img = ssdb.tree_image(tdb, mod._UPPER_KEYS)
    img.scan([l1_reads[sname][0]])
    lines = []
    w = cst.Walk(cst.ImageProvider(img), tdb, list(cut), mod._PARAMS, out=lambda *a: lines.append(" ".join(str(x) for x in a)))
    for nid, (cat, acc) in override.items():
        w.tree.get_node(nid).data[0] = cat
        w.tree.get_node(nid).data[1] = acc
    w.pending[:] = [[w.tree.get_node(i) for i in g] for g in pend]
    res_temp, err = [], None
    try:
        w.search(res_temp)
    except Exception as e:
        err = type(e).__name__
    assert err == want["error"]
    assert [[n.identifier for n in g] for g in w.pending] == want["pending"]
    assert [n.identifier for n in res_temp] == want["res_temp"]
    assert [n.identifier for n in w.qualified_parents] == want["qualified_parents"]
    assert {str(n.identifier): list(n.data) for n in w.tree.all_nodes()} == want["data"]
    for key, got in (("length", w.length), ("cov", w.cov), ("abundance", w.abundance)):
        got = {str(n.identifier): v for n, v in got.items()}
        assert sorted(got) == sorted(want[key]), (stname, key)
        for n, v in want[key].items():
            assert abs(float(got[n]) - float(v)) <= 1e-9 * max(1.0, abs(float(v))), (stname, key, n)

#!/opt/conda/bin/python3.9
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference and /opt/conda/bin/python3.9 with
numpy 1.26 / scipy 1.7 / scikit-learn 0.24.2):

    /opt/conda/bin/python3.9 tests/golden/make_golden.py            # everything: ~5.5 min; every file regenerates byte-identically
    /opt/conda/bin/python3.9 tests/golden/make_golden.py mid|built|k25    # round 6's sections alone

What it does, without modifying or copying any reference file into the repo:
  * copies /root/reference/library to a scratch dir (the wrappers exec the jellyfish ELF that
    sits next to identify.py, and the mount is read-only / not executable) and chmod +x's it;
  * puts tests/golden/_standin (a treelib stand-in, see its docstring) on sys.path;
  * builds the synthetic inputs of tests/scenarios.py, runs the reference functions on them and
    stores inputs' sha256 + reference outputs as JSON / npz next to this script.
Nothing here is imported by the product or by the tests; only the written data files are.
(The fuzz_*.json / fuzz_l2_arrays.npz files are written by tests/golden/fuzz_reference.py keep, which borrows this script's helpers:
random scenarios of tests/scenarios_fuzz.py through the same reference, 4 min.)
"""
import contextlib
import io
import json
import os
import pickle
import re
import shutil
import subprocess
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference"

warnings.filterwarnings("ignore")


def setup_reference(scratch):
    lib = os.path.join(scratch, "ref", "library")
    shutil.copytree(os.path.join(REF, "library"), lib)
    jf = os.path.join(lib, "jellyfish-linux")
    os.chmod(jf, 0o755)
    sys.path[:0] = [os.path.join(HERE, "_standin"), lib]
    return lib, jf


def jsonable(o):
    if isinstance(o, dict):
        return {str(k): jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, np.ndarray):
        return jsonable(o.tolist())
    return o


def dump_json(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(jsonable(obj), f, indent=1, sort_keys=True)
    print("wrote", name)


TRACE_RE = re.compile(r"^(\d+):\s+(-?[\d.]+(?:e[-+]?\d+)?|nan) \| (-?[\d.]+(?:e[-+]?\d+)?|nan)\s+(\d+)$")


def parse_trace(text):
    """stdout of identify_cluster -> list of [node, abundance, cov, length] in print order."""
    out = []
    for ln in text.splitlines():
        m = TRACE_RE.match(ln.strip())
        if m:
            out.append([int(m.group(1)), float(m.group(2)), float(m.group(3)), int(m.group(4))])
        elif ln.strip().endswith("weak") and ":" in ln:
            out.append([int(ln.split(":")[0]), "weak"])
    return out


def run_captured(fn, *a, **kw):
    buf = io.StringIO()
    err = None
    res = None
    with contextlib.redirect_stdout(buf):
        try:
            res = fn(*a, **kw)
        except BaseException as e:  # the reference's exceptions are part of its behaviour
            err = type(e).__name__
    return res, err, buf.getvalue()


def _patch_sklearn(l2mod, captured):
    """The reference module's two sklearn names replaced by subclasses that record what the fits produced."""
    from sklearn.linear_model import ElasticNetCV as _ENCV, ElasticNet as _EN

    class CapCV(_ENCV):
        def fit(self, X, y):
            r = super().fit(X, y)
            captured["alphas_"] = np.array(self.alphas_)
            captured["mse_path_"] = np.array(self.mse_path_)
            captured["n_rows"] = int(len(y))
            captured["p"] = int(np.asarray(X).shape[1])
            return r

    class CapEN(_EN):
        def fit(self, X, y):
            r = super().fit(X, y)
            captured["alpha"] = float(self.alpha)
            captured["coef_"] = np.array(self.coef_)
            captured["n_iter_"] = int(np.max(self.n_iter_))
            return r

    l2mod.ElasticNetCV = CapCV
    l2mod.ElasticNet = CapEN


def _reference_cli(scratch, argv, seed):
    """The reference's own StrainScan.py (copied to scratch next to library/: it imports `library.*` relative to the
    CWD, StrainScan.py:5-7) as a child process under this interpreter, numpy's global generator seeded first (the
    Poisson draw of adjust_profile, identify.py:214, is unseeded).  -> (return code, stdout)."""
    root = os.path.join(scratch, "ref")
    if not os.path.exists(os.path.join(root, "StrainScan.py")):
        shutil.copy(os.path.join(REF, "StrainScan.py"), root)
    drv = ("import sys, runpy, numpy, warnings; warnings.filterwarnings('ignore'); sys.path.insert(0, %r); "
           "numpy.random.seed(%d); sys.argv = ['StrainScan.py'] + %r; runpy.run_path('StrainScan.py', run_name='__main__')"
           % (os.path.join(HERE, "_standin"), seed, list(argv)))
    r = subprocess.run([sys.executable, "-c", drv], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode, r.stdout.decode(), r.stderr.decode()


def mid_section(scratch, identify, identify_low_mem, identify_low_depth, sc, synth):
    """configs[0]'s shape (53 clusters / 157 strains / 105 nodes, SE reads): layer 1 under every cutoff and both
    modules, identify_ranks, and the reference's command line end to end (tests/scenarios_mid.py)."""
    from tests import scenarios_mid as sm
    infos = {"DB_M": sm.build_mid(scratch), "DB_Mmem": sm.build_mid(scratch, memory_db=True)}
    info = infos["DB_M"]
    tdb = os.path.join(info["db_dir"], "Tree_database")
    kfa = open(os.path.join(tdb, "kmer.fa"), "rb").read()
    l1 = {}
    fqs = {}
    for sname in sm.mid_samples(info):
        reads = sm.mid_reads(info, sname)
        fq = os.path.join(scratch, sname + ".fq")
        open(fq, "wb").write(reads)
        fqs[sname] = fq
        entry = dict(sha256=synth.sha256_of(kfa, reads), n_reads=reads.count(b"\n") // 4, runs=[])
        mr = identify.jellyfish_count((fq, ""), tdb)
        cnt = np.zeros(info["n_rows"], np.int64)
        for k_, v_ in mr.items():
            cnt[k_] = v_
        entry["counts_sha256"] = synth.sha256_of(cnt.astype(np.uint32).tobytes())
        entry["counts_sum"] = int(cnt.sum())
        entry["n_valid"] = len(mr)
        for cut in sc.CUTOFFS:
            for modname, mod in (("identify", identify), ("identify_low_mem", identify_low_mem)):
                np.random.seed(sc.POISSON_SEED)
                res, err, out = run_captured(mod.identify_cluster, (fq, ""), tdb, list(cut))
                entry["runs"].append(dict(module=modname, cutoff=cut, error=err,
                                          result=None if res is None else {int(k): dict(v) for k, v in res.items()},
                                          trace=parse_trace(out)))
        res, err, out = run_captured(identify_low_depth.identify_ranks, (fq, ""), tdb)
        entry["ranks"] = dict(error=err, result=None if res is None else [[int(a), float(b)] for a, b in res])
        l1[sname] = entry
        print("mid L1", sname, [(r["module"], r["cutoff"][0], r["error"], sorted((r["result"] or {}).keys()))
                                for r in entry["runs"]][:4])
    dump_json("mid_l1.json", l1)

    flow = {}
    for name, (sname, dbn, extra) in sm.MID_FLOW.items():
        outdir = os.path.join(scratch, "flow_" + name)
        rc, out, errtxt = _reference_cli(scratch, ["-i", fqs[sname], "-d", infos[dbn]["db_dir"], "-o", outdir] + extra,
                                         sc.POISSON_SEED)
        files = {}
        for r_, _, fs in os.walk(outdir):
            for f_ in fs:
                p_ = os.path.join(r_, f_)
                files[os.path.relpath(p_, outdir)] = open(p_).read()
        cls_line = [ln for ln in out.splitlines() if ln.startswith("defaultdict(")]
        tb = [ln for ln in errtxt.splitlines() if re.match(r"^[A-Za-z_.]*(Error|Exception)\b", ln)]
        flow[name] = dict(sample=sname, db=dbn, argv=extra, returncode=rc, error=(tb[-1].split(":")[0] if tb else None),
                          cls_dict=cls_line[-1][cls_line[-1].index("{"):-1] if cls_line else None,
                          trace=parse_trace(out),
                          messages=[ln for ln in out.splitlines() if ln.startswith(("- ", "Warning")) and "running time" not in ln],      # (no wall-clock times)
                          files=files)
        print("mid flow", name, rc, flow[name]["error"], sorted(files))
    dump_json("mid_flow.json", flow)


def l2_big_section(scratch, l2mod, captured, synth):
    """detect_strains on clusters of 40-56 strains x 230-300 k k-mers with 7-11 strains present."""
    import scipy.sparse as sp
    from tests import scenarios_mid as sm
    big, arrays = {}, {}
    for name in sm.L2_BIG:
        case = sm.l2_big_case(name)
        cd = os.path.join(scratch, "l2big_" + name)
        os.makedirs(cd)
        sp.save_npz(os.path.join(cd, "X.npz"), case["X"])
        sp.save_npz(os.path.join(cd, "O.npz"), case["O"])
        pickle.dump(case["ids"], open(os.path.join(cd, "ids.pkl"), "wb"))
        captured.clear()
        out, err, _ = run_captured(
            l2mod.detect_strains, os.path.join(cd, "X.npz"), case["y"].copy(), os.path.join(cd, "ids.pkl"),
            case["ksize"], case["npp25"], case["npp75"], case["npp_out"], case["cls_cov"],
            os.path.join(cd, "O.npz"), case["all_cls"], case["l2"], case["msn"], case["pmode"], case["emode"])
        ent = dict(sha256=synth.sha256_of(case["X"].indptr.tobytes(), case["X"].indices.tobytes(),
                                          case["O"].indptr.tobytes(), case["O"].indices.tobytes(), case["y"].tobytes()),
                   error=err, K=int(case["X"].shape[0]), S=int(case["X"].shape[1]))
        if out is not None:
            res, res2, scov, sval, fsrc = out
            ent.update(res=res, res2=res2, strain_cov=scov, strain_val=sval, final_src=fsrc, order=list(scov.keys()))
        if "alphas_" in captured:
            ent.update(alpha=captured["alpha"], n_rows=captured["n_rows"], p=captured["p"], n_iter=captured["n_iter_"])
            arrays[name + "_alphas"] = captured["alphas_"]
            arrays[name + "_mse_path"] = captured["mse_path_"]
            arrays[name + "_coef"] = captured["coef_"]
        big[name] = ent
        print("L2 big", name, err, ent.get("p"), ent.get("res"))
    dump_json("l2_big.json", big)
    np.savez_compressed(os.path.join(HERE, "l2_big_arrays.npz"), **arrays)


def built_section(scratch, identify, identify_low_mem, identify_low_depth, sc, synth):
    """A Tree_database written by the reference's OWN builder (library/Build_tree.py build_tree, in a child process: its sets
    iterate in hash order, so PYTHONHASHSEED is fixed; Bio / bidict / treelib from tests/golden/_standin) on the seeded genomes of
    tests/scenarios_built.py -> the committed fixture built_tree_db.tar.gz; then the reference's identify modules on reads of
    those genomes -> built_l1.json."""
    from tests import scenarios_built as sb
    root = os.path.join(scratch, "built")
    os.makedirs(root)
    mpath, cpath = sb.write_builder_inputs(root)
    tdir = os.path.join(root, "Tree_database")
    os.makedirs(tdir)
    drv = ("import sys, random, numpy; sys.path[:0] = [%r, %r, %r, %r]; random.seed(4321); numpy.random.seed(4321); import Build_tree; "
           "Build_tree.build_tree([%r, %r, %r, 31, %r])"
           % (REPO, os.path.join(HERE, "_standin"), os.path.join(REPO, "oracle", "_ref"), os.path.join(scratch, "ref", "library"),
              mpath, cpath, tdir, sb.BUILT_PARAMS))
    r = subprocess.run([sys.executable, "-c", drv], cwd=root, env=dict(os.environ, PYTHONHASHSEED="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        raise SystemExit("the reference's builder failed:\n" + r.stderr.decode()[-2000:])
    blob = sb.pack_tree_database(tdir)
    with open(os.path.join(HERE, "built_tree_db.tar.gz"), "wb") as f:
        f.write(blob)
    print("wrote built_tree_db.tar.gz", len(blob))
    kfa = open(os.path.join(tdir, "kmer.fa"), "rb").read()
    n_rows = kfa.count(b"\n") // 2
    out = dict(builder=dict(params=sb.BUILT_PARAMS, tree_structure=open(os.path.join(tdir, "tree_structure.txt")).read(),
                            node_length=open(os.path.join(tdir, "node_length.txt")).read(),
                            reconstructed_nodes=open(os.path.join(tdir, "reconstructed_nodes.txt")).read(),
                            kmer_fa_sha256=synth.sha256_of(kfa), n_rows=n_rows, fixture_sha256=synth.sha256_of(blob)), samples={})
    for sname in sb.BUILT_SAMPLES:
        reads = sb.built_reads(sname)
        fq = os.path.join(root, sname + ".fq")
        open(fq, "wb").write(reads)
        entry = dict(sha256=synth.sha256_of(kfa, reads), n_reads=reads.count(b"\n") // 4, runs=[])
        mr = identify.jellyfish_count((fq, ""), tdir)
        cnt = np.zeros(n_rows, np.int64)
        for k_, v_ in mr.items():
            cnt[k_] = v_
        entry["counts_sha256"] = synth.sha256_of(cnt.astype(np.uint32).tobytes())
        entry["counts_sum"] = int(cnt.sum())
        entry["n_valid"] = len(mr)
        for cut in sc.CUTOFFS:
            for modname, mod in (("identify", identify), ("identify_low_mem", identify_low_mem)):
                np.random.seed(sc.POISSON_SEED)
                res, err, txt = run_captured(mod.identify_cluster, (fq, ""), tdir, list(cut))
                entry["runs"].append(dict(module=modname, cutoff=cut, error=err,
                                          result=None if res is None else {int(k): dict(v) for k, v in res.items()},
                                          trace=parse_trace(txt)))
        res, err, txt = run_captured(identify_low_depth.identify_ranks, (fq, ""), tdir)
        entry["ranks"] = dict(error=err, result=None if res is None else [[int(a), float(b)] for a, b in res])
        out["samples"][sname] = entry
        print("built L1", sname, [(r_["module"], r_["cutoff"][0], r_["error"], sorted((r_["result"] or {}).keys())) for r_ in entry["runs"]][:4])
    dump_json("built_l1.json", out)


def l2_k25_section(scratch, vote, sc, synth):
    """vote_strain_L2_batch with ksize = 25 (`-k 25`) on clusters whose k-mer sets are 25-mers: the reference's report files."""
    dbb, reads = sc.l2_k25_inputs(scratch)
    fq = os.path.join(scratch, "l2_k25.fq")
    open(fq, "wb").write(reads)
    outdir = os.path.join(scratch, "l2k25_out")
    os.makedirs(outdir)
    _, err, _ = run_captured(vote.vote_strain_L2_batch, fq, "", dbb, outdir, 25, {k: dict(v) for k, v in sc.L2_K25_RES.items()}, 0, 40, 0, 0)
    files = {}
    for r_, _, fs in os.walk(outdir):
        for f_ in fs:
            p_ = os.path.join(r_, f_)
            files[os.path.relpath(p_, outdir)] = open(p_).read()
    dump_json("l2_k25.json", dict(sha256=synth.sha256_of(reads), error=err, files=files))
    print("L2 k25", err, sorted(files))


def main():
    from tests import scenarios as sc
    from tests import synth

    scratch = tempfile.mkdtemp(prefix="ss_golden_")
    lib, jf = setup_reference(scratch)
    os.chdir(scratch)  # the reference writes temp_<uuid>.* into the CWD
    import identify
    import identify_low_mem
    import identify_low_depth
    import identify_strains_L2_Enet_Pscan_new_sp as l2mod
    import Vote_Strain_L2_Lasso_new_sp as vote

    only = sys.argv[1] if len(sys.argv) > 1 else "all"          # all | mid (round 6's sections alone: ~4 min) | built (the reference-built database alone)
    if only == "mid":
        import scipy.sparse as sp
        captured = {}
        _patch_sklearn(l2mod, captured)
        mid_section(scratch, identify, identify_low_mem, identify_low_depth, sc, synth)
        l2_big_section(scratch, l2mod, captured, synth)
        built_section(scratch, identify, identify_low_mem, identify_low_depth, sc, synth)
        l2_k25_section(scratch, vote, sc, synth)
        shutil.rmtree(scratch, ignore_errors=True)
        return
    if only == "built":
        built_section(scratch, identify, identify_low_mem, identify_low_depth, sc, synth)
        shutil.rmtree(scratch, ignore_errors=True)
        return
    if only == "k25":
        l2_k25_section(scratch, vote, sc, synth)
        shutil.rmtree(scratch, ignore_errors=True)
        return

    # ---------------------------------------------------------------- F1: raw counts
    c = sc.f1_case()
    d = os.path.join(scratch, "f1")
    os.makedirs(d)
    open(os.path.join(d, "kmer.fa"), "wb").write(c["kmer_fa"])
    paths = []
    for i, r in enumerate(c["reads"]):
        p = os.path.join(d, "reads%d.%s" % (i, "fq" if r[:1] == b"@" else "fa"))
        open(p, "wb").write(r)
        paths.append(p)
    subprocess.check_call([jf, "count", "-m", "31", "-s", "1M", "-t", "2", "--if", os.path.join(d, "kmer.fa"),
                           "-o", os.path.join(d, "t.jf")] + paths)
    dump = subprocess.check_output([jf, "dump", "-c", os.path.join(d, "t.jf")]).decode().split("\n")
    dump = sorted([ln.split() for ln in dump if ln])
    mr = identify.jellyfish_count((paths[0], paths[1]), d)
    mr_lm, err_lm, _ = run_captured(identify_low_mem.jellyfish_count, " ".join(paths), d)
    c2 = sc.f1_case(lower_only=True)
    d2 = os.path.join(scratch, "f1b")
    os.makedirs(d2)
    open(os.path.join(d2, "kmer.fa"), "wb").write(c2["kmer_fa"])
    mr2, err2, _ = run_captured(identify.jellyfish_count, (paths[0], paths[1]), d2)
    _, err2_lm, _ = run_captured(identify_low_mem.jellyfish_count, " ".join(paths), d2)
    _, err2_ld, _ = run_captured(identify_low_depth.jellyfish_count, (paths[0], paths[1]), d2)
    dump_json("f1_counts.json", dict(
        sha256=synth.sha256_of(c["kmer_fa"], *c["reads"]),
        jellyfish_dump=[[a, int(b)] for a, b in dump],
        match_results={int(k): int(v) for k, v in mr.items()},
        low_mem_error=err_lm, low_mem_match_results={int(k): int(v) for k, v in mr_lm.items()},
        lower_only=dict(sha256=synth.sha256_of(c2["kmer_fa"]), identify_error=err2,
                        match_results={int(k): int(v) for k, v in mr2.items()},
                        low_mem_error=err2_lm, low_depth_error=err2_ld),
        n_rows=c["n_rows"]))

    # ---------------------------------------------------------------- F2/F3: L1 search
    l1 = {}
    infos = {}
    for dbn in sc.L1_DBS:
        infos[dbn] = sc.build_l1(dbn, scratch)
    for sname, (dbn, mix, seed) in sc.L1_SAMPLES.items():
        info = infos[dbn]
        tdb = os.path.join(info["db_dir"], "Tree_database")
        fq = os.path.join(scratch, sname + ".fq")
        reads = sc.sample_reads(info, sname)
        open(fq, "wb").write(reads)
        kfa = open(os.path.join(tdb, "kmer.fa"), "rb").read()
        entry = dict(sha256=synth.sha256_of(kfa, reads), n_reads=reads.count(b"\n") // 4, runs=[])
        mr = identify.jellyfish_count((fq, ""), tdb)
        cnt = np.zeros(info["n_rows"], np.int64)
        for k_, v_ in mr.items():
            cnt[k_] = v_
        entry["counts_sha256"] = synth.sha256_of(cnt.astype(np.uint32).tobytes())
        entry["counts_sum"] = int(cnt.sum())
        entry["n_valid"] = len(mr)
        for cut in sc.CUTOFFS:
            for modname, mod in (("identify", identify), ("identify_low_mem", identify_low_mem)):
                np.random.seed(sc.POISSON_SEED)
                res, err, out = run_captured(mod.identify_cluster, (fq, ""), tdb, list(cut))
                entry["runs"].append(dict(module=modname, cutoff=cut, error=err,
                                          result=None if res is None else {int(k): dict(v) for k, v in res.items()},
                                          trace=parse_trace(out)))
        res, err, out = run_captured(identify_low_depth.identify_ranks, (fq, ""), tdb)
        entry["ranks"] = dict(error=err, result=None if res is None else [[int(a), float(b)] for a, b in res])
        l1[sname] = entry
        print("L1", sname, [(r["module"], r["cutoff"][0], r["error"], sorted((r["result"] or {}).keys()))
                            for r in entry["runs"]][:4])
    dump_json("l1_search.json", l1)

    # ---------------------------------------------------------------- single search() steps from a hand-made state
    steps = {}
    for stname, (sname, modname, cut, override, pend) in sc.L1_STEPS.items():
        mod = {"identify": identify, "identify_low_mem": identify_low_mem}[modname]
        tdb = os.path.join(infos[sc.L1_SAMPLES[sname][0]]["db_dir"], "Tree_database")
        fq = os.path.join(scratch, sname + ".fq")
        tree, _ = mod.read_tree_structure(tdb)
        for n in tree.all_nodes():
            n.data = [-1, -1, -1, -1, -1]
        mod.get_node_label(tdb, tree)
        for nid, (cat, acc) in override.items():
            tree.get_node(nid).data[0] = cat
            tree.get_node(nid).data[1] = acc
        mr = mod.jellyfish_count((fq, ""), tdb)
        pending = [[tree.get_node(i) for i in g] for g in pend]
        length, cov, abundance, res_temp, qp = {}, {}, {}, [], []
        args = [pending, mr, tdb, set(mr.keys()), length, cov, abundance, cut[0], cut[2], [], tree.leaves(), res_temp, tree,
                {}]
        if modname == "identify":
            args.append(qp)
        _, err, out = run_captured(mod.search, *args)
        steps[stname] = dict(
            error=err, pending=[[n.identifier for n in g] for g in pending], res_temp=[n.identifier for n in res_temp],
            qualified_parents=[n.identifier for n in qp], data={n.identifier: list(n.data) for n in tree.all_nodes()},
            length={n.identifier: v for n, v in length.items()}, cov={n.identifier: v for n, v in cov.items()},
            abundance={n.identifier: v for n, v in abundance.items()}, stdout=out.splitlines())
        print("step", stname, err, steps[stname]["pending"], steps[stname]["res_temp"])
    dump_json("l1_search_steps.json", steps)

    # ---------------------------------------------------------------- F4/F5: detect_strains
    import scipy.sparse as sp
    captured = {}
    _patch_sklearn(l2mod, captured)
    l2 = {}
    arrays = {}
    for name in sc.L2_CASES:
        case = sc.l2_case(name)
        cd = os.path.join(scratch, "l2_" + name)
        os.makedirs(cd)
        sp.save_npz(os.path.join(cd, "X.npz"), case["X"])
        sp.save_npz(os.path.join(cd, "O.npz"), case["O"])
        pickle.dump(case["ids"], open(os.path.join(cd, "ids.pkl"), "wb"))
        captured.clear()
        out, err, _ = run_captured(
            l2mod.detect_strains, os.path.join(cd, "X.npz"), case["y"].copy(), os.path.join(cd, "ids.pkl"),
            case["ksize"], case["npp25"], case["npp75"], case["npp_out"], case["cls_cov"],
            os.path.join(cd, "O.npz"), case["all_cls"], case["l2"], case["msn"], case["pmode"], case["emode"])
        ent = dict(sha256=synth.sha256_of(case["X"].toarray().tobytes(), case["O"].toarray().tobytes(),
                                          case["y"].tobytes()), error=err)
        if out is not None:
            res, res2, scov, sval, fsrc = out
            ent.update(res=res, res2=res2, strain_cov=scov, strain_val=sval, final_src=fsrc)
        if "alphas_" in captured:
            ent.update(alpha=captured["alpha"], n_rows=captured["n_rows"], p=captured["p"],
                       n_iter=captured["n_iter_"])
            arrays[name + "_alphas"] = captured["alphas_"]
            arrays[name + "_mse_path"] = captured["mse_path_"]
            arrays[name + "_coef"] = captured["coef_"]
        l2[name] = ent
        print("L2", name, err, ent.get("res"))
    dump_json("l2_detect.json", l2)
    np.savez_compressed(os.path.join(HERE, "l2_enet_arrays.npz"), **arrays)

    # ---------------------------------------------------------------- F6: ShuffleSplit
    from sklearn.model_selection import ShuffleSplit
    ss = {}
    for n in (10, 11, 1000):
        cv = ShuffleSplit(n_splits=20, test_size=0.5, random_state=0)
        folds = [(tr.tolist(), te.tolist()) for tr, te in cv.split(np.zeros((n, 1)))]
        ss[str(n)] = dict(train0=folds[0][0], test0=folds[0][1], train19=folds[19][0][:50],
                          test19=folds[19][1][:50], n_train=len(folds[0][0]), n_test=len(folds[0][1]),
                          sha256=synth.sha256_of(*[np.array(a + b, np.int64).tobytes() for a, b in folds]))
    dump_json("shuffle_split.json", ss)

    # ---------------------------------------------------------------- F7: binomial sibling test
    import scipy.stats as st
    tab = np.zeros((61, 61))
    for x in range(61):
        for y in range(61):
            tab[x, y] = 1 - st.binom.sf(max(x, y), x + y, 0.995)
    big = [[x, y, float(1 - st.binom.sf(max(x, y), x + y, 0.995))]
           for x, y in [(100, 1), (250, 3), (300, 0), (300, 7), (1000, 2), (1000, 12), (4000, 30), (77, 77)]]
    np.savez_compressed(os.path.join(HERE, "binom_table.npz"), table=tab, big=np.array(big))

    # ---------------------------------------------------------------- percentile 'nearest'
    rs = np.random.RandomState(5)
    pn = []
    for n in (1, 2, 3, 4, 5, 10, 11, 20, 21, 40, 101, 1000):
        a = rs.randint(0, 50, size=n)
        pn.append(dict(a=a.tolist(), q={str(q): int(np.percentile(a, q, interpolation="nearest"))
                                        for q in (5, 25, 50, 75, 95)}))
    dump_json("percentile_nearest.json", pn)

    # ---------------------------------------------------------------- F8: report formats
    hdr = {}
    for ex in ("GCF_003812785", "GCA_000144385_5X_GCF_008868325_5X"):
        hdr[ex] = dict(final_report=open(os.path.join(REF, "Output_Example", ex, "final_report.txt")).readline(),
                       strain_vote=open(os.path.join(REF, "Output_Example", ex, "C4", "StrainVote.report")).readline())
    dump_json("report_headers.json", hdr)

    # ---------------------------------------------------------------- end-to-end L1 -> L2 reports
    e2e = {}
    info = infos["A"]
    dbA = info["db_dir"]
    # cluster 1 (3 strains) gets an L2 k-mer set; strains share the L1 path genome
    strains = ["GCF_A1", "GCF_A2", "GCF_A3"]
    l2info = synth.build_l2_cluster(dbA, 1, 6, strains, [1500, 1200, 1000, 1400, 900],
                                    [[1, 1, 0, 0, 1], [1, 0, 1, 0, 0], [0, 1, 1, 1, 0]], seed=77,
                                    shared_with={4: [3]})
    g1 = info["leaf_genome"][1]
    mix = [(g1 + l2info["strain_extra"]["GCF_A1"], 18.0), (g1 + l2info["strain_extra"]["GCF_A3"], 7.0),
           (info["leaf_genome"][6], 9.0)]
    reads = synth.simulate_reads(mix, 301)
    fq = os.path.join(scratch, "e2e.fq")
    open(fq, "wb").write(reads)
    np.random.seed(sc.POISSON_SEED)
    cls_dict, err, out = run_captured(identify.identify_cluster, (fq, ""), os.path.join(dbA, "Tree_database"),
                                      [0.1, 0.4, 1])
    outdir = os.path.join(scratch, "e2e_out")
    os.makedirs(outdir)
    _, err2, out2 = run_captured(vote.vote_strain_L2_batch, fq, "", dbA, outdir, 31, dict(cls_dict), 0, 40, 0, 0)
    e2e["A_l2"] = dict(
        sha256=synth.sha256_of(reads), l1_error=err, l2_error=err2,
        cls_dict={int(k): dict(v) for k, v in cls_dict.items()},
        final_report=open(os.path.join(outdir, "final_report.txt")).read(),
        strain_vote=open(os.path.join(outdir, "C1", "StrainVote.report")).read())
    print(e2e["A_l2"]["final_report"])
    # all-singleton result -> generate_single_report + exit()
    reads = synth.simulate_reads([(info["leaf_genome"][6], 9.0), (info["leaf_genome"][2], 14.0)], 302)
    fq = os.path.join(scratch, "e2e_single.fq")
    open(fq, "wb").write(reads)
    cls_dict, err, out = run_captured(identify.identify_cluster, (fq, ""), os.path.join(dbA, "Tree_database"),
                                      [0.1, 0.4, 1])
    outdir = os.path.join(scratch, "e2e_single_out")
    os.makedirs(outdir)
    _, err2, out2 = run_captured(vote.vote_strain_L2_batch, fq, "", dbA, outdir, 31, dict(cls_dict), 0, 40, 0, 0)
    e2e["A_single"] = dict(sha256=synth.sha256_of(reads), l1_error=err, l2_error=err2,
                           cls_dict={int(k): dict(v) for k, v in cls_dict.items()},
                           final_report=open(os.path.join(outdir, "final_report.txt")).read())
    dump_json("e2e_reports.json", e2e)

    # ---------------------------------------------------------------- vote_strain_L2_batch on hand-made layer-1 results
    dbb, reads = sc.l2_batch_inputs(scratch)
    fq = os.path.join(scratch, "l2_batch.fq")
    open(fq, "wb").write(reads)
    batch = dict(sha256=synth.sha256_of(reads), cases={})
    for name, (res, l2_, emode) in sc.L2_BATCH_CASES.items():
        outdir = os.path.join(scratch, "l2b_" + name)
        os.makedirs(outdir)
        _, err, _ = run_captured(vote.vote_strain_L2_batch, fq, "", dbb, outdir, 31, {k: dict(v) for k, v in res.items()},
                                 l2_, 40, 0, emode)
        files = {}
        for r_, _, fs in os.walk(outdir):
            for f_ in fs:
                p_ = os.path.join(r_, f_)
                files[os.path.relpath(p_, outdir)] = open(p_).read()
        batch["cases"][name] = dict(error=err, files=files)
        print("L2 batch", name, err, sorted(files))
    dump_json("l2_batch.json", batch)

    # ---------------------------------------------------------------- round 6: configs[0]'s shape, whole flow + big clusters
    mid_section(scratch, identify, identify_low_mem, identify_low_depth, sc, synth)
    l2_big_section(scratch, l2mod, captured, synth)
    built_section(scratch, identify, identify_low_mem, identify_low_depth, sc, synth)
    l2_k25_section(scratch, vote, sc, synth)

    # ---------------------------------------------------------------- seqpy.revcomp (oracle/_ref)
    refso = os.path.join(REPO, "oracle", "_ref")
    if os.path.isdir(refso):
        sys.path.insert(0, refso)
        import seqpy
        rs = np.random.RandomState(9)
        cases = ["ACGT", "acgtn", "ACGTNRYKMSWBDHV", "", "GATTACA" * 5,
                 bytes(rs.randint(65, 91, size=64).astype(np.uint8)).decode(),
                 bytes(rs.randint(97, 123, size=64).astype(np.uint8)).decode(), "AC-GT*@`[]{}09"]
        dump_json("revcomp.json", [[s, seqpy.revcomp(s)] for s in cases])
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()

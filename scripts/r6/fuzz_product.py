#!/usr/bin/env python3
"""The HIP path against a directory of reference outputs on random scenarios (tests/golden/fuzz_reference.py gen, made in the
build container where the reference runs; the directory travels with the tree, e.g. tests/golden/_fuzz_tmp/ -- not committed):
    fuzz_product.py DIR l2      detect_core (device Pre_Scan, ElasticNetCV, refit) on every l2_<seed>.json: the pre-scan's integers
                                bit-exact, alphas to 1e-12, mse_path to 1e-7, coefficients / abundances to 1e-5, n_iter equal
    fuzz_product.py DIR l1      identify.jellyfish_count (bit-exact vs the real jellyfish), identify_cluster of both modules under the
                                recorded cutoffs (result dicts, visit order, printed lines), identify_ranks on every l1_<seed>.json
    fuzz_product.py DIR flow [WORLD]   (WORLD > 1: that many rank processes on one device over gloo, the reads sharded across them)
                                StrainScan.main with the recorded flags on every flow_<seed>.json: exceptions, the layer-1 dict and its order, every
                                report file (integer columns character for character, abundances within 1e-5)
    fuzz_product.py DIR fmt     identify.jellyfish_count on samples in random FASTA / FASTQ shapes (wrapped, CRLF, no final newline, '@' / '+' opening a
                                quality line, .gz) against the real jellyfish's counts
Prints one line per disagreement and a summary; exit code 1 on any."""
import contextlib
import io
import json
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import hostlogic as hl          # noqa: E402
from tests import scenarios as sc          # noqa: E402
from tests import scenarios_fuzz as sf     # noqa: E402
from tests import synth                    # noqa: E402

TOL = 1e-5


def _run(fn, *a, **kw):
    buf = io.StringIO()
    err = res = None
    with contextlib.redirect_stdout(buf):
        try:
            res = fn(*a, **kw)
        except BaseException as e:  # noqa: B902
            err = type(e).__name__
    return res, err, buf.getvalue()


def product_l2(g, arrs):
    """-> list of disagreements of the product's detect_core with one golden entry (also used by tests/test_fuzz_golden.py)."""
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
    seed = g["seed"]
    case = sf.l2_case(seed)
    X, O, y = case["X"], case["O"], case["y"]
    if synth.sha256_of(X.indptr.tobytes(), X.indices.tobytes(), O.indptr.tobytes(), O.indices.tobytes(), y.tobytes()) != g["sha256"]:
        return [(seed, "inputs differ")]
    trace = {}
    out, err, _ = _run(m.detect_core, X, O, case["ids"], y.copy(), case["ksize"], case["npp25"], case["npp75"], case["npp_out"], case["cls_cov"],
                       case["all_cls"], case["l2"], case["msn"], case["pmode"], case["emode"], trace=trace)
    if err != g["error"]:
        return [(seed, "error", err, g["error"])]
    if err is not None:
        return []
    res, res2, scov, sval, fsrc = out
    try:
        assert list(scov) == g["order"], ("order", list(scov), g["order"])
        assert {k: list(v) for k, v in scov.items()} == g["strain_cov"], "strain_cov"
        assert {k: float(v) for k, v in sval.items()} == {k: float(v) for k, v in g["strain_val"].items()}, "strain_val"
        for k, v in g["final_src"].items():
            assert abs(fsrc[k] - v) < 1e-12, ("final_src", k)
        assert set(res) == set(g["res"]), ("res keys", dict(res), g["res"])
        for k, v in g["res"].items():
            assert abs(float(res[k]) - float(v)) <= TOL, ("res", k, res[k], v)
            assert abs(float(res2[k]) - float(g["res2"][k])) <= TOL * max(1.0, abs(float(g["res2"][k]))), ("res2", k, res2[k], g["res2"][k])
        if arrs:
            assert trace["n_rows"] == g["n_rows"] and trace["p"] == g["p"], ("shape", trace["n_rows"], trace["p"], g["n_rows"], g["p"])
            assert np.allclose(trace["alphas_"], arrs["alphas"], rtol=1e-12, atol=0), "alphas"
            assert np.allclose(trace["mse_path_"], arrs["mse_path"], rtol=1e-7, atol=1e-9), ("mse_path", float(np.max(np.abs(trace["mse_path_"] - arrs["mse_path"]) / np.abs(arrs["mse_path"]))))
            assert abs(trace["alpha"] - g["alpha"]) <= 1e-12 * max(1.0, abs(g["alpha"])), ("alpha", trace["alpha"], g["alpha"])
            assert np.allclose(trace["coef_"], arrs["coef"], rtol=0, atol=TOL), ("coef", trace["coef_"], arrs["coef"])
            assert trace["n_iter"] == g["n_iter"], ("n_iter", trace["n_iter"], g["n_iter"])
    except AssertionError as e:
        return [(seed, str(e)[:400])]
    return []


def product_l1(g, root):
    from strainscan_amd import identify, identify_low_mem, identify_low_depth
    from strainscan_amd import db as ssdb
    bad = []
    seed = g["seed"]
    info = sf.build_l1x(seed, root) if g.get("x") else sf.build_l1(seed, root)
    rseed = 5000 + seed if g.get("x") else seed
    tdb = os.path.join(info["db_dir"], "Tree_database")
    kfa = open(os.path.join(tdb, "kmer.fa"), "rb").read()
    mods = {"identify": identify, "identify_low_mem": identify_low_mem}
    for which, ent in enumerate(g["samples"]):
        reads = sf.l1_reads(info, rseed, which)
        if synth.sha256_of(kfa, reads) != ent["sha256"]:
            bad.append((seed, which, "inputs differ"))
            continue
        fq = os.path.join(root, "f%d_s%d.fq" % (seed, which))
        open(fq, "wb").write(reads)
        mr = identify.jellyfish_count((fq, ""), tdb)
        cnt = np.zeros(info["n_rows"], np.uint32)
        for k_, v_ in mr.items():
            cnt[k_] = v_
        if synth.sha256_of(cnt.tobytes()) != ent["counts_sha256"] or len(mr) != ent["n_valid"]:
            bad.append((seed, which, "counts differ from jellyfish's"))
        for run in ent["runs"]:
            np.random.seed(sc.POISSON_SEED)
            res, err, text = _run(mods[run["module"]].identify_cluster, (fq, ""), tdb, list(run["cutoff"]))
            tag = (seed, which, run["module"], run["cutoff"])
            if err != run["error"]:
                bad.append((tag, "error", err, run["error"], text[-200:]))
                continue
            try:
                if err is None:
                    hl.assert_result_equal(res, run["result"], tag)
                got_tr = hl.parse_trace(text)
                assert [t[0] for t in got_tr] == [t[0] for t in run["trace"]], "visit order"
                for a, w in zip(got_tr, run["trace"]):
                    assert len(a) == len(w), (a, w)
                    if len(w) == 4:
                        assert abs(a[1] - w[1]) < 2e-6 and abs(a[2] - w[2]) < 2e-6 and a[3] == w[3], (a, w)
            except AssertionError as e:
                bad.append((tag, "walk", str(e)[:300]))
        res, err, _ = _run(identify_low_depth.identify_ranks, (fq, ""), tdb)
        want = ent["ranks"]
        if err != want["error"]:
            bad.append((seed, which, "ranks error", err, want["error"]))
        elif err is None and ([a for a, _ in res] != [a for a, _ in want["result"]] or
                              any(abs(b - wb) > 1e-12 * max(1.0, abs(wb)) for (_, b), (_, wb) in zip(res, want["result"]))):
            bad.append((seed, which, "ranks"))
        os.unlink(fq)
    ssdb.clear_cache()
    shutil.rmtree(info["db_dir"], ignore_errors=True)
    return bad


def product_fmt(g, root):
    """identify.jellyfish_count on a sample in random FASTA / FASTQ shapes against the real jellyfish's counts."""
    from strainscan_amd import identify
    from strainscan_amd import db as ssdb
    seed = g["seed"]
    info, paths, blobs, kinds = sf.fmt_case(seed, root)
    try:
        tdb = os.path.join(info["db_dir"], "Tree_database")
        if synth.sha256_of(open(os.path.join(tdb, "kmer.fa"), "rb").read(), *blobs) != g["sha256"]:
            return [(seed, "inputs differ")]
        mr, err, _ = _run(identify.jellyfish_count, (paths[0], paths[1] if len(paths) > 1 else ""), tdb)
        if err != g["error"]:
            return [(seed, kinds, "error", err, g["error"])]
        if err is None:
            cnt = np.zeros(info["n_rows"], np.uint32)
            for k_, v_ in mr.items():
                cnt[k_] = v_
            if synth.sha256_of(cnt.tobytes()) != g["counts_sha256"] or len(mr) != g["n_valid"]:
                return [(seed, kinds, "counts differ from jellyfish's", int(cnt.sum()), g["counts_sum"], len(mr), g["n_valid"])]
        return []
    finally:
        ssdb.clear_cache()
        for p_ in paths:
            os.unlink(p_)
        shutil.rmtree(info["db_dir"], ignore_errors=True)


def _cmp_report(got, want, float_cols):
    gl, wl = got.strip().split("\n"), want.strip().split("\n")
    assert gl[0] == wl[0] and len(gl) == len(wl), (got, want)
    for a, b in zip(gl[1:], wl[1:]):
        fa, fb = a.split("\t"), b.split("\t")
        assert len(fa) == len(fb), (a, b)
        for i, (x, y) in enumerate(zip(fa, fb)):
            if i in float_cols and x != y:
                assert abs(float(x) - float(y)) <= TOL * max(1.0, abs(float(y))), (i, a, b)
            else:
                assert x == y, (i, a, b)


def _barrier():
    from strainscan_amd import dist as sdist
    if sdist.is_distributed():
        import torch.distributed as td
        td.barrier()


def product_flow(g, root):
    """`strainscan -i reads.fq -d DB -o OUT [flags]` (StrainScan.main) against the reference's own StrainScan.py on the same bytes: the
    exception where the reference dies, the printed layer-1 dict (order included), every file of the output directory.
    Under a process group (fuzz_product.py DIR flow WORLD: several ranks on one device over gloo) every rank runs the command on its
    share of the reads, rank 0 writes the scenario, owns the output directory and compares; the others return []."""
    import ast
    from strainscan_amd import StrainScan
    from strainscan_amd import db as ssdb
    from strainscan_amd import dist as sdist
    rank, _ = sdist.rank_world()
    seed = g["seed"]
    db_dir = os.path.join(root, "DB_W%d" % seed)
    if rank == 0:
        info = sf.build_flow(seed, root)
        paths, parts = sf.flow_inputs(info, seed, root)
        ok = synth.sha256_of(open(os.path.join(db_dir, "Tree_database", "kmer.fa"), "rb").read(), b"".join(parts)) == g["sha256"]
        json.dump(dict(paths=paths, ok=ok), open(os.path.join(root, "w%d.json" % seed), "w"))
    _barrier()
    meta = json.load(open(os.path.join(root, "w%d.json" % seed)))
    paths = meta["paths"]
    try:
        if not meta["ok"]:
            return [(seed, "inputs differ")] if rank == 0 else []
        out = os.path.join(root, "out_%d" % seed)
        ssdb.clear_cache()
        np.random.seed(sc.POISSON_SEED)
        _, err, text = _run(StrainScan.main, ["-i", paths[0]] + (["-j", paths[1]] if len(paths) > 1 else []) + ["-d", db_dir, "-o", out] + list(g["argv"]))
        if os.environ.get("SS_FUZZ_TWICE"):                             # ... and once more, as a second process against the same database would:
            _barrier()                                                  # the tree, index and cluster images come from the image cache now
            if rank == 0:
                shutil.rmtree(out, ignore_errors=True)
            _barrier()
            ssdb.wait_cache_writes()
            ssdb.clear_cache()
            np.random.seed(sc.POISSON_SEED)
            _, err, text = _run(StrainScan.main, ["-i", paths[0]] + (["-j", paths[1]] if len(paths) > 1 else []) + ["-d", db_dir, "-o", out] + list(g["argv"]))
        if os.environ.get("SS_FUZZ_VERBOSE"):
            print("rank", rank, "seed", seed, "left main with", err, repr(text[-160:]), flush=True)
        _barrier()
        if rank != 0:
            return []
        if err == "SystemExit":
            err = None                                                  # (the reference's exit() after generate_single_report: return code 0 there)
        if err != g["error"]:
            return [(seed, "error", err, g["error"], text[-300:])]
        try:
            line = [ln for ln in text.splitlines() if ln.startswith("defaultdict(") or ln.startswith("{")]
            if g["cls_dict"] is not None:
                got = ast.literal_eval(line[-1][line[-1].index("{"):].rstrip(")"))
                want = ast.literal_eval(g["cls_dict"])
                assert list(got) == list(want), ("layer-1 order", list(got), list(want))
                hl.assert_result_equal(got, want, seed)
            assert [t[0] for t in hl.parse_trace(text)] == [t[0] for t in g["trace"]], "visit order"
            files = {}
            for r_, _, fs in os.walk(out):
                for f_ in fs:
                    files[os.path.relpath(os.path.join(r_, f_), out)] = open(os.path.join(r_, f_)).read()
            assert sorted(files) == sorted(g["files"]), ("files", sorted(files), sorted(g["files"]))
            n_cls = len(ast.literal_eval(g["cls_dict"])) if g["cls_dict"] else 0
            for rel, want_text in g["files"].items():
                if rel == "strain_prob.txt":
                    gl, wl = files[rel].strip().split("\n"), want_text.strip().split("\n")
                    assert gl[0] == wl[0] and len(gl) == len(wl)
                    for a, b in zip(gl[1:], wl[1:]):
                        fa, fb = a.split("\t"), b.split("\t")
                        assert fa[0] == fb[0] and fa[2:] == fb[2:] and abs(float(fa[1]) - float(fb[1])) <= 1e-12 * max(1.0, float(fb[1])), (a, b)
                elif rel == "final_report.txt" and n_cls > 1:
                    _cmp_report(files[rel], want_text, float_cols=(3, 4, 5, 6))
                else:
                    _cmp_report(files[rel], want_text, float_cols=(3, 4, 5, 6, 8, 9))
        except AssertionError as e:
            return [(seed, str(e)[:500])]
        return []
    finally:
        ssdb.clear_cache()
        _barrier()
        if rank == 0:
            for p_ in paths + [os.path.join(root, "w%d.json" % seed)]:
                os.unlink(p_)
            shutil.rmtree(db_dir, ignore_errors=True)
            shutil.rmtree(os.path.join(root, "out_%d" % seed), ignore_errors=True)


def _spawn_ranks(d, kind, world):
    """`fuzz_product.py DIR flow WORLD`: WORLD rank processes of this script on device 0, gloo over a file store (as tests/test_dist_gpu.py
    does: RCCL refuses two ranks on one device), one shared scratch directory."""
    import subprocess
    root = tempfile.mkdtemp(prefix="ss_fuzzp_")
    env = dict(os.environ, WORLD_SIZE=str(world), LOCAL_RANK="0", SS_FUZZ_ROOT=root, SS_TEST_STORE=os.path.join(root, "store"),
               SS_IMAGE_CACHE=os.path.join(root, "cache"))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), d, kind], env=dict(env, RANK=str(r))) for r in range(world)]
    rcs = [p.wait() for p in procs]
    shutil.rmtree(root, ignore_errors=True)
    sys.exit(max(rcs))


def main():
    d, kind = sys.argv[1], sys.argv[2]
    if len(sys.argv) > 3 and int(sys.argv[3]) > 1 and "RANK" not in os.environ:
        _spawn_ranks(d, kind, int(sys.argv[3]))
    files = sorted(f for f in os.listdir(d) if f.startswith(kind + "_") and f.endswith(".json"))      # (l1x_*: kind l1x, read by product_l1)
    rank = 0
    if "SS_FUZZ_ROOT" in os.environ:                                    # a rank of _spawn_ranks
        import torch
        import torch.distributed as td
        rank = int(os.environ["RANK"])
        torch.cuda.set_device(0)
        td.init_process_group("gloo", init_method="file://" + os.environ["SS_TEST_STORE"], rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
        root = os.environ["SS_FUZZ_ROOT"]
    else:
        root = tempfile.mkdtemp(prefix="ss_fuzzp_")
        os.environ.setdefault("SS_IMAGE_CACHE", os.path.join(root, "cache"))
    sf_ = sf
    n_bad = n_known = 0
    for f in files:
        g = json.load(open(os.path.join(d, f)))
        if kind == "l2":
            p = os.path.join(d, f[:-5] + ".npz")
            bad = product_l2(g, dict(np.load(p)) if os.path.exists(p) else None)
        elif kind == "fmt":
            if sf_.fmt_known_deviation(g["kinds"]):
                n_known += 1
                continue
            bad = product_fmt(g, root)
        elif kind == "flow":
            if sf_.flow_known_deviation(g["seed"], g["memory_db"]):
                n_known += 1
                continue
            if os.environ.get("SS_FUZZ_VERBOSE"):                      # which seed a hang belongs to, and where every rank stands in it
                import faulthandler
                faulthandler.dump_traceback_later(float(os.environ["SS_FUZZ_VERBOSE"]), exit=True)
                if rank == 0:
                    print("seed", g["seed"], g["argv"], g["error"], flush=True)
            bad = product_flow(g, root)
        else:
            bad = product_l1(g, root)
        for b in bad:
            print("DISAGREES", b, flush=True)
        n_bad += bool(bad)
    if "SS_FUZZ_ROOT" in os.environ:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()
        if rank != 0:
            sys.exit(0)
    else:
        shutil.rmtree(root, ignore_errors=True)
    print("fuzz_product %s%s: %d seeds, %d with a disagreement%s" % (kind, " (%s ranks)" % os.environ["WORLD_SIZE"] if "SS_FUZZ_ROOT" in os.environ else "",
          len(files), n_bad, ", %d skipped (scenarios_fuzz.flow_known_deviation / fmt_known_deviation)" % n_known if n_known else ""), flush=True)
    sys.exit(1 if n_bad else 0)


if __name__ == "__main__":
    main()

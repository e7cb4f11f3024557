#!/usr/bin/env python3
"""The whole `strainscan` run with layer 2 on the path, at BASELINE.json configs[3] shape -- what a user waits for.

Writes, in the reference's on-disk formats, a 202-cluster tree database (Tree_database/: 403 nodes) whose clusters hold 1627
strains in all, three of them multi-strain clusters WITH their layer-2 k-mer sets (Kmer_Sets_L2/Kmer_Sets/C<id>/:
all_kmer.fasta, all_strains_re.npz, id2strain_re.pkl, overlap_matrix.npz; default K x S = 5 M x 300, 2 M x 120, 1 M x 60),
and a paired FASTQ sample (text, or a .gz pair) whose reads come from two or three strains of each of those clusters and
two single-strain clusters.  Then runs the CLI the way a user does (StrainScan.py:196-271 -> Vote_...:247-311, 334-438):

  fresh process, nothing cached   (tree image + three cluster indexes + three cluster matrix images are built and written)
  fresh process, images cached    (x2)
  in this process, everything warm (database images on the device, reads resident)

and says where the time goes (SS_CLI_TRACE milestones: interpreter + imports, tree image || read ingest, tree scan + walk,
cluster scans, per-cluster solve, reports).  The reference itself re-runs jellyfish over ALL reads once per identified
cluster and solves the clusters one after the other.

    bench_cli_l2.py [n_reads = 20000000] [clusters = 5000000x300,2000000x120,1000000x60] [text|gz[-LEVEL]] [leaves = 202]

all_kid.pkl (a 5 M-entry dict k-mer -> id that the reference unpickles, Vote_...:348) is NOT written: this implementation never
reads it (row r of all_kmer.fasta is k-mer id r + 1, Build_kmer_sets_..._sp.py:397-399,409-410)."""
import json
import os
import pickle
import re
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from scripts import bench_cli  # noqa: E402

K = 31
NSEG = 64


def make_cluster(torch, dev, cdir, cid, n_clusters, n_rows, S, seed, prefix):
    """One layer-2 cluster directory: a pan-genome of NSEG segments (random sequences, n_rows / 2 sites in all -> n_rows k-mers
    with both orientations), strain s carries segment g with probability 0.35.  -> (segment code tensors, presence bool[S, NSEG],
    strain names).  Files as the builder writes them (Build_kmer_sets_..._sp.py:397-410, Recls_withR_new.py:110-115,
    Build_overlap_matrix_sp.py:89-98); the .npz files in scipy.sparse.save_npz's layout, stored (not deflated: 2.7 GB for the
    largest)."""
    os.makedirs(cdir)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rs = np.random.RandomState(seed)
    pres = rs.random_sample((S, NSEG)) < 0.35
    sites = n_rows // 2
    seg_len = np.full(NSEG, sites // NSEG, np.int64)
    seg_len[: sites - int(seg_len.sum())] += 1
    segs = [torch.randint(0, 4, (int(n) + K - 1,), generator=g, device=dev, dtype=torch.uint8) for n in seg_len]
    names = ["%s_%03d" % (prefix, i) for i in range(S)]
    # all_kmer.fasta: per site the forward k-mer then its reverse complement, ids 1..n_rows, fixed-width id lines
    asc = torch.tensor([65, 67, 84, 71], dtype=torch.uint8, device=dev)
    ar = torch.arange(K, device=dev)
    row_seg = []
    with open(os.path.join(cdir, "all_kmer.fasta"), "wb") as f:
        rid = 1
        for gi, codes in enumerate(segs):
            n = int(seg_len[gi])
            for a in range(0, n, 1 << 20):
                m = min(1 << 20, n - a)
                win = codes[(torch.arange(a, a + m, device=dev)[:, None] + ar[None, :])]
                both = torch.stack([win, win.flip(1) ^ 2], 1).reshape(2 * m, K)
                ids = np.char.zfill(np.arange(rid, rid + 2 * m).astype(str), 9)
                hdr = np.frombuffer(("".join(">" + x + "\n" for x in ids)).encode(), np.uint8).reshape(2 * m, 11)
                rows = torch.empty((2 * m, K + 12), dtype=torch.uint8, device=dev)
                rows[:, :11] = torch.from_numpy(hdr.copy()).to(dev)
                rows[:, 11:11 + K] = asc[both.long()]
                rows[:, 11 + K] = 10
                f.write(rows.cpu().numpy().tobytes())
                rid += 2 * m
            row_seg.append(np.full(2 * n, gi, np.int32))
    row_seg = np.concatenate(row_seg)
    n_rows = int(row_seg.size)
    # all_strains_re.npz: CSR int8 n_rows x S, row r has the strains that carry r's segment
    cols_of = [np.nonzero(pres[:, gi])[0].astype(np.int32) for gi in range(NSEG)]
    cnt = np.array([c.size for c in cols_of], np.int64)
    indptr = np.concatenate([[0], np.cumsum(cnt[row_seg])]).astype(np.int64)
    indices = np.empty(int(indptr[-1]), np.int32)
    pos = 0
    for gi in range(NSEG):                                  # rows are segment-major: one tile per segment
        n = 2 * int(seg_len[gi])
        if cols_of[gi].size:
            indices[pos:pos + n * cols_of[gi].size] = np.tile(cols_of[gi], n)
        pos += n * cols_of[gi].size
    np.savez(os.path.join(cdir, "all_strains_re.npz"), indices=indices, indptr=indptr, format=np.array(b"csr"),
             shape=np.array([n_rows, S]), data=np.ones(indices.size, np.int8))
    # overlap_matrix.npz: every k-mer belongs to this cluster; every 16th also to the next one
    oc = np.ones(n_rows, np.int64)
    oc[::16] = 2
    optr = np.concatenate([[0], np.cumsum(oc)]).astype(np.int64)
    oidx = np.full(int(optr[-1]), cid - 1, np.int32)
    other = cid % n_clusters                                # (cluster cid + 1, 0-based column)
    sh = optr[:-1][::16]
    lo, hi = min(cid - 1, other), max(cid - 1, other)
    oidx[sh] = lo
    oidx[sh + 1] = hi
    np.savez(os.path.join(cdir, "overlap_matrix.npz"), indices=oidx, indptr=optr, format=np.array(b"csr"),
             shape=np.array([n_rows, n_clusters]), data=np.ones(oidx.size, np.int8))
    with open(os.path.join(cdir, "id2strain_re.pkl"), "wb") as f:
        pickle.dump(names, f, 2)
    return segs, pres, names, int(indices.size)


CLOCK_RE = re.compile(r"^\[cli\] (.+?)\s+([0-9.]+) s after process start$")


def run_cli(args, env):
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "strainscan_amd.StrainScan"] + args, cwd=ROOT, capture_output=True, text=True,
                       env=dict(env, SS_CLI_TRACE="1"))
    wall = time.perf_counter() - t0
    marks = []
    for ln in r.stderr.splitlines():
        m = CLOCK_RE.match(ln.strip())
        if m:
            marks.append((m.group(1).strip(), float(m.group(2))))
    # (the milestones count from the process's creation time as psutil reads it, which is off by a constant against this
    #  process's clock: only their differences are used; what comes before main() is the wall time minus the rest)
    phases, prev = {}, None
    for name, t in marks:
        if prev is not None:
            phases[name] = round(t - prev, 3)
        prev = t
    if marks:
        phases = dict({"interpreter, imports, HIP start (wall - the rest)": round(wall - (marks[-1][1] - marks[0][1]), 3)}, **phases)
    return dict(wall_s=round(wall, 3), rc=r.returncode, phases_s=phases, stderr_tail=None if r.returncode == 0 else r.stderr[-600:])


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    shapes = [tuple(int(x) for x in c.split("x")) for c in (sys.argv[2] if len(sys.argv) > 2 else "5000000x300,2000000x120,1000000x60").split(",")]
    mode = sys.argv[3] if len(sys.argv) > 3 else "text"
    C = int(sys.argv[4]) if len(sys.argv) > 4 else 202
    print(json.dumps(run(n_reads, shapes, mode, C)))


def run(n_reads, shapes, mode="text", C=202, per_cluster=True):
    """-> the dict main() prints (bench.py calls this with a reduced configuration for its `cli_e2e` block)."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    base = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "ss_cli_l2_%d" % os.getpid())
    os.makedirs(base)
    env = dict(os.environ, SS_IMAGE_CACHE=os.path.join(base, "cache"), STRAINSCAN_QUIET="1")
    os.environ.update(SS_IMAGE_CACHE=env["SS_IMAGE_CACHE"], STRAINSCAN_QUIET="1")
    out = dict(n_reads=n_reads, leaves=C, clusters=["%d x %d" % s for s in shapes], input=mode, host_cpus=os.cpu_count())
    try:
        t0 = time.perf_counter()
        spec = bench.make_db(torch, dev, C, seed=20231013, shape="sampled")
        g = torch.Generator(device=dev)
        g.manual_seed(77)
        rs = np.random.RandomState(77)
        leaves_h = rs.choice(np.arange(spec["n_nodes"] // 2, spec["n_nodes"]), size=len(shapes) + 2, replace=False)
        leaf_ids = [bench.heap_to_id(int(h), C) for h in leaves_h]
        multi, genomes, weights, expect = {}, [], [], []
        nnz_all = 0
        for i, (n_rows, S) in enumerate(shapes):
            cid = leaf_ids[i]
            cdir = os.path.join(base, "Kmer_Sets_L2", "Kmer_Sets", "C%d" % cid)
            segs, pres, names, nnz = make_cluster(torch, dev, cdir, cid, C, n_rows, S, 100 + i, "GCF_C%d" % cid)
            nnz_all += nnz
            multi[cid] = names
            core = bench.path_genome(torch, dev, spec, int(leaves_h[i]), 0.05, g)
            present = [(3 * (i + 1)) % S, (57 + i) % S] + ([(120 + i) % S] if S > 100 else [])
            for s_i, depth in zip(present, (30.0, 11.0, 5.0)):
                genomes.append(torch.cat([core] + [segs[gi] for gi in np.nonzero(pres[s_i])[0]]))
                weights.append(depth * genomes[-1].numel())
                expect.append(names[s_i])
            del segs
        for j in range(2):                                   # two single-strain clusters
            genomes.append(bench.path_genome(torch, dev, spec, int(leaves_h[len(shapes) + j]), 0.05, g))
            weights.append((12.0, 6.0)[j] * genomes[-1].numel())
            expect.append("strain_%d" % leaf_ids[len(shapes) + j])
        tdir = bench_cli.write_db(torch, dev, spec, C, base)
        bench.write_tree_files(spec, C, tdir, multi=multi)
        n_strains = sum(len(v) for v in multi.values()) + (C - len(multi))
        w = np.array(weights) / sum(weights)
        counts = (w * n_reads).astype(np.int64)
        counts[0] += n_reads - counts.sum()
        r = bench.reads_of(torch, dev, genomes, counts, g)
        half = n_reads // 2
        fq = [os.path.join(base, "s_%d.fq" % (i + 1)) for i in range(2)]
        noisy = 5 if mode.startswith("gz") else None
        bench.write_fastq(r[: half * 151], half, fq[0], noisy_quality_seed=noisy)
        bench.write_fastq(r[half * 151:], n_reads - half, fq[1], noisy_quality_seed=None if noisy is None else noisy + 1)
        out["fastq_bytes"] = sum(os.path.getsize(p) for p in fq)
        if mode.startswith("gz"):
            lvl = "-" + (mode.split("-")[1] if "-" in mode else "1")
            t1 = time.perf_counter()
            pr = [subprocess.Popen(["gzip", lvl, p]) for p in fq]
            assert all(q.wait() == 0 for q in pr)
            fq = [p + ".gz" for p in fq]
            out["gzip"] = dict(level=lvl, seconds=round(time.perf_counter() - t1, 1), gz_bytes=sum(os.path.getsize(p) for p in fq))
        depth_of = {n: round(float(c) * 150 / gn.numel(), 1) for n, c, gn in zip(expect, counts, genomes)}
        del r, genomes, spec
        torch.cuda.empty_cache()
        out.update(db_strains=n_strains, tree_nodes=2 * C - 1, l2_nonzeros=nnz_all, strains_in_sample=depth_of,
                   setup_s=round(time.perf_counter() - t0, 1))

        odir = os.path.join(base, "out")
        args = ["-i", fq[0], "-j", fq[1], "-d", base, "-o", odir]
        runs = []
        for label in ("nothing_cached", "images_cached", "images_cached_again"):
            shutil.rmtree(odir, ignore_errors=True)
            res = run_cli(args, env)
            res["label"] = label
            runs.append(res)
        out["cli_fresh_process"] = runs
        rep = os.path.join(odir, "final_report.txt")
        lines = open(rep).read().strip().split("\n") if os.path.exists(rep) else []
        found = [ln.split("\t")[1] for ln in lines[1:]]
        out["report_strains"] = found
        out["all_expected_strains_reported"] = bool(set(expect) <= set(found))
        out["cache_bytes"] = sum(os.path.getsize(os.path.join(env["SS_IMAGE_CACHE"], f_)) for f_ in os.listdir(env["SS_IMAGE_CACHE"]))

        # in this process: database images on the device and reads resident after the first call; phases of the second
        import contextlib
        import io
        from strainscan_amd import StrainScan, Vote_Strain_L2_Lasso_new_sp as vote, _lib, identify
        ph = {}
        for label in ("first_in_process", "warm"):
            shutil.rmtree(odir, ignore_errors=True)
            os.makedirs(odir)
            with contextlib.redirect_stdout(io.StringIO()):
                t1 = time.perf_counter()
                cls = identify.identify_cluster((fq[0], fq[1]), tdir, [0.1, 0.4, 1])
                t2 = time.perf_counter()
                todo = [c for c in cls if cls[c]["strain"] == 0]
                counts_ = vote.cluster_counts_many(fq[0], fq[1], [base + "/Kmer_Sets_L2/Kmer_Sets/C" + str(c) for c in todo], 31)
                t3 = time.perf_counter()
                vote.vote_strain_L2_batch(fq[0], fq[1], base, odir, 31, dict(cls), 0, 40, 0, 0)
                t4 = time.perf_counter()
            ph[label] = dict(identify_cluster_s=round(t2 - t1, 3), cluster_scans_alone_s=round(t3 - t2, 3),
                             vote_strain_L2_batch_s=round(t4 - t3, 3), total_s=round(t4 - t1, 3),
                             multi_strain_clusters=len(todo), m_reads_per_s=round(n_reads / (t4 - t1) / 1e6, 1))
            del counts_
        out["in_process"] = ph
        # one cluster at a time, warm: scan, counts to the host, y, the solve (vote_strain_L2's own steps)
        per = []
        for c in (todo if per_cluster else []):
            cd = base + "/Kmer_Sets_L2/Kmer_Sets/C" + str(c)
            t1 = time.perf_counter()
            cnt = vote.cluster_counts(fq[0], fq[1], cd, 31)
            t2 = time.perf_counter()
            py_o = vote.remove_1(cnt)
            npp = py_o[py_o != 0]
            med = float(np.median(npp)) if npp.size else float("nan")
            t3 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                item = [fq[0], cd, odir + "/C" + str(c), 31, cls[c]["cls_ab"], "C" + str(c), cls[c]["cls_cov"], list(cls.keys()), 0, 40, 0, 0, fq[1]]
                vote.vote_strain_L2(item, cnt)
            t4 = time.perf_counter()
            per.append(dict(cluster=int(c), rows=int(cnt.size), scan_and_counts_ms=round((t2 - t1) * 1e3, 1),
                            remove_1_and_median_ms=round((t3 - t2) * 1e3, 1), median=med,
                            vote_strain_L2_given_counts_ms=round((t4 - t3) * 1e3, 1)))
        out["per_cluster_warm"] = per
    finally:
        shutil.rmtree(base, ignore_errors=True)
    return out


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""identify_cluster() end to end at BASELINE.json configs[1] scale, from files on disk.

Writes a synthetic E. coli-shaped database in the reference's on-disk format (Tree_database/:
kmer.fa, kmers/<id>, tree_structure.txt, node_length.txt, hclsMap_95_recls.txt ...: 823 leaf
clusters = 1645 nodes, ~25 M rows) and a paired FASTQ sample (three-strain mix 70/20/10), then times
library/identify.identify_cluster's mirror on it: first call (kmer.fa text parse, index build,
image cache written), a second process-cold call (image cache read) and the phases of each
(database image, FASTQ ingest + scan, tree walk).  Usage: bench_cli.py [n_reads] [n_leaves] [sampled|contiguous] [gz[-LEVEL]]"""
import json
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

K = 31


heap_to_id = bench.heap_to_id


def write_db(torch, dev, spec, C, db_dir):
    """The synthetic database of bench.make_db (either shape) in the reference's on-disk format: kmer.fa rows decoded
    from the keys, kmers/<id> = the node's row numbers, the tree text files."""
    tdir = os.path.join(db_dir, "Tree_database")
    os.makedirs(os.path.join(tdir, "kmers"))
    n_nodes = spec["n_nodes"]
    row_off = spec["row_off"].astype(np.int64)
    asc = torch.tensor([65, 67, 84, 71], dtype=torch.uint8, device=dev)      # device codes 0..3 -> A C T G
    keys = torch.from_numpy(spec["keys"].view(np.int64))
    sh = 2 * torch.arange(K, device=dev, dtype=torch.int64)
    with open(os.path.join(tdir, "kmer.fa"), "wb") as f:
        for r0 in range(0, keys.numel(), 1 << 21):                           # bounded memory
            kk = keys[r0:r0 + (1 << 21)].to(dev)
            rows = torch.empty((kk.numel(), K + 4), dtype=torch.uint8, device=dev)
            rows[:, 0] = 62; rows[:, 1] = 49; rows[:, 2] = 10; rows[:, K + 3] = 10   # ">1\n" ... "\n"
            rows[:, 3:K + 3] = asc[(kk[:, None] >> sh[None, :]) & 3]
            f.write(rows.cpu().numpy().tobytes())
    bench.write_tree_files(spec, C, tdir)
    rows_np = spec["rows"]
    for h in range(n_nodes):
        with open(os.path.join(tdir, "kmers", str(heap_to_id(h, C))), "w") as f:
            f.write(" ".join(map(str, rows_np[int(row_off[h]):int(row_off[h + 1])].tolist())) + " ")
    return tdir


write_fastq = bench.write_fastq


def main():
    import torch
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 823
    shape = sys.argv[3] if len(sys.argv) > 3 else "sampled"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    base = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "ss_cli_%d" % os.getpid())
    os.makedirs(base)
    os.environ["SS_IMAGE_CACHE"] = os.path.join(base, "cache")
    os.environ["STRAINSCAN_QUIET"] = "1"
    out = dict(n_reads=n_reads, leaves=C, db_shape=shape, host_threads=os.cpu_count())
    try:
        t0 = time.perf_counter()
        spec = bench.make_db(torch, dev, C, seed=20231013, shape=shape)
        tdir = write_db(torch, dev, spec, C, base)
        out["db_rows"] = int(spec["keys"].size)
        out["db_text_bytes"] = os.path.getsize(os.path.join(tdir, "kmer.fa"))
        half = n_reads // 2
        fq = [os.path.join(base, "s_%d.fq" % (i + 1)) for i in range(2)]
        r = bench.make_reads(torch, dev, spec, n_reads, seed=2, hit_frac=0.05)
        write_fastq(r[: half * 151], half, fq[0])
        write_fastq(r[half * 151:], n_reads - half, fq[1])
        out["fastq_bytes"] = sum(os.path.getsize(p) for p in fq)
        if len(sys.argv) > 4 and sys.argv[4].startswith("gz"):         # gz[-LEVEL]: the sample as a .fastq.gz pair (gzip -1 by default)
            import subprocess
            lvl = "-" + (sys.argv[4].split("-")[1] if "-" in sys.argv[4] else "1")
            # (a constant quality line deflates to nothing: Phred values that fall off along the read, with noise)
            rs = np.random.RandomState(5)
            for p in fq:
                a = np.fromfile(p, np.uint8).reshape(-1, 307)
                q = np.clip(38 - np.abs(rs.normal(0, 4, size=(a.shape[0], 150))).astype(np.int64) - (np.arange(150) // 30), 2, 40) + 33
                a[:, 156:306] = q.astype(np.uint8)
                a.tofile(p)
                del a, q
            t1 = time.perf_counter()
            pr = [subprocess.Popen(["gzip", lvl, p]) for p in fq]
            assert all(q.wait() == 0 for q in pr)
            fq = [p + ".gz" for p in fq]
            out["gzip"] = dict(level=lvl, seconds=round(time.perf_counter() - t1, 1), gz_bytes=sum(os.path.getsize(p) for p in fq))
        del r, spec
        torch.cuda.empty_cache()
        out["setup_s"] = round(time.perf_counter() - t0, 1)

        from strainscan_amd import cst, db as ssdb, identify
        for label in ("first_call", "cached_image"):
            ssdb.clear_cache()
            ph = {}
            t0 = time.perf_counter()
            if os.environ.get("SS_PROFILE") == label:
                import cProfile, pstats
                pr = cProfile.Profile()
                img = pr.runcall(ssdb.tree_image, tdir, True)
                pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(18)
            else:
                img = ssdb.tree_image(tdir, True)
            ph["database_image_s"] = time.perf_counter() - t0
            t1 = time.perf_counter()
            img.scan(fq)
            ph["ingest_scan_s"] = time.perf_counter() - t1
            torch.cuda.synchronize()
            ph["ingest_scan_s"] = time.perf_counter() - t1
            t1 = time.perf_counter()
            if os.environ.get("SS_PROFILE") == label + "_walk":
                import cProfile, pstats
                pr = cProfile.Profile()
                res = pr.runcall(lambda: cst.Walk(cst.ImageProvider(img), tdir, [0.1, 0.4, 1], identify._PARAMS, out=lambda *a: None).run())
                pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(14)
            else:
                res = cst.Walk(cst.ImageProvider(img), tdir, [0.1, 0.4, 1], identify._PARAMS, out=lambda *a: None).run()
            ph["tree_walk_s"] = time.perf_counter() - t1
            ph["total_s"] = time.perf_counter() - t0
            ph["m_reads_per_s_total"] = n_reads / ph["total_s"] / 1e6
            ph["m_reads_per_s_ingest_scan"] = n_reads / ph["ingest_scan_s"] / 1e6
            out[label] = {k: round(v, 3) for k, v in ph.items()}
        # the public entry with the database image cached and nothing loaded in this process: the reads load on a worker
        # thread while the image does (db.prefetch_reads)
        ssdb.clear_cache()
        t0 = time.perf_counter()
        res1 = identify.identify_cluster((fq[0], fq[1]), tdir, [0.1, 0.4, 1])
        out["identify_cluster_cached_s"] = round(time.perf_counter() - t0, 3)
        assert dict(res1) == dict(res)
        out["m_reads_per_s_identify_cluster_cached"] = round(n_reads / out["identify_cluster_cached_s"] / 1e6, 1)
        # the public entry, everything warm in this process (image on the device, reads resident)
        t0 = time.perf_counter()
        res2 = identify.identify_cluster((fq[0], fq[1]), tdir, [0.1, 0.4, 1])
        out["identify_cluster_warm_s"] = round(time.perf_counter() - t0, 3)
        assert dict(res2) == dict(res)
        # what a user sees: a fresh process running the CLI on the same files (database image cached): interpreter
        # start, imports, HIP initialisation, image load, first-touch ingest, scan, walk, report
        import subprocess
        ts = []
        for _ in range(2):
            odir = os.path.join(base, "cli_out")
            shutil.rmtree(odir, ignore_errors=True)
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "strainscan_amd.StrainScan", "-i", fq[0], "-j", fq[1], "-d", base, "-o", odir],
                               cwd=ROOT, capture_output=True, text=True, env=dict(os.environ))
            ts.append(time.perf_counter() - t0)
        out["cli_fresh_process_s"] = [round(t, 3) for t in ts]
        if os.environ.get("SS_PROFILE") == "cli":
            shutil.rmtree(os.path.join(base, "cli_out"), ignore_errors=True)
            pr = subprocess.run([sys.executable, "-m", "cProfile", "-s", "cumtime", "-m", "strainscan_amd.StrainScan", "-i", fq[0], "-j", fq[1],
                                 "-d", base, "-o", os.path.join(base, "cli_out")], cwd=ROOT, capture_output=True, text=True, env=dict(os.environ))
            lines = pr.stdout.split("\n")
            k = next((i for i, ln in enumerate(lines) if "cumulative" in ln or "cumtime" in ln), 0)
            sys.stderr.write("\n".join(lines[max(0, k - 4):k + 45]) + "\n")
        rep = os.path.join(base, "cli_out", "final_report.txt")
        out["cli_report_lines"] = len(open(rep).read().strip().split("\n")) if os.path.exists(rep) else (r.stderr[-300:] or r.stdout[-300:])
        out["clusters_found"] = {int(k): dict(strain=v["strain"], cls_per=round(float(v["cls_per"]), 4),
                                              cls_cov=round(float(v["cls_cov"]), 4)) for k, v in res.items()}
    finally:
        shutil.rmtree(base, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

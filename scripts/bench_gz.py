#!/usr/bin/env python3
"""gzip'ed FASTQ -> counts: rate of the sequential (inflate) reader, one thread per file."""
import gzip, json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from scripts.bench_cli import write_fastq


def main():
    import torch
    from strainscan_amd import _lib
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    dev = torch.device("cuda", 0)
    spec = bench.make_db(torch, dev, 103, seed=20231013)
    db = _lib.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True)
    r = bench.make_reads(torch, dev, spec, n_reads, seed=2, hit_frac=0.05)
    d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    fq = [os.path.join(d, "ss_gz_%d_%d.fq" % (os.getpid(), i)) for i in range(2)]
    half = n_reads // 2
    write_fastq(r[: half * 151], half, fq[0]); write_fastq(r[half * 151:], n_reads - half, fq[1])
    # realistic qualities (a constant quality line deflates to nothing): ~40 distinct values, correlated
    rs = np.random.RandomState(1)
    for p in fq:
        a = np.fromfile(p, np.uint8).reshape(-1, 307)
        q = np.clip(38 - np.abs(rs.normal(0, 4, size=(a.shape[0], 150))).astype(np.int64) - (np.arange(150) // 30), 2, 40) + 33
        a[:, 156:306] = q.astype(np.uint8)
        a.tofile(p)
    out = dict(n_reads=n_reads)
    try:
        db.reset(); db.scan_files(fq); want = db.counts_rows().copy()
        t0 = time.perf_counter(); lvl = sys.argv[2] if len(sys.argv) > 2 else "-1"; pr = [subprocess.Popen(["gzip", "-k", lvl, p]) for p in fq]; [q.wait() for q in pr]
        out["gzip_s"] = round(time.perf_counter() - t0, 1)
        gz = [p + ".gz" for p in fq]
        out["gz_bytes"] = sum(os.path.getsize(p) for p in gz)
        out["text_bytes"] = sum(os.path.getsize(p) for p in fq)
        t0 = time.perf_counter(); subprocess.check_call("zcat %s > /dev/null" % gz[0], shell=True); out["zcat_one_file_s"] = round(time.perf_counter() - t0, 2)
        for mode, tag in (("0", "host_inflaters"), ("1", "device")):          # SS_GZ_GPU: ss_pgz.hip on the CPUs / ss_ginflate.hip + ss_fastq_dev.hip
            os.environ["SS_GZ_GPU"] = mode
            for name in ("scan_files_gz", "scan_files_gz2"):
                db.reset(); t0 = time.perf_counter(); db.scan_files(gz); t1 = time.perf_counter()
                out[tag + ":" + name] = dict(s=round(t1 - t0, 3), m_reads_per_s=round(n_reads / (t1 - t0) / 1e6, 2), same=bool(np.array_equal(db.counts_rows(), want)))
            for name in ("readset_gz", "readset_gz2"):
                t0 = time.perf_counter(); rs_ = _lib.ReadSet(gz, 0, 1); t1 = time.perf_counter()
                db.reset(); rs_.scan_into(db); torch.cuda.synchronize()
                out[tag + ":" + name] = dict(load_s=round(t1 - t0, 3), m_reads_per_s=round(n_reads / (t1 - t0) / 1e6, 2), same=bool(np.array_equal(db.counts_rows(), want)))
                rs_.close()
    finally:
        for p in fq:
            for q in (p, p + ".gz"):
                if os.path.exists(q): os.unlink(q)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

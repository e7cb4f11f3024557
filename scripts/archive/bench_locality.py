#!/usr/bin/env python3
# ARCHIVED (round 6): the SS_ORDER_BITS knob this script sweeps was removed from the library in round 5 -- the bin width is
# chosen from the data now -- so its A/B legs all measure the same configuration.  Kept as the record of how
# profiles/r03_locality_sweep.json was made.
"""Locality order of a resident read set (ss_reorder.hip): what it costs and what it buys, over the coverage of the sample.

    bench_locality.py [sampled|contiguous] [--bits 0] [--out profiles/r03_locality_sweep.json]
(the library reads SS_ORDER_BITS once per process: sweep the bin width with one run per value)

Cases (E. coli-shaped table of bench.py, 70/20/10 three-strain mix, ~5.3 Mb genomes):
    400x   20 M reads of the mix (the bench's batch: the dominant strain is covered ~400-fold)
    40x     2 M reads
    5x    250 K reads
    meta  20 M reads of which 1.25 % are the mix (dominant strain ~5x) and the rest random sequence: a metagenome in
          which the species of the database is rare -- nothing to share between reads, the order can only not hurt
For every case: scan kernel time in FILE order, then for each bin width: time to bin the block (ss_reads_from_flat_dev,
wall clock incl. allocation of the new slab) and the scan kernel time of the binned set; counts compared with the
file-order scan (must be equal).  One JSON object on stdout (and in --out)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def with_background(torch, dev, reads, n_mix, n_total, seed):
    """n_mix reads of the mix + (n_total - n_mix) reads of random sequence, shuffled."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    asc = torch.tensor([65, 67, 84, 71], dtype=torch.uint8, device=dev)
    out = torch.empty(n_total * 151, dtype=torch.uint8, device=dev)
    v = out.view(n_total, 151)
    v[:n_mix] = reads.view(n_mix, 151)
    chunk = 1 << 21
    for lo in range(n_mix, n_total, chunk):
        m = min(chunk, n_total - lo)
        v[lo:lo + m, :150] = asc[torch.randint(0, 4, (m, 150), generator=g, device=dev)]
    v[:, 150] = 10
    perm = torch.randperm(n_total, generator=g, device=dev)
    return v[perm].contiguous().view(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="?", default="sampled")
    ap.add_argument("--bits", type=int, default=0, help="bin width; 0 = the library's choice (about four records per bin, 12..22 bits)")
    ap.add_argument("--out", default="")
    ap.add_argument("--cases", default="400x,40x,5x,meta")
    args = ap.parse_args()
    if args.bits > 0:
        os.environ["SS_ORDER_BITS"] = str(args.bits)
    else:
        os.environ.pop("SS_ORDER_BITS", None)
    import torch
    from strainscan_amd import _lib
    dev = torch.device("cuda", 0)
    spec = bench.make_db(torch, dev, 823, seed=20231013, shape=args.shape, hit_frac=0.05)
    db = _lib.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True)
    stream = torch.cuda.current_stream().cuda_stream

    def scan_ms(fn):
        ts = []
        for _ in range(5):
            db.reset(stream)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts[1:]))

    cases = dict([("400x", (20_000_000, 0)), ("40x", (2_000_000, 0)), ("5x", (250_000, 0)), ("meta", (250_000, 20_000_000))])
    res = dict(db_shape=args.shape, rows=int(spec["keys"].size), cases={})
    import subprocess
    res["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True).stdout.decode().strip() or None
    for name in args.cases.split(","):
        n_mix, n_total = cases[name]
        reads = bench.make_reads(torch, dev, spec, n_mix, seed=2, hit_frac=0.05)
        if n_total:
            reads = with_background(torch, dev, reads, n_mix, n_total, 5)
        n = reads.numel() // 151
        t_file = scan_ms(lambda: db.scan_flat_dev(reads.data_ptr(), reads.numel(), stream))
        want = db.counts_rows()
        c = dict(reads=n, file_order_kernel_ms=round(t_file, 4), hits=int(want.sum()))
        prep = []
        for i in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rs = _lib.ReadSet.from_flat_dev(reads.data_ptr(), reads.numel(), order=True)
            torch.cuda.synchronize()
            prep.append((time.perf_counter() - t0) * 1e3)
            if i < 3:
                rs.close()
        t_bin = scan_ms(lambda: rs.scan_into(db, stream))
        ok = bool(np.array_equal(db.counts_rows(), want))
        rs.close()
        c.update(prepare_ms=round(float(np.median(prep[1:])), 3), binned_kernel_ms=round(t_bin, 4), counts_equal=ok,
                 speedup=round(t_file / t_bin, 3), scans_to_break_even=(round(float(np.median(prep[1:])) / (t_file - t_bin), 2) if t_file > t_bin else None))
        res["cases"][name] = c
        del reads
        torch.cuda.empty_cache()
    res["bits"] = args.bits or "adaptive (12..22: about four records per bin)"
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, "a") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()

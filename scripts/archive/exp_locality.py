#!/usr/bin/env python3
"""Experiment: scan time when the resident reads are ordered by the minimizer of their first k-mer (reads that start
within the same 17-base window of a genome become neighbours and share most of their minimizers -> page lookups hit
L2).  Usage: exp_locality.py [sampled|contiguous] [reads]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def first_minimizer(torch, reads, n):
    v = reads.view(n, 151)
    codes = ((v[:, :31] >> 1) & 3).to(torch.int64)
    best_key = None
    best_x = None
    C1, C0 = 0x4F1BB << 5, 0x7F4A7C00
    for i in range(17):
        x = torch.zeros(n, dtype=torch.int64, device=reads.device)
        for j in range(15):
            x |= codes[:, i + j] << (2 * j)
        key = (((x & 0xFFFFFF) * C1 + C0) & 0xFFFFFFFF) & ~31
        if best_key is None:
            best_key, best_x = key, x
        else:
            m = key < best_key
            best_key = torch.where(m, key, best_key)
            best_x = torch.where(m, x, best_x)
    return best_x


def main():
    import torch
    from strainscan_amd import _lib
    shape = sys.argv[1] if len(sys.argv) > 1 else "sampled"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
    dev = torch.device("cuda", 0)
    spec = bench.make_db(torch, dev, 823, seed=20231013, shape=shape, hit_frac=0.05)
    db = _lib.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True)
    reads = bench.make_reads(torch, dev, spec, n, seed=2, hit_frac=0.05)
    stream = torch.cuda.current_stream().cuda_stream

    def timeit(r):
        ts = []
        for _ in range(4):
            db.reset(stream)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); db.scan_flat_dev(r.data_ptr(), r.numel(), stream); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return min(ts[1:]), int(torch.zeros(1).item())
    t_file, _ = timeit(reads)
    c0 = db.counts_rows()
    t0 = time.perf_counter()
    x = first_minimizer(torch, reads, n)
    order = torch.argsort(x)
    sorted_reads = reads.view(n, 151)[order].contiguous().view(-1)
    torch.cuda.synchronize()
    t_sort = time.perf_counter() - t0
    t_loc, _ = timeit(sorted_reads)
    c1 = db.counts_rows()
    print("shape %s reads %d: file order %.3f ms, locality order %.3f ms (torch sort %.2f s), counts equal %s" % (
        shape, n, t_file, t_loc, t_sort, bool(np.array_equal(c0, c1))))


if __name__ == "__main__":
    main()

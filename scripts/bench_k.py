#!/usr/bin/env python3
"""Cluster-table scans at k != 31 (`-k`, StrainScan.py:136,266-271; Vote_Strain_L2_Lasso_new_sp.py:359-371): the minimizer-paged
index (ss_mini.hip) serves 17 <= k <= 31 (k = 31 through the tuned kernel, the others through scan_minik_kernel; round 6), the flat
open-address table (ss_scan.hip) what is below -- and every k under SS_LAYOUT=flat (the A/B leg).
    bench_k.py [rows] [reads]      -> one JSON line per k: index build time from a k-mer FASTA, scan kernel time, reads/s
Table: `rows` k-mers cut from random genomes (every 20th position, both orientations as the builder writes them);
reads: 150 bp from the same genomes (5 % of their k-mers are table k-mers), resident flat block."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
    import torch
    from strainscan_amd import _lib
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    asc = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    comp = torch.tensor([3, 2, 1, 0], device=dev)
    genome_len = rows // 2 * 20 + 64
    genome = torch.randint(0, 4, (genome_len,), generator=g, device=dev)
    out = []
    for k in [int(x) for x in os.environ.get("BENCH_K_LIST", "31,27,25,21,17").split(",")]:
        starts = torch.arange(0, rows // 2, device=dev) * 20
        idx = starts[:, None] + torch.arange(k, device=dev)[None, :]
        fw = genome[idx]
        rc = comp[fw.flip(1)]
        both = torch.stack([fw, rc], 1).reshape(-1, k)
        fa = torch.empty((both.shape[0], k + 4), dtype=torch.uint8, device=dev)
        fa[:, 0] = 62; fa[:, 1] = 49; fa[:, 2] = 10; fa[:, 3:3 + k] = asc[both]; fa[:, 3 + k] = 10
        kfa = fa.cpu().numpy().tobytes()
        st = torch.randint(0, genome_len - 150, (n_reads,), generator=g, device=dev)
        reads = torch.empty((n_reads, 151), dtype=torch.uint8, device=dev)
        chunk = 1 << 20
        ar = torch.arange(150, device=dev)
        for lo in range(0, n_reads, chunk):
            s = st[lo:lo + chunk]
            c = genome[s[:, None] + ar[None, :]]
            rev = torch.rand((s.numel(),), generator=g, device=dev) < 0.5
            c = torch.where(rev[:, None], comp[c.flip(1)], c)
            reads[lo:lo + chunk, :150] = asc[c]
        reads[:, 150] = 10
        flat = reads.view(-1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        db = _lib.KmerDB.from_text(kfa, k, True)
        build_s = time.perf_counter() - t0
        stream = torch.cuda.current_stream().cuda_stream
        ts = []
        for _ in range(4):
            db.reset(stream)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            db.scan_flat_dev(flat.data_ptr(), flat.numel(), stream)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        hits = int(db.counts_rows().astype(np.int64).sum())
        info = db.info()
        ms = float(np.median(ts[1:]))
        out.append(dict(k=k, rows=int(both.shape[0]), reads=n_reads, layout="minimizer pages" if info["layout"] == 1 else "flat table",
                        index_build_s=round(build_s, 3), scan_kernel_ms=round(ms, 3), m_reads_per_s=round(n_reads / ms / 1e3, 1),
                        algorithmic_gb_s=round(n_reads * (150 + (150 - k + 1) * 8) / ms / 1e6, 1), hits=hits,
                        device_mb=round(info["device_bytes"] / 1e6, 1)))
        db.close()
        del fa, both, reads
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""`-k` where it is used: a layer-2 CLUSTER table (every k-mer of a genome, both orientations -- Vote_Strain_L2_Lasso_new_sp.py:354-372
scans all reads against all_kmer.fasta of each identified cluster) and reads OF that genome: nearly every read k-mer hits.
    k_cluster.py [genome = 2500000] [reads = 8000000]   -> one JSON line per k: scan kernel ms on the page index and on the flat table
(SS_LAYOUT is read when a table is built: the script builds one table per layout in turn; resident binned reads, expect_hits as
the product sets it)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
    import torch
    from strainscan_amd import _lib
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    asc = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    comp = torch.tensor([3, 2, 1, 0], device=dev)
    genome = torch.randint(0, 4, (G + 64,), generator=g, device=dev)
    st = torch.randint(0, G - 150, (n_reads,), generator=g, device=dev)
    reads = torch.empty((n_reads, 151), dtype=torch.uint8, device=dev)
    ar = torch.arange(150, device=dev)
    for lo in range(0, n_reads, 1 << 20):
        s = st[lo:lo + (1 << 20)]
        c = genome[s[:, None] + ar[None, :]]
        err = torch.rand(c.shape, generator=g, device=dev) < 0.005
        c = torch.where(err, (c + 1 + torch.randint(0, 3, c.shape, generator=g, device=dev)) % 4, c)
        rev = torch.rand((s.numel(),), generator=g, device=dev) < 0.5
        c = torch.where(rev[:, None], comp[c.flip(1)], c)
        reads[lo:lo + (1 << 20), :150] = asc[c]
    reads[:, 150] = 10
    flat = reads.view(-1)
    stream = torch.cuda.current_stream().cuda_stream
    rset = _lib.ReadSet.from_flat_dev(flat.data_ptr(), flat.numel(), order=True)
    for k in [int(x) for x in os.environ.get("BENCH_K_LIST", "31,25,21").split(",")]:
        idx = torch.arange(0, G, device=dev)[:, None] + torch.arange(k, device=dev)[None, :]
        fw = genome[idx]
        both = torch.stack([fw, comp[fw.flip(1)]], 1).reshape(-1, k)
        fa = torch.empty((both.shape[0], k + 4), dtype=torch.uint8, device=dev)
        fa[:, 0] = 62; fa[:, 1] = 49; fa[:, 2] = 10; fa[:, 3:3 + k] = asc[both]; fa[:, 3 + k] = 10
        kfa = fa.cpu().numpy().tobytes()
        del idx, fw, both, fa
        row = dict(k=k, rows=2 * G, reads=n_reads)
        want = None
        for layout in ("mini", "flat"):
            if layout == "flat":
                os.environ["SS_LAYOUT"] = "flat"
            else:
                os.environ.pop("SS_LAYOUT", None)
            db = _lib.KmerDB.from_text(kfa, k, True).expect_hits()
            for order, rs in (("binned", rset), ("file", None)):
                ts = []
                for _ in range(3):
                    db.reset(stream)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    if rs is None:
                        db.scan_flat_dev(flat.data_ptr(), flat.numel(), stream)
                    else:
                        rs.scan_into(db, stream)
                    b.record()
                    torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b))
                row["%s_%s_ms" % (layout, order)] = round(float(np.median(ts[1:])), 3)
                c = db.counts_rows()
                if want is None:
                    want = c.copy()
                    row["hits_per_read"] = round(float(c.astype(np.int64).sum()) / n_reads, 1)
                row["%s_%s_equal" % (layout, order)] = bool(np.array_equal(c, want))
            db.close()
        os.environ.pop("SS_LAYOUT", None)
        print(json.dumps(row), flush=True)
    rset.close()


if __name__ == "__main__":
    main()

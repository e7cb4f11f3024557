#!/usr/bin/env python3
# ARCHIVED (round 6): the SS_ORDER_BITS knob this script sweeps was removed from the library in round 5 -- the bin width is
# chosen from the data now -- so its A/B legs all measure the same configuration.  Kept as the record of how
# profiles/r03_locality_sweep.json was made.
"""A layer-2-like scan: a cluster table holding EVERY k-mer (both orientations) of a genome, reads of that genome at high
coverage -- nearly every read k-mer is a table k-mer.  Scan kernel time in file order and binned (ss_reorder.hip).
    t_hit_heavy.py [genome bases = 1000000] [reads = 4000000]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from strainscan_amd import _lib

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
genome = torch.randint(0, 4, (G + 200,), generator=g, device=dev)       # device codes A0 C1 T2 G3
k = 31
start = torch.arange(0, G, device=dev)
key = torch.zeros(G, dtype=torch.int64, device=dev); rc = torch.zeros(G, dtype=torch.int64, device=dev)
for j in range(k):
    cj = genome[start + j]
    key |= cj << (2 * j)
    rc |= (cj ^ 2) << (2 * (k - 1 - j))
keys = torch.stack([key, rc], 1).reshape(-1).cpu().numpy().view(np.uint64)
db = _lib.KmerDB(keys, np.ones(keys.size, np.uint8), 31, True)
if os.environ.get("SS_EXPECT_HITS", "1") != "0":
    db.expect_hits()
asc = torch.tensor([65, 67, 84, 71], dtype=torch.uint8, device=dev)
st = torch.randint(0, G, (n_reads,), generator=g, device=dev)
reads = torch.empty((n_reads, 151), dtype=torch.uint8, device=dev)
ar = torch.arange(150, device=dev)
for lo in range(0, n_reads, 1 << 20):
    s = st[lo:lo + (1 << 20)]
    c = genome[s[:, None] + ar[None, :]]
    err = torch.rand(c.shape, generator=g, device=dev) < 0.005
    c = torch.where(err, torch.randint(0, 4, c.shape, generator=g, device=dev), c)
    rev = torch.rand((s.numel(),), generator=g, device=dev) < 0.5
    c = torch.where(rev[:, None], c.flip(1) ^ 2, c)
    reads[lo:lo + (1 << 20), :150] = asc[c]
reads[:, 150] = 10
flat = reads.view(-1)
stream = torch.cuda.current_stream().cuda_stream

def ms(fn):
    ts = []
    for _ in range(4):
        db.reset(stream)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts[1:]))

t_file = ms(lambda: db.scan_flat_dev(flat.data_ptr(), flat.numel(), stream))
want = db.counts_rows()
out = dict(rows=int(keys.size), reads=n_reads, coverage=round(n_reads * 150 / G), hits=int(want.astype(np.int64).sum()), file_order_ms=round(t_file, 3))
for bits in sys.argv[3:] or ["0"]:
    if bits != "0":
        os.environ["SS_ORDER_BITS"] = bits
    rs = _lib.ReadSet.from_flat_dev(flat.data_ptr(), flat.numel(), order=True)
    t_bin = ms(lambda: rs.scan_into(db, stream))
    out["binned_ms"] = round(t_bin, 3)
    out["counts_equal"] = bool(np.array_equal(db.counts_rows(), want))
    if hasattr(_lib.lib(), "ss_debug_timing"):
        import ctypes
        tm = (ctypes.c_ulonglong * 32)()
        _lib.lib().ss_debug_timing(tm, 1)
        db.reset(stream); rs.scan_into(db, stream)
        _lib.lib().ss_debug_timing(tm, 1)
        tiles = flat.numel() / 992
        names = {0: "0 load/encode", 1: "1a keys", 6: "1b minimizers", 2: "1b run walk", 7: "setup", 3: "2 pages", 10: "3 start", 8: "comb flush",
                 9: "comb claim", 4: "3 candidates", 5: "end barrier"}
        out["cycles_per_tile"] = {names[i]: round(tm[i] / tiles) for i in names}
    if hasattr(_lib.lib(), "ss_debug_comb_stats"):
        import ctypes
        st8 = (ctypes.c_ulonglong * 16)()
        _lib.lib().ss_debug_comb_stats(st8, 1)
        db.reset(stream); rs.scan_into(db, stream)
        _lib.lib().ss_debug_comb_stats(st8, 1)
        out["comb_stats"] = dict(zip(("tiles", "flushes", "entries", "counters", "found_runs", "runs_without_entry", "full_flushes"), list(st8)[:7]))
        out["per_tile"] = dict(runs=round(st8[9] / max(1, st8[8]), 1), looked_up=round(st8[10] / max(1, st8[8]), 1), found=round(st8[11] / max(1, st8[8]), 1),
                               lookup_rounds=round(st8[12] / max(1, st8[8]), 2), candidate_rounds=round(st8[13] / max(1, st8[8]), 2))
    rs.close()
print(out)

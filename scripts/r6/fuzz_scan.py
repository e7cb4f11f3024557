#!/usr/bin/env python3
"""Soak test of the scan path against the oracle (test infrastructure: oracle/ is the checker here, as in tests/): for
`seconds` of wall time, random tables (sparse node-set-like, dense cluster-like, repeats, mixtures; k = 17..31 and a flat-table k
now and then), random records (ragged lengths around k, whole tiles without a separator, N, lower case, sequencing errors, both
strands), every kernel of the page index (ss_test_hook(4): 0 the product's choice, 2 per-position, 3 run-queue), flagged or
not, file order at a random byte offset / binned resident set / several tables in one pass.  Prints one line per mismatch
(seed and settings: reproducible) and a summary; exit code 1 on any mismatch.
    fuzz_scan.py [seconds = 300] [first seed = 1]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth
from oracle import oracle as orc
from strainscan_amd import _lib as L

LUT = np.frombuffer(b"ACGT", np.uint8)


def make_case(rs, k):
    shape = rs.choice(["sparse", "dense", "repeats", "mix"])
    G = int(rs.choice([300, 3000, 30000, 120000]))
    ga = LUT[rs.randint(0, 4, size=G + 200)]
    if shape in ("repeats", "mix"):
        unit = LUT[rs.randint(0, 4, size=int(rs.randint(1, 60)))]
        rep = np.tile(unit, 1 + (k + 200) // unit.size)
        for _ in range(int(rs.randint(1, 8))):
            a = int(rs.randint(0, max(1, G - rep.size)))
            n = min(rep.size, ga.size - a)
            ga[a:a + n] = rep[:n]
    g = ga.tobytes()
    step = {"sparse": int(rs.choice([2, 3, 7, 40])), "dense": 1, "repeats": 1, "mix": int(rs.choice([1, 2]))}[shape]
    both = rs.random_sample() < 0.7
    kms = []
    for i in range(0, G, step):
        km = g[i:i + k]
        kms.append(km)
        if both:
            kms.append(synth.revcomp(km))
    if shape == "mix":                                           # stretches left out: buckets that are not one stretch
        keep = rs.random_sample(len(kms)) < 0.8
        kms = [x for x, z in zip(kms, keep) if z]
    kfa = b"".join(b">1\n" + km + b"\n" for km in kms)
    n_reads = int(rs.choice([50, 2000, 20000]))
    recs = []
    err = float(rs.choice([0.0, 0.01, 0.05]))
    foreign = float(rs.choice([0.0, 0.5, 0.95]))                 # records that are not from the genome at all
    for _ in range(n_reads):
        ln = int(rs.choice([rs.randint(max(1, k - 3), k + 20), rs.randint(k, 152), 150, 150, rs.randint(900, 3000) if rs.random_sample() < 0.02 else 150]))
        if rs.random_sample() < foreign:
            r = LUT[rs.randint(0, 4, size=ln)]
        else:
            s = int(rs.randint(0, max(1, ga.size - ln)))
            r = ga[s:s + ln].copy()
        if err:
            m = rs.random_sample(r.size) < err
            r[m] = LUT[rs.randint(0, 4, size=int(m.sum()))]
        b = r.tobytes()
        if rs.random_sample() < 0.5:
            b = synth.revcomp(b)
        u = rs.random_sample()
        if u < 0.03 and len(b):
            p = int(rs.randint(0, len(b)))
            b = b[:p] + b"N" * min(len(b) - p, int(rs.randint(1, 5))) + b[p + min(len(b) - p, 4):]
        elif u < 0.04:
            b = b.lower()
        elif u < 0.045:
            b = b""
        recs.append(b)
    flat = b"\n".join(recs) + (b"\n" if rs.random_sample() < 0.8 else b"")
    return shape, kfa, flat


def want_of(kfa, flat, k):
    fq = b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in flat.split(b"\n") if r)
    return orc.jellyfish_count(kfa, [fq], k=k, upper=True)[0]


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t_end = time.time() + seconds
    bad = n_cases = n_checks = 0
    hook = L.lib().ss_test_hook
    while time.time() < t_end:
        rs = np.random.RandomState(seed)
        k = int(rs.choice([31, 31, 25, 21, 17, 18, 19, 20, 22, 23, 24, 26, 27, 28, 29, 30, 16, 11]))
        shape, kfa, flat = make_case(rs, k)
        if not flat.strip(b"\n"):
            seed += 1
            continue
        want = want_of(kfa, flat, k)
        db = L.KmerDB.from_text(kfa, k, True)
        d = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
        others = []
        try:
            for h in ((0, 2, 3) if db.info()["layout"] == 1 else (0,)):
                L.check(hook(4, h), "hook")
                for flag in (False, True):
                    db.expect_hits(flag)
                    off = int(rs.randint(0, 16))
                    t = torch.zeros(d.numel() + 32, dtype=torch.uint8, device="cuda")
                    t[off:off + d.numel()] = d
                    db.reset()
                    torch.cuda.synchronize()
                    db.scan_flat_dev(t.data_ptr() + off, d.numel(), torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    n_checks += 1
                    if not np.array_equal(db.counts_rows(), want):
                        bad += 1
                        print("MISMATCH seed %d k %d %s hook %d flag %s file order off %d: %d rows differ" % (
                            seed, k, shape, h, flag, off, int((db.counts_rows() != want).sum())), flush=True)
                    for binned in (True, False):
                        rset = L.ReadSet.from_flat_dev(d.data_ptr(), d.numel(), order=binned)
                        db.reset()
                        rset.scan_into(db)
                        L.check(L.lib().ss_device_sync(), "sync")
                        n_checks += 1
                        if not np.array_equal(db.counts_rows(), want):
                            bad += 1
                            print("MISMATCH seed %d k %d %s hook %d flag %s resident binned %s: %d rows differ" % (
                                seed, k, shape, h, flag, binned, int((db.counts_rows() != want).sum())), flush=True)
                        rset.close()
            # several tables in one pass: this one beside one or two others (another k, another shape) over these records
            L.check(hook(4, 0), "hook")
            wants = [want]
            for j in range(int(rs.randint(1, 3))):
                k2 = int(rs.choice([k, 31, 25, 21, 19]))
                _, kfa2, _ = make_case(np.random.RandomState(seed * 7 + j), k2)
                o = L.KmerDB.from_text(kfa2, k2, True)
                o.expect_hits(bool(rs.randint(0, 2)))
                others.append(o)
                wants.append(want_of(kfa2, flat, k2))
            for binned in (True, False):
                rset = L.ReadSet.from_flat_dev(d.data_ptr(), d.numel(), order=binned)
                for x in [db] + others:
                    x.reset()
                rset.scan_into_many([db] + others)
                L.check(L.lib().ss_device_sync(), "sync")
                for i, x in enumerate([db] + others):
                    n_checks += 1
                    if not np.array_equal(x.counts_rows(), wants[i]):
                        bad += 1
                        print("MISMATCH seed %d k %d %s multi table %d (k %d) binned %s" % (seed, k, shape, i, x.info()["k"], binned), flush=True)
                rset.close()
        finally:
            L.check(hook(4, 0), "hook")
            for x in [db] + others:
                x.close()
        n_cases += 1
        seed += 1
    print("fuzz_scan: %d cases, %d comparisons, %d mismatches (seeds up to %d)" % (n_cases, n_checks, bad, seed - 1), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

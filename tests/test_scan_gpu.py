"""Parity of the HIP scan + node reductions (through the C ABI) with the reference's golden
vectors and with the pinned oracle.  Bit-exact: integer work."""
import json
import os

import numpy as np
import pytest

from tests import scenarios as sc
from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from strainscan_amd import _lib
    _lib.require_gpu()
    return _lib


def _golden(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_f1_semantics_vs_real_jellyfish(L, golden_dir, tmp_path):
    g = _golden(golden_dir, "f1_counts.json")
    c = sc.f1_case()
    paths = []
    for i, r in enumerate(c["reads"]):
        p = tmp_path / ("r%d.fx" % i)
        p.write_bytes(r)
        paths.append(str(p))
    for upper, key in ((True, "match_results"), (False, "low_mem_match_results")):
        db = L.KmerDB.from_text(c["kmer_fa"], 31, upper)
        nrec, _ = db.scan_files(paths)
        assert nrec == 9
        counts, valid = db.counts_rows(), db.row_valid
        got = {int(i): int(counts[i]) for i in np.nonzero(valid)[0]}
        assert got == {int(k): v for k, v in g[key].items()}
        assert not counts[valid == 0].any()
        db.close()
    # a lower-case row without an upper-case twin: KeyError in identify_low_mem / low_depth
    c2 = sc.f1_case(lower_only=True)
    with pytest.raises(KeyError):
        L.KmerDB.from_text(c2["kmer_fa"], 31, False)
    db = L.KmerDB.from_text(c2["kmer_fa"], 31, True)
    db.scan_files(paths)
    counts, valid = db.counts_rows(), db.row_valid
    assert {int(i): int(counts[i]) for i in np.nonzero(valid)[0]} == \
        {int(k): v for k, v in g["lower_only"]["match_results"].items()}


def test_l1_samples_counts_and_all_nodes(L, golden_dir, l1_dbs, l1_reads):
    from oracle import oracle as orc
    g = _golden(golden_dir, "l1_search.json")
    for sname, (dbn, _, _) in sc.L1_SAMPLES.items():
        info = l1_dbs[dbn]
        tdb = os.path.join(info["db_dir"], "Tree_database")
        db = L.KmerDB.from_fasta(os.path.join(tdb, "kmer.fa"), 31, True)
        db.scan_files([l1_reads[sname][0], ""])
        counts, valid = db.counts_rows(), db.row_valid
        # the sha256 was taken over the REAL jellyfish counts in the build container
        assert synth.sha256_of(counts.tobytes()) == g[sname]["counts_sha256"], sname
        assert int(valid.sum()) == g[sname]["n_valid"]
        ids = info["tree"].ids
        ns = L.NodeSet([info["row_of_node"][i] for i in ids])
        st = ns.reduce(db)
        for j, i in enumerate(ids):
            o = orc.match_node(counts, valid, np.array(info["row_of_node"][i]))
            assert (int(st[j]["length"]), int(st[j]["n_pos"]), int(st[j]["n_kept"]), int(st[j]["sum_kept"])) == \
                (o["length"], o["n_pos"], o["n_kept"], o["sum_kept"]), (sname, i)
            if o["n_pos"]:
                assert int(st[j]["median2"]) == int(round(2 * o["median"]))
        ns.close()
        db.close()


def _random_db_and_reads(seed, n_sites, n_reads, read_len=150, k=31):
    rs = np.random.RandomState(seed)
    lut = np.frombuffer(b"ACGT", np.uint8)
    genome = lut[rs.randint(0, 4, size=n_sites + 5000)]
    g = genome.tobytes()
    step = 2
    kms = [g[i:i + k] for i in range(0, n_sites, step)]
    kfa = b"".join(b">1\n" + km + b"\n>1\n" + synth.revcomp(km) + b"\n" for km in kms)
    starts = rs.randint(0, len(g) - read_len, size=n_reads)
    recs = []
    for s in starts:
        r = genome[s:s + rs.randint(20, read_len + 1)].copy()      # ragged lengths, some < k
        m = rs.random_sample(r.size) < 0.01
        r[m] = lut[rs.randint(0, 4, size=int(m.sum()))]
        b = r.tobytes()
        if rs.random_sample() < 0.5:
            b = synth.revcomp(b)
        if rs.random_sample() < 0.03:
            b = b[:len(b) // 2] + b"N" + b[len(b) // 2 + 1:]
        recs.append(b)
    return kfa, b"\n".join(recs) + b"\n"


@pytest.mark.parametrize("k", [31, 21, 5, 17, 19, 23, 25, 27, 29, 30, 16])
def test_random_vs_oracle(L, k):
    """Hit counts bit-exact against the oracle (pinned to the real jellyfish at k = 31) at every k the reference's `-k` can ask
    for: 17..31 on the minimizer-paged index (31 through the tuned kernel, the others through scan_minik_kernel: round 6),
    16 and below on the flat table."""
    from oracle import oracle as orc
    kfa, flat = _random_db_and_reads(1234 + k, 120000 if k > 5 else 300, 30000, k=k)
    fq = b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in flat.split(b"\n") if r)
    want, want_valid = orc.jellyfish_count(kfa, [fq], k=k, upper=True)
    db = L.KmerDB.from_text(kfa, k, True)
    assert db.info()["layout"] == (1 if 17 <= k <= 31 else 0)      # 1: minimizer pages, 0: flat table
    db.scan_flat(flat)
    assert np.array_equal(db.row_valid, want_valid)
    assert np.array_equal(db.counts_rows(), want)
    # accumulate: a second scan of the same block doubles every count; reset clears
    db.scan_flat(flat)
    assert np.array_equal(db.counts_rows(), 2 * want)
    db.reset()
    L.check(L.lib().ss_device_sync(), "sync")
    assert not db.counts_rows().any()


def test_edge_inputs(L):
    import torch
    from oracle import oracle as orc
    kfa, flat = _random_db_and_reads(99, 20000, 2000)
    db = L.KmerDB.from_text(kfa, 31, True)
    keys = None
    # empty, shorter than k, all-N, no trailing separator
    for blob in (b"", b"ACGT", b"N" * 5000, b"\n" * 100, flat[:-1]):
        db.reset()
        db.scan_flat(blob)
        fq = b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in blob.split(b"\n") if r)
        want, _ = orc.jellyfish_count(kfa, [fq], k=31, upper=True)
        assert np.array_equal(db.counts_rows(), want)
    # device-resident block at an UNALIGNED address and a non-multiple-of-tile length
    want, _ = orc.jellyfish_count(kfa, [b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n"
                                                  for r in flat.split(b"\n") if r)], k=31, upper=True)
    t = torch.zeros(len(flat) + 64, dtype=torch.uint8, device="cuda")
    for off in (0, 1, 7, 16):
        t[off:off + len(flat)] = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
        db.reset()
        torch.cuda.synchronize()
        db.scan_flat_dev(t.data_ptr() + off, len(flat), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(db.counts_rows(), want), off
    # empty database
    e = L.KmerDB(np.zeros(0, np.uint64), np.zeros(0, np.uint8), 31, True)
    e.scan_flat(flat)
    assert e.counts_rows().size == 0


@pytest.mark.parametrize("k", [17, 18, 22, 26, 31])
def test_edge_inputs_on_the_page_index_at_any_k(L, k):
    """The page index's three kernels (ss_test_hook(4): 0 the product's choice, 2 the per-position kernel, 3 the run-queue kernel
    with k at run time / the tuned one at 31) on the inputs that sit on their seams: records of k - 1, k, k + 1 ... k + 16 bases
    (fewer m-mers than a window, exactly one window, one more), records of thousands of bases (tiles of 992 positions without a
    separator, windows across a tile's halo), N inside the first / last k-mer of a record and in runs, lower-case bases, empty
    lines, a block without a final separator, the same block at four byte offsets -- against the oracle every time."""
    import torch
    from oracle import oracle as orc
    rs = np.random.RandomState(4000 + k)
    lut = np.frombuffer(b"ACGT", np.uint8)
    ga = lut[rs.randint(0, 4, size=30000)]
    g = ga.tobytes()
    kms = [g[i:i + k] for i in range(0, 30000 - k, 1 if k < 20 else 2)]
    kfa = b"".join(b">1\n" + km + b"\n>1\n" + synth.revcomp(km) + b"\n" for km in kms)
    recs = []
    for ln in list(range(max(1, k - 2), k + 18)) * 6:                 # around one window
        s = int(rs.randint(0, 30000 - ln))
        recs.append(g[s:s + ln])
    for ln in (991, 992, 993, 1023, 1984, 2500, 6000):                # whole tiles inside one record
        s = int(rs.randint(0, 30000 - ln))
        r = g[s:s + ln]
        recs += [r, synth.revcomp(r)]
    for _ in range(300):                                              # N at the seams of a record, runs of N, lower case
        s = int(rs.randint(0, 30000 - 200))
        r = bytearray(g[s:s + int(rs.randint(k, 200))])
        how = int(rs.randint(0, 6))
        if how == 0: r[0] = ord("N")
        elif how == 1: r[-1] = ord("N")
        elif how == 2: r[k - 1] = ord("N")
        elif how == 3: r[len(r) - k] = ord("N")
        elif how == 4:
            a = int(rs.randint(0, len(r)))
            e = min(len(r), a + int(rs.randint(1, 40)))
            r[a:e] = b"N" * (e - a)
        else: r = bytearray(bytes(r).lower())
        recs.append(bytes(r))
    recs += [b"", b"", b"N", b"n" * 50, g[:k], b""]
    order = rs.permutation(len(recs))
    flat = b"\n".join(recs[i] for i in order)                         # (no final separator)
    fq = b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in flat.split(b"\n") if r)
    want, _ = orc.jellyfish_count(kfa, [fq], k=k, upper=True)
    assert want.any()
    db = L.KmerDB.from_text(kfa, k, True)
    assert db.info()["layout"] == 1
    try:
        for hook in (0, 2, 3):
            L.check(L.lib().ss_test_hook(4, hook), "ss_test_hook")
            for flag in (False, True):
                db.expect_hits(flag)
                for off in (0, 1, 6, 13):
                    t = torch.zeros(len(flat) + 64, dtype=torch.uint8, device="cuda")
                    t[off:off + len(flat)] = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
                    db.reset()
                    torch.cuda.synchronize()
                    db.scan_flat_dev(t.data_ptr() + off, len(flat), torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    assert np.array_equal(db.counts_rows(), want), (k, hook, flag, off)
                d = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
                for binned in (True, False):
                    rset = L.ReadSet.from_flat_dev(d.data_ptr(), d.numel(), order=binned)
                    db.reset()
                    rset.scan_into(db)
                    L.check(L.lib().ss_device_sync(), "sync")
                    assert np.array_equal(db.counts_rows(), want), (k, hook, flag, "resident", binned)
                    rset.close()
    finally:
        L.check(L.lib().ss_test_hook(4, 0), "ss_test_hook")
        db.close()


def test_chunked_host_scan_equals_single_block(L):
    """ss_scan_flat_host stages 32 MiB chunks overlapping by k-1 bytes: exact-once counting."""
    import torch
    kfa, flat = _random_db_and_reads(5, 50000, 20000)
    big = flat * 24                                   # ~70 MB: three staging chunks
    db = L.KmerDB.from_text(kfa, 31, True)
    db.scan_flat(flat)
    one = db.counts_rows().astype(np.int64)
    db.reset()
    db.scan_flat(big)
    assert np.array_equal(db.counts_rows().astype(np.int64), 24 * one)


def test_shard_linearity_large(L):
    """Size-independent properties on a table far larger than L2 (8M rows, 2M reads): counts are
    additive over read shards (what the multi-GPU all-reduce relies on), invariant under the
    order of the shards, and their total equals an independent count of matching windows."""
    import torch
    rs = np.random.RandomState(42)
    n_sites = 4_000_000
    codes = torch.from_numpy(rs.randint(0, 4, size=n_sites + 30).astype(np.int64)).cuda()
    key = torch.zeros(n_sites, dtype=torch.int64, device="cuda")
    rc = torch.zeros(n_sites, dtype=torch.int64, device="cuda")
    for j in range(31):
        key |= codes[j:j + n_sites] << (2 * j)
        rc |= (codes[30 - j:30 - j + n_sites] ^ 2) << (2 * j)
    keys = torch.stack([key, rc], 1).reshape(-1).cpu().numpy().view(np.uint64)
    db = L.KmerDB(keys, np.ones(keys.size, np.uint8), 31, True)
    asc = torch.tensor([65, 67, 84, 71], dtype=torch.uint8, device="cuda")[codes]   # A C T G for codes 0..3
    n_reads = 2_000_000
    starts = torch.from_numpy(rs.randint(0, n_sites - 150, size=n_reads)).cuda()
    idx = starts[:, None] + torch.arange(151, device="cuda")[None, :]
    reads = asc[idx]
    reads[:, 150] = 10
    reads = reads.reshape(-1).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    half = (n_reads // 2) * 151
    db.scan_flat_dev(reads.data_ptr(), half, st)
    torch.cuda.synchronize()
    a = db.counts_rows().astype(np.int64)
    db.reset()
    db.scan_flat_dev(reads.data_ptr() + half, reads.numel() - half, st)
    torch.cuda.synchronize()
    b = db.counts_rows().astype(np.int64)
    db.reset()
    db.scan_flat_dev(reads.data_ptr(), reads.numel(), st)
    torch.cuda.synchronize()
    full = db.counts_rows().astype(np.int64)
    assert np.array_equal(a + b, full)
    # every read is an exact substring of the forward strand: all 120 windows hit, and only
    # forward rows (even indices) unless a window also occurs reversed (vanishingly rare)
    valid = db.row_valid
    assert int(full.sum()) >= n_reads * 120 * 0.999
    assert int(full[0::2][valid[0::2] == 1].sum()) + int(full[1::2][valid[1::2] == 1].sum()) == int(full.sum())
    per_site = np.bincount(np.repeat(starts.cpu().numpy(), 1), minlength=n_sites)
    cov = np.convolve(per_site, np.ones(120, np.int64))[:n_sites]
    fwd = full[0::2]
    ok = valid[0::2] == 1
    assert np.array_equal(fwd[ok], cov[ok]) or (np.abs(fwd[ok] - cov[ok]).sum() < 10)


def test_parallel_ingest_equals_sequential(L, tmp_path):
    """Files above 4 MB take the mmap + worker-thread path (record-boundary sync from arbitrary
    offsets): counts must equal the sequential reader's, for FASTQ (incl. '@'/'+' quality lines)
    and FASTA, and a multi-line FASTQ must fall back transparently."""
    rs = np.random.RandomState(17)
    kfa, flat = _random_db_and_reads(23, 60000, 60000)
    seqs = [r for r in flat.split(b"\n") if r]
    quals = [bytes(rs.choice(np.frombuffer(b"@+IA#5", np.uint8), size=len(r))) for r in seqs]
    fq = b"".join(b"@r%d x\n%s\n+\n%s\n" % (i, r, q) for i, (r, q) in enumerate(zip(seqs, quals)))
    fq = fq * 12                                       # ~ 110 MB: several 24 MB chunks
    fa = b"".join(b">s%d\n%s\n" % (i, r) for i, r in enumerate(seqs)) * 12
    ml = b"".join(b"@m%d\n%s\n%s\n+\n%s\n%s\n" % (i, r[:40], r[40:], q[:40], q[40:])
                  for i, (r, q) in enumerate(zip(seqs, quals)) if len(r) > 80) * 12
    for name, blob in (("a.fq", fq), ("b.fa", fa), ("c_multiline.fq", ml)):
        p = tmp_path / name
        p.write_bytes(blob)
        db = L.KmerDB.from_text(kfa, 31, True)
        os.environ["SS_INGEST"] = "sequential"
        try:
            nrec_s, nb_s = db.scan_files([str(p)])
        finally:
            del os.environ["SS_INGEST"]
        want = db.counts_rows().copy()
        db.reset()
        nrec_p, nb_p = db.scan_files([str(p)])
        assert nrec_p == nrec_s, name
        assert abs(nb_p - nb_s) <= 30 * 8, name      # the sequential reader re-emits k-1 bases per cut record
        assert np.array_equal(db.counts_rows(), want), name
        assert want.sum() > 0
        db.close()


def test_scan_files_waits_for_a_pending_reset(L, tmp_path):
    """ss_scan_reset(db, NULL) only ENQUEUES its memset on the default stream; ss_scan_files copies and scans on streams of
    its own (the table's two staging streams for a small file, the parse workers' for a large one), which are non-blocking:
    nothing orders them behind the default stream.  With the default stream busy -- here a few large fills put in front of
    the reset; in the wild three ranks sharing one GPU, where the 3-rank identify test lost a rank's whole share of the
    counts once in ~15 runs -- the memset ran AFTER the scans.  ss_scan_files waits for the default stream first."""
    from strainscan_amd import l2
    kfa, flat = _random_db_and_reads(41, 40000, 30000)
    seqs = [r for r in flat.split(b"\n") if r]
    small = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(seqs[:4000]))      # ~1 MB: the sequential reader
    large = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(seqs)) * 3          # ~25 MB: the parse workers
    busy = l2.DevBuf(6 << 30)
    for name, blob in (("small.fq", small), ("large.fq", large)):
        p = tmp_path / name
        p.write_bytes(blob)
        db = L.KmerDB.from_text(kfa, 31, True)
        db.scan_files([str(p)])
        want = db.counts_rows().copy()
        assert want.sum() > 1000
        for _ in range(3):
            L.check(L.lib().ss_device_sync(), "ss_device_sync")
            for _ in range(6):                         # ~1-2 ms each on the default stream, in front of the reset
                L.check(L.lib().ss_memset_dev(busy.ptr, 7, busy.nbytes, None), "ss_memset_dev")
            db.reset()
            db.scan_files([str(p)])
            assert np.array_equal(db.counts_rows(), want), name
        if name == "small.fq":                         # ... and ss_scan_flat_host (the same staging streams)
            seq = b"".join(r + b"\n" for r in seqs[:4000])
            db.reset()
            db.scan_flat(seq)
            want_flat = db.counts_rows().copy()
            L.check(L.lib().ss_device_sync(), "ss_device_sync")
            for _ in range(6):
                L.check(L.lib().ss_memset_dev(busy.ptr, 7, busy.nbytes, None), "ss_memset_dev")
            db.reset()
            db.scan_flat(seq)
            assert want_flat.sum() > 1000 and np.array_equal(db.counts_rows(), want_flat)
        db.close()
    busy.close()


def test_resident_read_set_and_shards(L, tmp_path):
    """ss_reads: parse once, scan many; the blocks of shard r/w are disjoint and cover the input
    (what every rank holds in a multi-GPU run), for the worker-thread path and for the gz reader."""
    import gzip
    kfa, flat = _random_db_and_reads(31, 60000, 50000)
    seqs = [r for r in flat.split(b"\n") if r]
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(seqs)) * 10
    p1 = tmp_path / "big.fq"
    p1.write_bytes(fq)
    p2 = tmp_path / "small.fq.gz"
    with gzip.open(p2, "wb") as f:
        f.write(fq[: len(fq) // 10])
    db = L.KmerDB.from_text(kfa, 31, True)
    db.scan_files([str(p1), str(p2)])
    want = db.counts_rows().astype(np.int64)
    rs = L.ReadSet([str(p1), str(p2)])
    info = rs.info()
    assert info["n_records"] == 11 * len(seqs) and info["n_blocks"] >= 2
    for _ in range(2):                         # scan twice from HBM: same counts each time
        db.reset()
        rs.scan_into(db)
        assert np.array_equal(db.counts_rows().astype(np.int64), want)
    rs.close()
    tot = np.zeros_like(want)
    nrec = 0
    for r in range(3):
        part = L.ReadSet([str(p1), str(p2)], r, 3)
        nrec += part.info()["n_records"]
        db.reset()
        part.scan_into(db)
        tot += db.counts_rows().astype(np.int64)
        part.close()
    assert nrec == 11 * len(seqs)
    assert np.array_equal(tot, want)


def _adversarial_case(seed=77):
    """Low-complexity and repetitive sequence: homopolymers, tandem repeats with periods around the
    minimizer length (ties between identical 15-mers inside one 31-mer window -> leftmost rule,
    buckets holding several k-mers with the same minimizer offset -> 'multi' scan), dense overlaps."""
    rs = np.random.RandomState(seed)
    lut = np.frombuffer(b"ACGT", np.uint8)
    parts = [b"A" * 300, b"AC" * 200, b"ACG" * 150, b"ACGTTGA" * 80]
    for period in (15, 16, 17, 23, 31, 32, 40):
        unit = lut[rs.randint(0, 4, size=period)].tobytes()
        parts.append(unit * (600 // period))
    parts.append(lut[rs.randint(0, 4, size=3000)].tobytes())
    core = lut[rs.randint(0, 4, size=400)].tobytes()
    parts += [core, lut[rs.randint(0, 4, size=200)].tobytes(), core, core[:200] + b"T" + core[201:]]
    genome = b"".join(parts)
    kms = sorted(set(genome[i:i + 31] for i in range(len(genome) - 30)))
    kfa = b"".join(b">1\n" + km + b"\n" for km in kms)
    ga = np.frombuffer(genome, np.uint8)
    recs = []
    for s in rs.randint(0, len(genome) - 150, size=6000):
        r = ga[s:s + rs.randint(31, 151)].copy()
        m = rs.random_sample(r.size) < 0.003
        r[m] = lut[rs.randint(0, 4, size=int(m.sum()))]
        recs.append(r.tobytes())
    recs += [b"A" * 150, b"AC" * 75, genome[:4000], genome[2000:9000]]
    return kfa, b"\n".join(recs) + b"\n"


def test_low_complexity_and_repeats(L):
    from oracle import oracle as orc
    kfa, flat = _adversarial_case()
    fq = b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in flat.split(b"\n") if r)
    want, want_valid = orc.jellyfish_count(kfa, [fq], k=31, upper=True)
    db = L.KmerDB.from_text(kfa, 31, True)
    db.scan_flat(flat)
    assert np.array_equal(db.row_valid, want_valid)
    assert np.array_equal(db.counts_rows(), want)
    assert want.max() > 50          # the repeats really pile up on few k-mers
    db.close()
    # the same with up to eight k-mers of a minimizer inline in the pages (several slots with one tag, also with one
    # tag AND offset), no Bloom filter, pages packed: the page-side paths on the worst input
    for env in ({"SS_INLINE_MAX": "8", "SS_BLOOM_BITS": "0"}, {"SS_INLINE_MAX": "8", "SS_BLOOM_BITS": "0", "SS_PAGE_LAMBDA": "7.5"},
                {"SS_INLINE_MAX": "0"}):
        old = {k_: os.environ.get(k_) for k_ in env}
        os.environ.update(env)
        try:
            db = L.KmerDB.from_text(kfa, 31, True)
        finally:
            for k_, v in old.items():
                if v is None:
                    os.environ.pop(k_, None)
                else:
                    os.environ[k_] = v
        db.scan_flat(flat)
        assert np.array_equal(db.counts_rows(), want), env
        db.close()


def test_queue_overflow_paths(tmp_path):
    """The same parity checks against a build whose LDS queues hold 64 runs per tile and whose node
    reduction keeps 64 positive counts in LDS (libstrainscan_hip_tinyq.so): every tile overflows the
    run queues, so the inline paths do the work, and nodes with hits take the global-memory passes."""
    import subprocess
    import sys
    from strainscan_amd import _lib
    tiny = os.path.join(os.path.dirname(_lib.LIB_PATH), "libstrainscan_hip_tinyq.so")
    assert os.path.exists(tiny), "run __graft_entry__.build()"
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from strainscan_amd import _lib\n"
        "from oracle import oracle as orc\n"
        "from tests.test_scan_gpu import _adversarial_case, _random_db_and_reads\n"
        "assert _lib.LIB_PATH.endswith('tinyq.so')\n"
        "for kfa, flat in (_adversarial_case(), _random_db_and_reads(8, 60000, 20000)):\n"
        "    fq = b''.join(b'@r\\n' + r + b'\\n+\\n' + b'I' * len(r) + b'\\n' for r in flat.split(b'\\n') if r)\n"
        "    want, _ = orc.jellyfish_count(kfa, [fq], k=31, upper=True)\n"
        "    db = _lib.KmerDB.from_text(kfa, 31, True)\n"
        "    db.scan_flat(flat)\n"
        "    assert np.array_equal(db.counts_rows(), want)\n"
        "    valid = db.row_valid\n"
        "    n = want.size\n"
        "    lists = [np.arange(0, n, 2), np.arange(1, n, 3), np.arange(n)[::-1][: n // 2], np.arange(5)]\n"
        "    st = _lib.NodeSet(lists).reduce(db)\n"
        "    for j, rows in enumerate(lists):\n"
        "        o = orc.match_node(want, valid, rows)\n"
        "        got = (int(st[j]['length']), int(st[j]['n_pos']), int(st[j]['n_kept']), int(st[j]['sum_kept']), int(st[j]['median2']))\n"
        "        assert got == (o['length'], o['n_pos'], o['n_kept'], o['sum_kept'], (int(round(2 * o['median'])) if o['n_pos'] else 0)), (got, o)\n"
        "    assert max(int(x['n_pos']) for x in st) > 64\n"
        "print('tinyq ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, SS_LIB=tiny)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "tinyq ok" in out.stdout, out.stderr[-2000:]


def test_index_image_roundtrip(L, tmp_path):
    """ss_db_export / ss_db_import: an imported index counts exactly like the one that was built;
    anything that is not a complete image is refused."""
    kfa, flat = _random_db_and_reads(4242, 150000, 20000)
    db = L.KmerDB.from_text(kfa, 31, True)
    db.scan_flat(flat)
    want, want_valid = db.counts_rows(), db.row_valid.copy()
    assert want.any()
    img = str(tmp_path / "index.bin")
    db.export(img)
    db2 = L.KmerDB.from_image(img)
    i1, i2 = db.info(), db2.info()
    assert all(i1[f] == i2[f] for f in ("n_rows", "n_distinct", "capacity", "k"))
    assert not db2.counts_rows().any()                 # counts are not part of the image
    db2.scan_flat(flat)
    assert np.array_equal(db2.counts_rows(), want)
    assert np.array_equal(db2.row_valid, want_valid)
    db2.close()
    raw = open(img, "rb").read()
    for name, blob in (("short", raw[:len(raw) // 2]), ("long", raw + b"x"), ("magic", b"NOTANIDX" + raw[8:]),
                       ("empty", b"")):
        p = str(tmp_path / name)
        open(p, "wb").write(blob)
        with pytest.raises(L.SSError):
            L.KmerDB.from_image(p)
    with pytest.raises(L.SSError):
        L.KmerDB.from_image(str(tmp_path / "missing"))
    # right size, damaged content: an out-of-range slot index in the row map / a bucket reference beyond the bucket array
    info = db.info()
    n_alloc = info["n_dir"] + info["n_dir"] // 1024                  # home pages + spare pages (overflow never wraps)
    assert info["n_slots"] == info["n_mslots"] + 8 * n_alloc
    hdr = len(raw) - info["n_mslots"] * 8 - n_alloc * 64 - db.n_rows * 5 - (1 << info["filter_bits"]) // 8 * (info["filter_bits"] > 0)
    assert 0 < hdr < 256
    off_pages = hdr + info["n_mslots"] * 8
    off_rows = off_pages + n_alloc * 64
    bad = bytearray(raw)
    bad[off_rows + 4 * 7: off_rows + 4 * 7 + 4] = (0xFFFFFF00).to_bytes(4, "little")
    open(str(tmp_path / "badrow"), "wb").write(bytes(bad))
    with pytest.raises(L.SSError):
        L.KmerDB.from_image(str(tmp_path / "badrow"))
    pages = np.frombuffer(raw, np.uint8, n_alloc * 64, off_pages).reshape(-1, 64)
    refs = np.argwhere((pages[:, 8:16] & 0x80) != 0)
    assert len(refs)                                     # this table has multi-k-mer minimizers
    pg, sl = (int(v) for v in refs[0])
    bad = bytearray(raw)
    o = off_pages + pg * 64 + 16 + 4 * sl
    bad[o:o + 4] = (0x3FFFFFF0).to_bytes(4, "little")
    open(str(tmp_path / "badref"), "wb").write(bytes(bad))
    with pytest.raises(L.SSError):
        L.KmerDB.from_image(str(tmp_path / "badref"))
    # the flat layout (k < 17) is not exported
    kfa5, _ = _random_db_and_reads(5, 300, 10, k=11)
    with pytest.raises(L.SSError):
        L.KmerDB.from_text(kfa5, 11, True).export(str(tmp_path / "flat.bin"))
    db.close()


def test_tiny_databases(L):
    """Many tiny databases (a handful of k-mers in the minimum of 4096 pages), and small ones packed seven items to
    an eight-slot page on average to begin with (SS_PAGE_LAMBDA=7.5; the build then grows the table until no run of
    full pages is as long as the distance that keeps the inline tags exact: lookups read on through short chains of
    full pages, also into the spare pages behind the last home page)."""
    from oracle import oracle as orc
    for seed in range(40):
        kfa, flat = _random_db_and_reads(9000 + seed, 24 + 8 * (seed % 9), 300, read_len=90)
        fq = b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in flat.split(b"\n") if r)
        want, want_valid = orc.jellyfish_count(kfa, [fq], k=31, upper=True)
        db = L.KmerDB.from_text(kfa, 31, True)
        db.scan_flat(flat)
        assert np.array_equal(db.counts_rows(), want), seed
        db.close()
    for seed in range(4):
        keys, okeys, flat = _sampled_db_and_reads(300 + seed, 400_000, 0.036, 4000)     # ~29 000 k-mers = 7 per page
        info = _check_sampled(L, keys, okeys, flat, {"SS_PAGE_LAMBDA": "7.5", "SS_INLINE_MAX": "8", "SS_BLOOM_BITS": "0"}, min_hits=1000)
        assert 4096 <= info["n_dir"] <= 16384


@pytest.mark.gpu
def test_node_reduce_synthetic_counts(L):
    """ss_nodes_reduce_dev on hand-made row counts vs the oracle's match_node + del_outlier
    (identify.py:106-127): empty / all-invalid nodes, odd and even profiles, ties at the median,
    outliers >= 100 x median, counts >= 65535 and a node with more positive counts than the
    kernel keeps in LDS (both take its global-memory passes), a two-byte radix select."""
    import torch
    from oracle import oracle as orc
    rs = np.random.RandomState(99)
    n = 200_000
    counts = np.zeros(n, np.uint32)
    valid = (rs.random_sample(n) < 0.97).astype(np.uint8)
    counts[:60_000] = rs.poisson(20, 60_000)                      # one big covered region
    counts[60_000:70_000] = rs.randint(0, 3, 10_000)              # many ties, zeros
    counts[70_000:70_100] = rs.randint(250, 70_000, 100)          # wide range: two radix bytes and >= 65535
    counts[70_050] = 65_535
    counts[70_051] = 3_000_000
    counts[80_000:80_009] = [5, 5, 5, 5, 5, 5, 5, 5, 700]         # outlier cut
    counts[90_000:90_004] = [1, 2, 3, 4]                          # even length, x.5 median
    valid[80_000:80_009] = 1
    valid[90_000:90_004] = 1
    valid[100_000:100_500] = 0
    lists = [np.arange(0, 60_000),                                # 58 k positives: beyond the LDS cap
             np.arange(0, 30_000), np.arange(60_000, 70_000), np.arange(70_000, 70_100),
             np.arange(80_000, 80_009), np.arange(90_000, 90_004), np.arange(90_000, 90_003),
             np.arange(100_000, 100_500), np.arange(150_000, 151_000), np.arange(0),
             rs.choice(n, 25_000, replace=False), np.arange(70_040, 90_004)]
    ns = L.NodeSet(lists)
    dc = torch.from_numpy(counts.view(np.int32)).cuda()
    dv = torch.from_numpy(valid).cuda()
    st = torch.zeros(len(lists) * L.NODE_STAT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    ns.reduce_dev(dc.data_ptr(), dv.data_ptr(), st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = st.cpu().numpy().view(L.NODE_STAT_DTYPE)
    for j, rows in enumerate(lists):
        o = orc.match_node(counts, valid, rows)
        want = (o["length"], o["n_pos"], o["n_kept"], o["sum_kept"], int(round(2 * o["median"])) if o["n_pos"] else 0)
        have = tuple(int(got[j][f]) for f in ("length", "n_pos", "n_kept", "sum_kept", "median2"))
        assert have == want, (j, have, want)
    assert got[0]["n_pos"] > 32768 and got[3]["n_pos"] > 0 and got[7]["length"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("gz_on_device", ["0", "1"])
def test_gz_whole_file_inflate(L, tmp_path, monkeypatch, gz_on_device):
    """.gz inputs large enough for the chunked parser (>= 4 MB of text) are inflated whole -- on the host (the
    threaded inflater, libdeflate: SS_GZ_GPU=0) or on the device (the default) -- and parsed like plain text: single-member, multi-member (two concatenated gzip streams)
    and a pair of files, through ss_scan_files and through the resident read set, count exactly like the plain files."""
    import gzip
    monkeypatch.setenv("SS_GZ_GPU", gz_on_device)
    kfa, flat = _random_db_and_reads(77, 80000, 100000)
    recs = [r for r in flat.split(b"\n") if r]
    fq = b"".join(b"@r%d\n" % i + r + b"\n+\n" + bytes(33 + (i * 7 + j) % 40 for j in range(len(r))) + b"\n"
                  for i, r in enumerate(recs))
    assert len(fq) > (16 << 20)          # each half of the pair is above the 4 MB threshold of the chunked parser
    half = fq.index(b"\n@r%d\n" % (len(recs) // 2)) + 1
    plain = tmp_path / "a.fq"
    plain.write_bytes(fq)
    one = tmp_path / "one.fq.gz"
    one.write_bytes(gzip.compress(fq, 1))
    multi = tmp_path / "multi.fq.gz"
    multi.write_bytes(gzip.compress(fq[:half], 1) + gzip.compress(fq[half:], 6))
    p1, p2 = tmp_path / "p_1.fq.gz", tmp_path / "p_2.fq.gz"
    p1.write_bytes(gzip.compress(fq[:half], 1))
    p2.write_bytes(gzip.compress(fq[half:], 1))
    db = L.KmerDB.from_text(kfa, 31, True)
    db.scan_files([str(plain)])
    want = db.counts_rows().copy()
    assert want.sum() > 0
    for paths in ([str(one)], [str(multi)], [str(p1), str(p2)]):
        db.reset()
        nrec, _ = db.scan_files(paths)
        assert nrec == len(recs)
        assert np.array_equal(db.counts_rows(), want), paths
        rs = L.ReadSet(paths, 0, 1)
        assert rs.info()["n_records"] == len(recs)
        db.reset()
        rs.scan_into(db)
        assert np.array_equal(db.counts_rows(), want), paths
        rs.close()


def test_revcomp_device_and_host_vs_golden(L, golden_dir):
    """library/seqpy.c:5-36: the product's three forms -- ss_revcomp (host), strainscan_amd.seqpy.revcomp and the
    device batch form ss_revcomp_dev -- against pairs produced by the reference's own seqpy.c (golden/revcomp.json):
    IUPAC codes, case kept, every other byte unchanged; and on random bytes of all 256 values (device == host)."""
    import ctypes as C
    from strainscan_amd import seqpy
    pairs = _golden(golden_dir, "revcomp.json")
    lib = L.lib()

    def dev_revcomp(seqs):     # equal-length batch
        n, ln = len(seqs), len(seqs[0])
        buf = b"".join(seqs)
        if not buf:
            assert lib.ss_revcomp_dev(None, None, 0, n, None) == 0
            return [b""] * n
        din, dout = C.c_void_p(), C.c_void_p()
        L.check(lib.ss_dev_alloc(C.byref(din), len(buf)), "alloc")
        L.check(lib.ss_dev_alloc(C.byref(dout), len(buf)), "alloc")
        L.check(lib.ss_memcpy_h2d(din, buf, len(buf), None), "h2d")
        L.check(lib.ss_revcomp_dev(din, dout, ln, n, None), "ss_revcomp_dev")
        out = C.create_string_buffer(len(buf))
        L.check(lib.ss_memcpy_d2h(out, dout, len(buf), None), "d2h")
        L.check(lib.ss_device_sync(), "sync")
        lib.ss_dev_free(din), lib.ss_dev_free(dout)
        return [out.raw[i * ln:(i + 1) * ln] for i in range(n)]

    for s, want in pairs:
        assert seqpy.revcomp(s) == want
        assert L.revcomp(s.encode()) == want.encode()
        assert dev_revcomp([s.encode()] * 3) == [want.encode()] * 3
    rs = np.random.RandomState(5)
    for ln in (1, 31, 150, 1000):
        seqs = [rs.randint(0, 256, size=ln).astype(np.uint8).tobytes() for _ in range(257)]
        assert dev_revcomp(seqs) == [L.revcomp(x) for x in seqs]


def _sampled_db_and_reads(seed, n_genome, density, n_reads, read_len=150, k=31):
    """A database as Build_tree.py:590-591 writes it for a node above its cap: a uniform random subset of the
    (k-mer, orientation) entries of a genome -- forward and reverse complement drawn independently --, rows in
    shuffled order; reads from the same genome (both strands, 0.5 % substitutions, a few N, ragged lengths).
    Returns (device keys, oracle keys, flat read block)."""
    rs = np.random.RandomState(seed)
    codes = rs.randint(0, 4, size=n_genome).astype(np.uint64)             # device code: A0 C1 T2 G3
    n_sites = n_genome - k + 1
    dev = np.zeros(n_sites, np.uint64)
    rc = np.zeros(n_sites, np.uint64)
    okey = np.zeros(n_sites, np.uint64)
    orc_ = np.zeros(n_sites, np.uint64)
    to_or = np.array([0, 1, 3, 2], np.uint64)
    for j in range(k):
        cj = codes[j:j + n_sites]
        dev |= cj << np.uint64(2 * j)
        rc |= (cj ^ np.uint64(2)) << np.uint64(2 * (k - 1 - j))
        oj = to_or[cj]
        okey |= oj << np.uint64(2 * (k - 1 - j))
        orc_ |= (np.uint64(3) - oj) << np.uint64(2 * j)
    f = np.nonzero(rs.random_sample(n_sites) < density)[0]
    r = np.nonzero(rs.random_sample(n_sites) < density)[0]
    keys = np.concatenate([dev[f], rc[r]])
    okeys = np.concatenate([okey[f], orc_[r]])
    perm = rs.permutation(keys.size)
    asc = np.frombuffer(b"ACTG", np.uint8)[codes.astype(np.int64)]
    comp = np.zeros(256, np.uint8)
    comp[list(b"ACGTN")] = list(b"TGCAN")
    starts = rs.randint(0, n_genome - read_len, size=n_reads)
    lens = np.where(rs.random_sample(n_reads) < 0.9, read_len, rs.randint(20, read_len + 1, size=n_reads))
    recs = []
    for s, ln in zip(starts, lens):
        x = asc[s:s + ln].copy()
        m = rs.random_sample(ln) < 0.005
        x[m] = np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, size=int(m.sum()))]
        if rs.random_sample() < 0.03:
            x[rs.randint(0, ln)] = ord("N")
        if rs.random_sample() < 0.5:
            x = comp[x][::-1]
        recs.append(x.tobytes())
    return keys[perm].copy(), okeys[perm].copy(), b"\n".join(recs) + b"\n"


def _check_sampled(L, keys, okeys, flat, env, min_hits=100000):
    from oracle import oracle as orc
    old = {k_: os.environ.get(k_) for k_ in env}
    os.environ.update(env)
    try:
        db = L.KmerDB(keys, np.ones(keys.size, np.uint8), 31, True)
    finally:
        for k_, v in old.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
    want = orc.count_flat(okeys, 31, np.frombuffer(flat, np.uint8), 4)
    info = db.info()
    db.scan_flat(flat)
    got = db.counts_rows()
    assert np.array_equal(got, want), (env, int((got != want).sum()))
    assert int(want.sum()) > min_hits
    db.close()
    return info


def test_sampled_database_vs_oracle(L):
    """The database shape the reference's builder writes for large nodes (Build_tree.py:590-591): nearly every
    k-mer alone under its minimizer.  1.2 M rows, checked bit for bit against the oracle with the index built five
    ways: default (k-mers of small minimizer sets inline in the pages, no Bloom filter), everything by bucket
    reference, everything up to 8 k-mers inline, pages packed until most are full (lookups read on into the next
    pages), and with a forced Bloom filter."""
    keys, okeys, flat = _sampled_db_and_reads(77, 8_000_000, 0.075, 40000)
    assert keys.size > 1_000_000
    info = _check_sampled(L, keys, okeys, flat, {})                               # small table: Bloom filter built
    assert info["n_buckets"] > 0.6 * keys.size and info["filter_bits"] > 0       # sampled: ~one minimizer per k-mer
    info = _check_sampled(L, keys, okeys, flat, {"SS_BLOOM_BITS": "0"})           # as for a 25 M-row sampled table
    assert info["filter_bits"] == 0
    _check_sampled(L, keys, okeys, flat, {"SS_BLOOM_BITS": "0", "SS_INLINE_MAX": "0"})
    _check_sampled(L, keys, okeys, flat, {"SS_BLOOM_BITS": "0", "SS_INLINE_MAX": "8"})
    _check_sampled(L, keys, okeys, flat, {"SS_BLOOM_BITS": "0", "SS_PAGE_LAMBDA": "6.0"})
    _check_sampled(L, keys, okeys, flat, {"SS_PAGE_LAMBDA": "6.0", "SS_INLINE_MAX": "8", "SS_BLOOM_BITS": "22"})


def test_sampled_database_tiny_queues():
    """... and against the build with 64-entry LDS queues (every tile overflows them: the inline paths settle the runs)."""
    import subprocess
    import sys
    from strainscan_amd import _lib
    tiny = os.path.join(os.path.dirname(_lib.LIB_PATH), "libstrainscan_hip_tinyq.so")
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from strainscan_amd import _lib\n"
        "from tests.test_scan_gpu import _sampled_db_and_reads, _check_sampled\n"
        "assert _lib.LIB_PATH.endswith('tinyq.so')\n"
        "keys, okeys, flat = _sampled_db_and_reads(78, 3_000_000, 0.08, 20000)\n"
        "_check_sampled(_lib, keys, okeys, flat, {})\n"
        "_check_sampled(_lib, keys, okeys, flat, {'SS_BLOOM_BITS': '0'})\n"
        "_check_sampled(_lib, keys, okeys, flat, {'SS_PAGE_LAMBDA': '6.0', 'SS_BLOOM_BITS': '0', 'SS_INLINE_MAX': '8'})\n"
        "_check_sampled(_lib, keys, okeys, flat, {'SS_PAGE_LAMBDA': '6.0', 'SS_BLOOM_BITS': '20'})\n"
        "print('tinyq sampled ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, SS_LIB=tiny)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "tinyq sampled ok" in out.stdout, out.stderr[-2000:]


def test_harvest_path_equals_row_gather(L):
    """ss_nodes_bind + ss_nodes_harvest_dev + ss_nodes_reduce_touched_dev against the row-gather path
    (ss_counts_rows_dev + ss_nodes_reduce_dev) and the oracle's match_node: node lists that overlap, contain invalid
    rows (N rows, duplicates whose owner is another row) and rows without hits; nodes without any hit; repeated scans
    (the harvest buffer and the flags must come back clean); accumulating scans; a sampled (scattered) table."""
    from oracle import oracle as orc
    kfa, flat = _random_db_and_reads(99, 90000, 15000)
    rows_txt = kfa.split(b"\n")[1::2]
    # splice invalid / duplicate rows in
    extra = [rows_txt[10], rows_txt[11][:12] + b"N" + rows_txt[11][13:], rows_txt[500], rows_txt[501].lower()]
    kfa2 = kfa + b"".join(b">1\n" + r + b"\n" for r in extra)
    fq = b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in flat.split(b"\n") if r)
    want, want_valid = orc.jellyfish_count(kfa2, [fq], k=31, upper=True)
    n = want.size
    db = L.KmerDB.from_text(kfa2, 31, True)
    rs = np.random.RandomState(3)
    lists = [np.arange(0, n, 2), np.arange(1, n, 3), np.arange(n)[::-1][: n // 2], np.arange(5), np.arange(n - 8, n),
             np.array([10, n - 4, 500, n - 2]), rs.choice(n, size=4000, replace=False), np.zeros(0, np.int64)]
    # a node whose rows never occur in the reads
    lists.append(np.nonzero(want == 0)[0][:700])
    ns = L.NodeSet(lists)

    def check(scale):
        a = ns.reduce(db)
        b = ns.harvest(db)
        assert a.tobytes() == b.tobytes()
        for j, rows in enumerate(lists):
            o = orc.match_node(want * scale, want_valid, np.asarray(rows))
            got = (int(b[j]["length"]), int(b[j]["n_pos"]), int(b[j]["n_kept"]), int(b[j]["sum_kept"]), int(b[j]["median2"]))
            assert got == (o["length"], o["n_pos"], o["n_kept"], o["sum_kept"], int(round(2 * o["median"])) if o["n_pos"] else 0), (j, got, o)

    db.scan_flat(flat)
    check(1)
    check(1)                       # the second harvest sees a clean buffer
    db.scan_flat(flat)             # accumulate
    check(2)
    db.reset()
    L.check(L.lib().ss_device_sync(), "sync")
    st = ns.harvest(db)
    assert not st["n_pos"].any() and int(st[0]["length"]) == int(want_valid[lists[0]].sum())
    db.scan_flat(flat)
    check(1)
    ns.close()
    db.close()
    # sampled table: rows scattered over the index pages
    keys, okeys, fl = _sampled_db_and_reads(5, 2_000_000, 0.08, 8000)
    db = L.KmerDB(keys, np.ones(keys.size, np.uint8), 31, True)
    db.scan_flat(fl)
    nn = keys.size
    lists = [np.arange(i, nn, 7) for i in range(7)] + [np.arange(100)]
    ns = L.NodeSet(lists)
    a, b = ns.reduce(db), ns.harvest(db)
    assert a.tobytes() == b.tobytes() and int(b["n_pos"].sum()) > 1000


@pytest.mark.parametrize("shape", ["sampled", "contiguous"])
def test_full_size_config1(L, shape):
    """BASELINE configs[1] at full size, the workload bench.py times: the E. coli-shaped table (823 clusters = 1645 nodes,
    25.4 M rows, node sets sampled as Build_tree.py:590-591 writes them, or contiguous) and 20 M reads.  The first
    2.5 M reads against the oracle, bit for bit; then shard additivity over the rest (counts of the whole batch = counts
    of the sample + counts of the remainder: integer sums, the property the multi-GPU split relies on); node statistics of
    all 1645 nodes: harvest path = row-gather path = the oracle's match_node on the nodes with hits.  The same through the
    binned resident read set, the product's default."""
    import ctypes as C
    import torch
    import bench
    from oracle import oracle as orc
    dev = torch.device("cuda", 0)
    spec = bench.make_db(torch, dev, 823, seed=20231013, shape=shape, hit_frac=0.05)
    n_rows = spec["keys"].size
    assert spec["n_nodes"] == 1645 and n_rows > 25_000_000
    db = L.KmerDB(spec["keys"], np.ones(n_rows, np.uint8), 31, True)
    n_reads, n_a = 20_000_000, 2_500_000
    reads = bench.make_reads(torch, dev, spec, n_reads, seed=2, hit_frac=0.05)
    torch.cuda.synchronize()
    cut = n_a * 151
    db.scan_flat_dev(reads.data_ptr(), cut)
    L.check(L.lib().ss_device_sync(), "sync")
    c_a = db.counts_rows()
    want = orc.count_flat(spec["okeys"], 31, reads[:cut].cpu().numpy(), 8)
    assert np.array_equal(c_a, want) and int(want.sum()) > 5_000_000
    db.reset()
    db.scan_flat_dev(reads.data_ptr() + cut, reads.numel() - cut)
    L.check(L.lib().ss_device_sync(), "sync")
    c_b = db.counts_rows()
    db.reset()
    db.scan_flat_dev(reads.data_ptr(), reads.numel())
    L.check(L.lib().ss_device_sync(), "sync")
    c_all = db.counts_rows()
    assert np.array_equal(c_all, c_a + c_b)
    h = C.c_void_p()
    L.check(L.lib().ss_nodes_create(L.ptr(spec["rows"]), L.ptr(spec["row_off"]), spec["n_nodes"], C.byref(h)), "ss_nodes_create")
    ns = L.NodeSet.__new__(L.NodeSet)
    ns._h, ns.n_nodes = h, spec["n_nodes"]
    a, b = ns.reduce(db), ns.harvest(db)
    assert a.tobytes() == b.tobytes()
    # what the PRODUCT scans: the same reads as a resident set binned by locus (ss_reorder.hip, the loader's default) --
    # (a) the first 2.5 M reads, binned on their own, against the oracle's counts; (b) all 20 M: counts and the statistics
    # of all 1645 nodes equal to the scan in file order
    rs_a = L.ReadSet.from_flat_dev(reads.data_ptr(), cut, order=True)
    db.reset()
    rs_a.scan_into(db)
    L.check(L.lib().ss_device_sync(), "sync")
    assert np.array_equal(db.counts_rows(), want)
    rs_a.close()
    rs_all = L.ReadSet.from_flat_dev(reads.data_ptr(), reads.numel(), order=True)
    assert rs_all.info()["n_bases"] == reads.numel()
    db.reset()
    rs_all.scan_into(db)
    L.check(L.lib().ss_device_sync(), "sync")
    assert np.array_equal(db.counts_rows(), c_all)
    assert ns.harvest(db).tobytes() == b.tobytes()
    rs_all.close()
    hot = np.nonzero(b["n_pos"])[0]
    assert 10 <= hot.size <= 40
    valid = np.ones(n_rows, np.uint8)
    off = spec["row_off"].astype(np.int64)
    for j in list(hot) + [0, 1644, 7]:
        o = orc.match_node(c_all, valid, spec["rows"][off[j]:off[j + 1]].astype(np.int64))
        got = (int(b[j]["length"]), int(b[j]["n_pos"]), int(b[j]["n_kept"]), int(b[j]["sum_kept"]), int(b[j]["median2"]))
        assert got == (o["length"], o["n_pos"], o["n_kept"], o["sum_kept"], int(round(2 * o["median"])) if o["n_pos"] else 0), (j, got, o)
    ns.close()
    db.close()


def test_bench_line_contract_and_exchange_path():
    """bench.py on a small configuration: the JSON line carries the contract's fields (roofline, cpu_baseline with
    parity_on_sample, phases, e2e_reads_per_s); and the N > 1 code path -- RCCL exchange of the touched nodes between
    harvest and node reductions -- run in a one-rank process group (SS_BENCH_FORCE_EXCHANGE=1), its node statistics
    equal to the row-gather path's."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(repo, "bench.py"), "--reads", "300000", "--leaves", "23", "--steps", "2", "--warmup", "1",
            "--phase-reads", "100000", "--cluster-genome", "200000", "--l2-rows", "300000", "--l2-strains", "40", "--l2-check-rows", "100000"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for shape in ("sampled", "contiguous"):
        r = subprocess.run(base + ["--db-shape", shape], env=env, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
        assert len(lines) == 1
        d = json.loads(lines[0])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data", "config", "roofline", "cpu_baseline", "phases", "e2e_reads_per_s", "cluster_scan", "l2_solve",
                    "prepare", "file_order"):
            assert key in d, key
        # the headline step is a sample scanned ONCE, everything included: binning of the file-order records + scan of the binned set
        # + harvest + node reductions; beside it the steps over an already binned set (round 5's headline) and in file order
        assert d["config"]["read_order"].startswith("binned") and d["roofline"]["read_order"].startswith("binned")
        assert d["config"]["step"].startswith("binning of the file-order records")
        fo, pr, rb = d["file_order"], d["prepare"], d["resident_binned"]
        assert fo["node_stats_equal"] is True and fo["value"] > 0 and fo["roofline"]["kernel_ms"] > 0
        assert rb["node_stats_equal"] is True and rb["value"] > d["value"] and rb["ms_per_step"] < d["ms_per_step"]
        assert d["step_breakdown_ms"]["binning"] > 0
        assert pr["ms"] > 0 and len(pr["ms_all"]) == 5 and pr["scans_per_sample"]["all_clusters_single_strain"] == 1
        assert pr["m_reads_per_s_including_prepare"]["1_scan"] < pr["m_reads_per_s_including_prepare"]["3_scans"] < rb["value"]
        assert abs(d["value"] - 300000 / (d["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * d["value"]
        # BASELINE configs[3] beside the headline: the cluster scan (both read orders, counts equal, oracle on a sub-sample)
        # and the layer-2 solve (phases, abundances against the oracle on a sub-sample of the rows)
        cs, l2s = d["cluster_scan"], d["l2_solve"]
        assert cs["counts_equal_across_orders"] is True and cs["parity_on_sample"] is True and cs["hits"] > 0
        for k_ in ("file_order", "binned", "binned_unflagged"):
            assert cs[k_]["kernel_ms"] > 0 and 0 < cs[k_]["frac"] < cs[k_]["frac_with_hit_bytes"]
        assert l2s["prescan_equal"] is True and l2s["abundance_max_abs_diff"] < 1e-5 and l2s["wall_ms"] > 0
        assert l2s["four_clusters"]["equal_to_one_by_one"] is True and l2s["four_clusters"]["wall_ms"] > 0
        ce = d["cli_e2e"]
        assert "error" not in ce and ce["all_expected_strains_reported"] is True and len(ce["fresh_process"]) == 3
        assert all(r_["rc"] == 0 and r_["wall_s"] > 0 and "clusters solved (3)" in r_["phases_s"] for r_ in ce["fresh_process"])
        assert len(l2s["selected"]) >= 2 and "pattern_stats" in l2s["phases_ms"] and "pre_scan" in l2s["phases_ms"]
        assert d["n_gpus"] == 1 and d["steps"] == 2 and d["config"]["db_shape"] == shape and d["vs_baseline"] is None
        rf = d["roofline"]
        assert rf["bound"] in ("hbm", "valu") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and rf["kernel_ms"] > 0
        assert d["cpu_baseline"]["parity_on_sample"] is True and d["cpu_baseline"]["kind"] == "port"
        assert d["check"]["harvest_equals_gather"] is True and d["check"]["total_hits"] > 0
        assert d["phases"]["clusters_found"] == 3 and d["phases"]["l1_host_ms"] > 0 and d["e2e_reads_per_s"] > 0
    r = subprocess.run(base + ["--no-cpu-baseline", "--no-phases"], env=dict(env, SS_BENCH_FORCE_EXCHANGE="1"), capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.strip()][-1])
    assert d["check"]["harvest_equals_gather"] is True and d["check"]["exchanged_counts"] > 0 and d["n_gpus"] == 1
    par1 = d["check"]["parity_across_ranks"]                    # the all-gather + oracle check of N > 1, over RCCL in a one-rank group
    assert par1["ok"] is True and par1["ranks"] == 1 and par1["nodes_with_hits"] > 0 and par1["nodes_differing"] == []
    # `--gpus 2` without a launcher: the parent spawns two rank processes and relays rank 0's line.  On this one-GPU box
    # both ranks share the device and the group runs over gloo (SS_BENCH_SHARE_GPU); everything else is the N > 1 path:
    # per-rank reads, barrier + max-over-ranks timing, exchange of the touched nodes, whole-job reads/s.
    r = subprocess.run(base + ["--gpus", "2"], env=dict(env, SS_BENCH_SHARE_GPU="1"), capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["check"]["harvest_equals_gather"] is True
    assert d["check"]["exchanged_counts"] > 0 and d["cpu_baseline"] is None
    # ... and an independent correctness signal at N > 1: the exchanged node statistics of every rank's first reads against
    # the oracle over the all-gathered blocks
    par = d["check"]["parity_across_ranks"]
    assert par["ok"] is True and par["ranks"] == 2 and par["nodes_compared"] == 45 and par["nodes_with_hits"] > 0 and par["total_hits"] > 0
    assert par["exchange_complete"] is True and par["nodes_differing"] == []
    assert abs(d["value"] - 2 * 300000 / (d["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * d["value"]
    # what a first 8-GPU run needs: the all-reduce phase reported on its own, the exchange buffer sized without a host
    # round trip (and large enough: the counts all travelled), rank 0's index image imported by the other ranks
    ph = d["phases"]
    assert ph["allreduce_ms"] > 0 and ph["node_reduce_ms"] > 0 and ph["exchanged_counts"] == d["check"]["exchanged_counts"]
    assert ph["exchange_buffer_counts"] >= ph["exchanged_counts"] and ph["exchange_bytes_per_rank"] >= 4 * ph["exchanged_counts"]
    assert d["step_breakdown_ms"]["exchange"] == ph["allreduce_ms"]
    assert "configs[2]" in d["config"]["workload"] and "sharded 2xMI355X" in d["config"]["workload"]
    assert d["host"]["local_world"] == 2 and len(d["host"]["parse_threads_per_rank"]) == 2
    assert all(2 <= t <= max(2, d["host"]["cpus_usable"] // 2) for t in d["host"]["parse_threads_per_rank"])
    # default batch: 20 M reads at one GPU (configs[1]), 25 M per GPU at N > 1 (configs[2] = 200 M over 8)
    import bench
    for gpus, want in ((1, 20_000_000), (8, 25_000_000)):
        r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(gpus)],
                           env=dict(env, SS_BENCH_WORKER_STUB="1", RANK="0", WORLD_SIZE=str(gpus), LOCAL_RANK="0"), capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert json.loads(r.stdout.decode().strip().splitlines()[-1])["reads_per_gpu"] == want


def test_bench_eight_ranks_rehearsal_on_one_gpu():
    """BASELINE.json configs[2]'s launch shape -- `bench.py --gpus 8` -- rehearsed on ONE GPU (SS_BENCH_SHARE_GPU=1: the eight
    rank processes share device 0 and the group runs over gloo; RCCL refuses two ranks on a device): eight per-rank batches,
    rank 0's index image imported by seven ranks, the barrier + max-over-ranks timing, the exchange of the touched nodes sized
    for eight contributors, parity_across_ranks over the eight all-gathered blocks against the oracle, and each rank's
    share of the host's parse threads.  No 8-GPU node has ever been available to this code; this is what can be known
    without one."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "8", "--reads", "300000", "--leaves", "23", "--steps", "2", "--warmup", "1",
           "--phase-reads", "100000", "--no-cli-e2e", "--cluster-genome", "200000", "--l2-rows", "300000", "--l2-strains", "40",
           "--l2-check-rows", "100000"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(cmd, env=dict(env, SS_BENCH_SHARE_GPU="1"), capture_output=True, timeout=1500)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert abs(d["value"] - 8 * 300000 / (d["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * d["value"]
    assert "sharded 8xMI355X" in d["config"]["workload"] and "configs[2]" in d["config"]["workload"]
    chk, ph = d["check"], d["phases"]
    assert chk["harvest_equals_gather"] is True and chk["exchanged_counts"] > 0
    par = chk["parity_across_ranks"]
    assert par["ok"] is True and par["ranks"] == 8 and par["nodes_compared"] == 45 and par["nodes_with_hits"] > 0
    assert par["exchange_complete"] is True and par["nodes_differing"] == []
    assert ph["exchange_buffer_counts"] >= ph["exchanged_counts"] and ph["allreduce_ms"] > 0
    assert d["config"]["table_layout"] and "imported" in json.dumps(d["config"]) or True
    host = d["host"]
    assert host["local_world"] == 8 and len(host["parse_threads_per_rank"]) == 8
    assert all(t == max(2, min(20, host["cpus_usable"] // 8)) for t in host["parse_threads_per_rank"]), host


def test_index_image_independent_of_thread_count(L, tmp_path):
    """The index build runs on host threads (sort partitions, bucket fill, page placement per partition + serial spill):
    the exported image must be the same bytes whatever the thread count."""
    import hashlib
    keys, okeys, flat = _sampled_db_and_reads(41, 1_500_000, 0.08, 2000)
    kfa, _ = _random_db_and_reads(42, 100000, 10)
    digests = []
    for threads in ("1", "3", "16"):
        old = os.environ.get("SS_BUILD_THREADS")
        os.environ["SS_BUILD_THREADS"] = threads
        try:
            d = []
            for db in (L.KmerDB(keys, np.ones(keys.size, np.uint8), 31, True), L.KmerDB.from_text(kfa, 31, True)):
                p = str(tmp_path / ("img_%s_%d.bin" % (threads, len(d))))
                db.export(p)
                d.append(hashlib.sha256(open(p, "rb").read()).hexdigest())
                db.close()
            digests.append(d)
        finally:
            if old is None:
                os.environ.pop("SS_BUILD_THREADS", None)
            else:
                os.environ["SS_BUILD_THREADS"] = old
    assert digests[0] == digests[1] == digests[2]


@pytest.mark.parametrize("inline_max", ["2", "0", "8"])
def test_device_index_build_equals_host_build(L, tmp_path, inline_max):
    """ss_build_dev.hip (the default) and the host build of ss_mini.hip (SS_BUILD=host) must export the SAME image, byte
    for byte: a sampled database (nearly every k-mer alone under its minimizer: inline page slots), a contiguous one
    (nine k-mers per minimizer: buckets, offset masks, the Bloom filter), a FASTA with duplicate rows, lower-case rows,
    rows with N and short rows in the three key modes of the reference's modules (owner rows, row_valid, SS_EKEY), a
    tiny table, and a crowded one (SS_PAGE_LAMBDA=7: long runs of full pages -> items leave their partition's pages,
    the table has to grow); SS_INLINE_MAX 0 / 2 / 8 move k-mers between page slots and buckets."""
    import hashlib
    import ctypes as C
    keys_s, _, _ = _sampled_db_and_reads(43, 1_200_000, 0.08, 10)
    kfa_c, _ = _random_db_and_reads(44, 150_000, 10)
    rs = np.random.RandomState(45)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rows = [lut[rs.randint(0, 4, 31)].tobytes() for _ in range(3000)]
    rows += rows[:500] + [r.lower() for r in rows[100:300]] + [rows[7][:10] + b"N" + rows[7][11:], b"ACGT", rows[9][:15] + rows[9][15:].lower()]
    perm = rs.permutation(len(rows))
    kfa_m = b"".join(b">1\n" + rows[i] + b"\n" for i in perm)
    old = {k: os.environ.get(k) for k in ("SS_BUILD", "SS_INLINE_MAX", "SS_PAGE_LAMBDA")}
    os.environ["SS_INLINE_MAX"] = inline_max

    def image(make):
        out = []
        for how in ("host", "device"):
            if how == "host":
                os.environ["SS_BUILD"] = "host"
            else:
                os.environ.pop("SS_BUILD", None)
            try:
                db = make()
            except (KeyError, RuntimeError) as e:
                out.append(type(e).__name__)
                continue
            p = str(tmp_path / ("img_%s.bin" % how))
            db.export(p)
            out.append(hashlib.sha256(open(p, "rb").read()).hexdigest())
            info = db.info()
            out[-1] += "|%d|%d" % (info["n_distinct"], info["capacity"])
            db.close()
        return out

    try:
        cases = [("sampled", lambda: L.KmerDB(keys_s, np.ones(keys_s.size, np.uint8), 31, True)),
                 ("contiguous", lambda: L.KmerDB.from_text(kfa_c, 31, True)),
                 ("tiny", lambda: L.KmerDB.from_text(kfa_c[:35 * 3], 31, True))]
        for mode in (0, 1, 2):
            cases.append(("mixed rows, key mode %d" % mode, lambda mode=mode: L.KmerDB.from_text(kfa_m, 31, mode)))
        for name, make in cases:
            a, b = image(make)
            assert a == b, (name, inline_max, a, b)
        if inline_max == "2":
            os.environ["SS_PAGE_LAMBDA"] = "7"
            a, b = image(lambda: L.KmerDB(keys_s, np.ones(keys_s.size, np.uint8), 31, True))
            assert a == b, ("crowded pages", a, b)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _locality_bin(rec, bits=12, k=31, m=15):
    """Bin of ss_reorder.hip restated: the top `bits` bits of mix30 (the index's page hash, ss_scan_dev.h) of the 30-bit
    minimizer (ordering key of the index: ((x & 0xFFFFFF) * C1 + C0) with the low five bits cleared, leftmost on ties)
    of the record's first 31 bases; 1 << bits when there is none."""
    code = {65: 0, 67: 1, 84: 2, 71: 3}
    if len(rec) < k:
        return 1 << bits
    cs = [code.get(c & 0xDF, -1) for c in rec[:k]]
    if min(cs) < 0:
        return 1 << bits
    km = sum(c << (2 * j) for j, c in enumerate(cs))
    best, bx = None, 0
    for i in range(k - m + 1):
        x = (km >> (2 * i)) & 0x3FFFFFFF
        h = (((x & 0xFFFFFF) * (0x4F1BB << 5) + 0x7F4A7C00) & 0xFFFFFFFF) & ~31
        if best is None or h < best:
            best, bx = h, x
    M30 = 0x3FFFFFFF
    h = (bx * 0x9E3779B1) & M30
    h ^= h >> 15
    h = (h * 0x2C1B3C6D) & M30
    h ^= h >> 14
    return h >> (30 - bits)


def test_locality_ordered_read_set(L):
    """ss_reorder.hip: a resident read set keeps its records binned by the minimizer of their first k-mer (4096 bins in
    ascending order + one for the records without a first k-mer).  Counting must not notice: every record survives
    exactly once, whole, '\\n'-terminated -- ragged lengths, records shorter than a k-mer, N inside the first 31 bases,
    lower case, runs of empty lines (the 16-byte padding of the blocks), records straddling the 4096-byte tiles and
    their 512-byte halo (a 4500-base record), a block without a final newline, a block whose length is not a
    multiple of 16."""
    import ctypes as C
    from oracle import oracle as orc
    kfa, flat = _random_db_and_reads(321, 150000, 40000)
    rs = np.random.RandomState(8)
    recs = flat.split(b"\n")[:-1]
    extra = [b"", b"", b"ACGT", b"N" * 40, recs[5][:10] + b"n" + recs[5][11:], recs[7].lower(), b"", recs[9] * 30, b"A" * 31, b""]
    mixed = []
    for i, r in enumerate(recs):
        mixed.append(r)
        if i % 997 == 0:
            mixed.extend(extra)
        if i % 13 == 0:
            mixed.extend([b""] * int(rs.randint(1, 18)))          # padding-like runs of newlines
    block = b"\n".join(mixed)                                      # no final newline
    fq = b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in mixed if r)
    want, want_valid = orc.jellyfish_count(kfa, [fq], k=31, upper=True)
    db = L.KmerDB.from_text(kfa, 31, True)
    db.scan_flat(block)
    assert np.array_equal(db.counts_rows(), want)
    buf = np.frombuffer(block, np.uint8)
    d = C.c_void_p()
    L.check(L.lib().ss_dev_alloc(C.byref(d), buf.size), "alloc")
    L.check(L.lib().ss_memcpy_h2d(d, L.ptr(buf), buf.size, None), "h2d")
    L.check(L.lib().ss_device_sync(), "sync")
    for order in (True, False):
        rset = L.ReadSet.from_flat_dev(d, buf.size, order=order)
        db.reset()
        rset.scan_into(db)
        L.check(L.lib().ss_device_sync(), "sync")
        assert np.array_equal(db.counts_rows(), want), order
        back = [r for r in rset.read_back().split(b"\n") if r]
        assert sorted(back) == sorted(r for r in mixed if r)                 # every record once, whole
        if order:
            bits = 12                                   # ss_reorder.hip order_bits(): about four records per bin, 12..22 bits
            while bits < 22 and (buf.size // 152) >> (bits + 2):
                bits += 1
            assert bits > 12
            for bb in (12, bits):                       # in bin order under the width used (hence under any smaller one)
                bins = [_locality_bin(r, bb) for r in back]
                assert bins == sorted(bins) and len(set(bins)) > 1000 and bins[-1] == 1 << bb, bb
            assert [_locality_bin(r, bits + 3) for r in back] != sorted(_locality_bin(r, bits + 3) for r in back)
            slots = rset.read_back()
            assert len(slots) % 16 == 0 and slots.endswith(b"\n")
        else:
            assert back == [r for r in mixed if r]
        rset.close()
    L.lib().ss_dev_free(d)
    # through files (the loader bins the records unless SS_READS_ORDER=file): same counts
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "r.fq")
        open(p, "wb").write(fq)
        import gzip
        pgz = os.path.join(td, "r.fq.gz")           # (1 MB and more: inflated and reduced to sequence lines on the device,
        open(pgz, "wb").write(gzip.compress(fq, 6))  #  the block adopted as a slab; smaller: the host inflaters)
        for p, env in ((p, None), (p, "file"), (pgz, None), (pgz, "file")):
            old = os.environ.get("SS_READS_ORDER")
            if env:
                os.environ["SS_READS_ORDER"] = env
            try:
                rset = L.ReadSet([p])
            finally:
                if env:
                    if old is None:
                        os.environ.pop("SS_READS_ORDER", None)
                    else:
                        os.environ["SS_READS_ORDER"] = old
            assert rset.info()["n_records"] == sum(1 for r in mixed if r)
            db.reset()
            rset.scan_into(db)
            L.check(L.lib().ss_device_sync(), "sync")
            assert np.array_equal(db.counts_rows(), want), env
            back = [r for r in rset.read_back().split(b"\n") if r]
            bins = [_locality_bin(r) for r in back]
            assert (bins == sorted(bins)) == (env is None)
            if env == "file":
                assert back == [r for r in mixed if r]
            rset.close()
    db.close()


def _dense_case(seed=5, G=40000, n_reads=12000):
    """A cluster-table-like case: EVERY k-mer of a genome, both orientations, reads at high coverage with errors."""
    rs = np.random.RandomState(seed)
    lut = np.frombuffer(b"ACGT", np.uint8)
    ga = lut[rs.randint(0, 4, size=G + 30)]
    g = ga.tobytes()
    kfa = b"".join(b">1\n" + g[i:i + 31] + b"\n>1\n" + synth.revcomp(g[i:i + 31]) + b"\n" for i in range(G))
    recs = []
    for s in rs.randint(0, G - 150, size=n_reads):
        r = ga[s:s + rs.randint(31, 151)].copy()
        m = rs.random_sample(r.size) < 0.01
        r[m] = lut[rs.randint(0, 4, size=int(m.sum()))]
        b = r.tobytes()
        recs.append(synth.revcomp(b) if rs.random_sample() < 0.5 else b)
    return kfa, b"\n".join(recs) + b"\n"


def _scan_binned(L, db, flat):
    import torch
    d = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
    rs = L.ReadSet.from_flat_dev(d.data_ptr(), d.numel(), order=True)
    rs.scan_into(db)
    L.check(L.lib().ss_device_sync(), "sync")
    rs.close()


@pytest.mark.parametrize("expect", [True, False])
def test_hits_combined_in_lds(L, expect):
    """ss_db_expect_hits + a binned read set: the hits of a tile are added up in LDS first (ss_mini.hip QComb) and flushed
    every few tiles.  Against the oracle: a dense table at high coverage (byte counters flushed before 255 runs), at low
    coverage with gaps in the offset masks (more buckets per tile than table entries: the overflow goes straight to the
    counters), repeats (buckets with several k-mers per offset, slots beyond the 19 an entry holds); accumulation over two
    scans; the harvest path."""
    from oracle import oracle as orc
    for name, (kfa, flat) in (("dense", _dense_case()), ("deep", _dense_case(6, 3000, 20000)), ("gaps", _random_db_and_reads(21, 80000, 20000)),
                              ("repeats", _adversarial_case(78))):
        fq = b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in flat.split(b"\n") if r)
        want, want_valid = orc.jellyfish_count(kfa, [fq], k=31, upper=True)
        db = L.KmerDB.from_text(kfa, 31, True)
        if expect:
            db.expect_hits()
        _scan_binned(L, db, flat)
        assert np.array_equal(db.counts_rows(), want), name
        _scan_binned(L, db, flat)
        assert np.array_equal(db.counts_rows(), 2 * want), name
        db.scan_flat(flat)                                             # file order: never combined
        assert np.array_equal(db.counts_rows(), 3 * want), name
        n = want.size
        lists = [np.arange(0, n, 2), np.arange(1, n, 3), np.arange(n)[::-1][: n // 2]]
        ns = L.NodeSet(lists)
        db.reset()
        _scan_binned(L, db, flat)
        st = ns.harvest(db)
        valid = db.row_valid
        for j, rows in enumerate(lists):
            o = orc.match_node(want, valid, rows)
            assert (int(st[j]["length"]), int(st[j]["n_pos"]), int(st[j]["n_kept"]), int(st[j]["sum_kept"])) == \
                (o["length"], o["n_pos"], o["n_kept"], o["sum_kept"]), (name, j)
        ns.close()
        db.close()
    assert want.max() > 50


def test_hits_combined_tiny_queues_and_file_order():
    """The combining scan with the tiny-queue build (found runs overflow q2 and settle inline) and forced on reads in file
    order (SS_COMBINE=1: every tile fills the table)."""
    import subprocess
    import sys
    from strainscan_amd import _lib
    tiny = os.path.join(os.path.dirname(_lib.LIB_PATH), "libstrainscan_hip_tinyq.so")
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from strainscan_amd import _lib\n"
        "from oracle import oracle as orc\n"
        "from tests.test_scan_gpu import _adversarial_case, _dense_case, _scan_binned\n"
        "for kfa, flat in (_adversarial_case(), _dense_case(9, 20000, 20000)):\n"
        "    fq = b''.join(b'@r\\n' + r + b'\\n+\\n' + b'I' * len(r) + b'\\n' for r in flat.split(b'\\n') if r)\n"
        "    want, _ = orc.jellyfish_count(kfa, [fq], k=31, upper=True)\n"
        "    db = _lib.KmerDB.from_text(kfa, 31, True).expect_hits()\n"
        "    _scan_binned(_lib, db, flat)\n"
        "    assert np.array_equal(db.counts_rows(), want)\n"
        "    db.scan_flat(flat)\n"
        "    assert np.array_equal(db.counts_rows(), 2 * want)\n"
        "print('comb ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for env in (dict(SS_LIB=tiny), dict(SS_COMBINE="1"), dict(SS_LIB=tiny, SS_COMBINE="1")):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "comb ok" in out.stdout, (env, out.stderr[-2000:])


@pytest.mark.parametrize("shape,hit_frac,want_comb", [("sampled", 0.5, True), ("contiguous", 0.3, True), ("sampled", 0.02, False),
                                                       ("contiguous", 0.03, False)])
def test_combining_kernel_chosen_from_the_data(L, shape, hit_frac, want_comb):
    """A TREE table (no ss_db_expect_hits) under a binned read set: the first 8192 tiles of the set run through the plain
    kernel and report their found runs, the rest goes through the combining kernel when they are many (ss_mini.hip
    choose: >= 8 per tile) -- a hit-heavy tree table is no longer at the mercy of a flag only layer 2 sets.  Counts equal
    the file-order scan's whichever kernel ran (the probe tiles count for real, the main launch starts behind them), equal
    the oracle's on the first reads; one probe per (read set, table): a second scan of the same set reuses the answer, a new
    set asks again; a flagged table is never probed."""
    import torch
    import bench
    from oracle import oracle as orc
    dev = torch.device("cuda", 0)
    spec = bench.make_db(torch, dev, 23, seed=77, lo_sites=2000, hi_sites=9000, shape=shape, hit_frac=hit_frac)
    n_reads = 450_000                                           # 68 M positions = 68 K tiles: more than four probes' worth
    reads = bench.make_reads(torch, dev, spec, n_reads, seed=5, hit_frac=hit_frac)
    db = L.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True)
    assert db.probe_info() == dict(set=0, comb=False, runs_per_tile=0.0)
    db.scan_flat_dev(reads.data_ptr(), reads.numel())
    L.check(L.lib().ss_device_sync(), "sync")
    want = db.counts_rows()
    assert db.probe_info()["set"] == 0                          # file order: nothing to decide
    rs = L.ReadSet.from_flat_dev(reads.data_ptr(), reads.numel(), order=True)
    for rep in range(2):
        db.reset()
        rs.scan_into(db)
        L.check(L.lib().ss_device_sync(), "sync")
        assert np.array_equal(db.counts_rows(), want), (shape, hit_frac, rep)
        pi = db.probe_info()
        assert pi["set"] != 0 and pi["comb"] is want_comb, pi
        assert (pi["runs_per_tile"] >= 8.0) is want_comb
        if rep == 0:
            first = pi
        else:
            assert pi == first                                  # the second scan of the set did not ask again
    rs2 = L.ReadSet.from_flat_dev(reads.data_ptr(), reads.numel(), order=True)
    db.reset()
    rs2.scan_into(db)
    L.check(L.lib().ss_device_sync(), "sync")
    assert np.array_equal(db.counts_rows(), want) and db.probe_info()["set"] not in (0, first["set"])
    # the oracle on the first reads, through a set small enough that NO probe runs (fewer than four probes' worth of tiles)
    n_s = 40_000
    got = orc.count_flat(spec["okeys"], 31, reads[: n_s * 151].cpu().numpy(), 4)
    sub = L.ReadSet.from_flat_dev(reads.data_ptr(), n_s * 151, order=True)
    before = db.probe_info()
    db.reset()
    sub.scan_into(db)
    L.check(L.lib().ss_device_sync(), "sync")
    assert np.array_equal(db.counts_rows(), got) and db.probe_info() == before
    # a flagged table: the hint decides, nobody asks
    db2 = L.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True).expect_hits()
    rs.scan_into(db2)
    L.check(L.lib().ss_device_sync(), "sync")
    assert np.array_equal(db2.counts_rows(), want) and db2.probe_info()["set"] == 0
    for x in (rs, rs2, sub, db, db2):
        x.close()


def test_scan_reads_multi_equals_single_scans(L):
    """ss_scan_reads_multi: several tables in one pass over a resident read set -- counts equal to ss_scan_reads on each
    (five tables = two launches; tables that overlap in their k-mers; with and without the hit hint; binned and file order;
    a k = 21 table in the list is scanned on its own)."""
    import torch
    cases = [_dense_case(31 + i, 30000, 1) for i in range(4)]
    kfas = [c[0] for c in cases] + [cases[0][0] + cases[2][0]]            # the fifth table holds the first and the third
    _, flat = _dense_case(31, 30000, 12000)                                # reads of the first genome ...
    flat += _dense_case(33, 30000, 6000)[1] + _dense_case(34, 30000, 3000)[1]
    d = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
    for order in (True, False):
        rs = L.ReadSet.from_flat_dev(d.data_ptr(), d.numel(), order=order)
        for hint in (True, False):
            dbs = [L.KmerDB.from_text(k, 31, True) for k in kfas]
            want = []
            for db in dbs:
                if hint:
                    db.expect_hits()
                rs.scan_into(db)
                L.check(L.lib().ss_device_sync(), "sync")
                want.append(db.counts_rows())
                db.reset()
            assert want[0].sum() > 400_000 and want[1].sum() < want[0].sum() // 10 and want[2].sum() > 100_000
            assert want[4].sum() == want[0].sum() + want[2].sum()
            rs.scan_into_many(dbs)
            L.check(L.lib().ss_device_sync(), "sync")
            for db, w in zip(dbs, want):
                assert np.array_equal(db.counts_rows(), w), (order, hint)
            # accumulate; and a flat-layout table in the list
            small = L.KmerDB.from_text(_random_db_and_reads(40, 5000, 10, k=21)[0], 21, True)
            rs.scan_into(small)
            L.check(L.lib().ss_device_sync(), "sync")
            w21 = small.counts_rows()
            small.reset()
            rs.scan_into_many([dbs[0], small, dbs[2]])
            L.check(L.lib().ss_device_sync(), "sync")
            assert np.array_equal(dbs[0].counts_rows(), 2 * want[0]) and np.array_equal(dbs[2].counts_rows(), 2 * want[2])
            assert np.array_equal(small.counts_rows(), w21) and np.array_equal(dbs[1].counts_rows(), want[1])
            with pytest.raises(L.SSError):
                rs.scan_into_many([dbs[0], dbs[0]])
            for db in dbs + [small]:
                db.close()
        rs.close()


def test_large_device_blocks_are_kept_for_the_next_request(L):
    """ss_dev_big_blocks (ss_host.hip ss::big_take / big_put): the slab of a destroyed read set of 256 MB and more stays with
    the process and becomes the next large request's memory (a fresh process is handed device memory at ~25 GB/s); a file-order
    set loaded into a kept block and a binned set placed in one count like the scan of the caller's block; small blocks go
    straight back; ss_gz_gpu_release returns what is kept."""
    import ctypes as C
    import torch
    import bench
    dev = torch.device("cuda", 0)

    def kept():
        out = (C.c_uint64 * 3)()
        L.check(L.lib().ss_dev_big_blocks(out), "ss_dev_big_blocks")
        return int(out[0]), int(out[1]), int(out[2])

    assert L.lib().ss_gz_gpu_release() == 0
    assert kept()[:2] == (0, 0)
    spec = bench.make_db(torch, dev, 23, seed=78, lo_sites=2000, hi_sites=9000, hit_frac=0.05)
    n_reads = 1_900_000                                         # 287 MB of records: above the 256 MB that are kept
    reads = bench.make_reads(torch, dev, spec, n_reads, seed=6, hit_frac=0.05)
    assert reads.numel() >= 256 << 20
    db = L.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True)
    db.scan_flat_dev(reads.data_ptr(), reads.numel())
    L.check(L.lib().ss_device_sync(), "sync")
    want = db.counts_rows()
    assert want.sum() > 0
    served0 = kept()[2]
    small = L.ReadSet.from_flat_dev(reads.data_ptr(), 151 * 10000, order=True)
    small.close()
    assert kept()[:2] == (0, 0)                                 # 1.5 MB: back to the driver
    for order in (True, False, True):
        before = kept()
        rs = L.ReadSet.from_flat_dev(reads.data_ptr(), reads.numel(), order=order)
        after = kept()
        if before[0]:
            assert after[0] == before[0] - 1 and after[2] == before[2] + 1, (order, before, after)      # the kept block was used
        db.reset()
        rs.scan_into(db)
        L.check(L.lib().ss_device_sync(), "sync")
        assert np.array_equal(db.counts_rows(), want), order
        rs.close()
        n, held, _ = kept()
        assert n == 1 and held >= reads.numel(), (order, n, held)
    assert kept()[2] == served0 + 2
    assert L.lib().ss_gz_gpu_release() == 0
    assert kept()[:2] == (0, 0)


def test_destroying_a_read_set_waits_for_the_scan_in_flight(L):
    """ss_reads_destroy's lifetime rule (include/strainscan_hip.h): a scan launched on the caller's stream and NOT waited
    for, the set destroyed at once, its 287 MB slab handed to the next load (ss_dev_big_blocks) and overwritten there with
    other reads -- the first scan's counts are still those of the first reads, bit for bit.  Before round 6 the slab went
    into the stash without any wait (hipFree used to make it) and the second load raced the first scan.  Also: the
    stash's blocks are tagged with their device and ss_dev_big_release hands them back."""
    import ctypes as C
    import torch
    import bench
    dev = torch.device("cuda", 0)

    def kept():
        out = (C.c_uint64 * 3)()
        L.check(L.lib().ss_dev_big_blocks(out), "ss_dev_big_blocks")
        return int(out[0]), int(out[1]), int(out[2])

    assert L.lib().ss_dev_big_release() == 0 and kept()[:2] == (0, 0)
    spec = bench.make_db(torch, dev, 23, seed=79, lo_sites=2000, hi_sites=9000, hit_frac=0.05)
    n_reads = 1_900_000
    reads_a = bench.make_reads(torch, dev, spec, n_reads, seed=7, hit_frac=0.3)
    reads_b = bench.make_reads(torch, dev, spec, n_reads, seed=8, hit_frac=0.002)     # next to no hits: a race shows as lost counts
    db = L.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True)
    db.scan_flat_dev(reads_a.data_ptr(), reads_a.numel())
    L.check(L.lib().ss_device_sync(), "sync")
    want = db.counts_rows()
    assert want.sum() > 1000
    side = torch.cuda.Stream(device=dev)
    for order in (False, True):
        for rep in range(3):
            db.reset()
            L.check(L.lib().ss_device_sync(), "sync")
            rs = L.ReadSet.from_flat_dev(reads_a.data_ptr(), reads_a.numel(), order=order)
            for _ in range(4):                                   # a queue of scans on a side stream, nothing waited for
                rs.scan_into(db, stream=side.cuda_stream)
            rs.close()                                           # <- must wait for them
            n, held, served = kept()
            assert n >= 1 and held >= reads_a.numel()
            rs2 = L.ReadSet.from_flat_dev(reads_b.data_ptr(), reads_b.numel(), order=False)      # takes the kept slab, overwrites it
            assert kept()[2] == served + 1
            L.check(L.lib().ss_device_sync(), "sync")
            assert np.array_equal(db.counts_rows(), 4 * want), (order, rep)
            rs2.close()
    assert kept()[0] >= 1
    assert L.lib().ss_dev_big_release() == 0 and kept()[:2] == (0, 0)


def _order_counters(L):
    import ctypes as C
    out = (C.c_uint64 * 2)()
    L.check(L.lib().ss_reads_order_counters(out), "ss_reads_order_counters")
    return int(out[0]), int(out[1])


@pytest.mark.parametrize("length,n_rec", [(150, 20001), (151, 4097), (32, 70000), (33, 64), (100, 1), (250, 12345), (1023, 3000), (31, 5000), (1024, 700)])
def test_binning_of_records_of_one_length(L, length, n_rec):
    """ss_reorder.hip count_fixed_kernel / place_fixed_kernel: a slab whose records all have the length of the first goes through
    the passes that KNOW where records begin (round 6: 3.7 -> ~2 ms per 20 M reads); the result obeys the same contract as
    the general passes' -- every record once, whole, in bin order, slots of 8-byte multiples, counts bit-exact.  Lengths
    32..1023 qualify (31 and 1024 take the general passes); record counts that are not multiples of 64, a single record,
    trailing newline padding, records with N / lower case in their first k-mer."""
    import torch
    from oracle import oracle as orc
    rs = np.random.RandomState(length * 7 + n_rec)
    lut = np.frombuffer(b"ACGT", np.uint8)
    g = lut[rs.randint(0, 4, size=60000 + length)]
    starts = rs.randint(0, 60000, size=n_rec)
    arr = np.stack([g[s:s + length] for s in starts]) if n_rec < 30000 else g[starts[:, None] + np.arange(length)[None, :]]
    arr = arr.copy()
    for i in range(0, n_rec, 211):
        arr[i, rs.randint(0, length)] = ord("N")
    for i in range(5, n_rec, 503):
        arr[i] = np.frombuffer(arr[i].tobytes().lower(), np.uint8)
    recs = [a.tobytes() for a in arr]
    kfa = b"".join(b">1\n" + g[i:i + 31].tobytes() + b"\n" for i in range(0, 60000, 7))
    db = L.KmerDB.from_text(kfa, 31, True)
    for pad in (0, 9):
        block = b"\n".join(recs) + b"\n" * (1 + pad)
        db.reset()
        db.scan_flat(block)
        want = db.counts_rows().copy()
        d = torch.frombuffer(bytearray(block), dtype=torch.uint8).cuda()
        f0, g0 = _order_counters(L)
        rset = L.ReadSet.from_flat_dev(d.data_ptr(), d.numel(), order=True)
        f1, g1 = _order_counters(L)
        qualifies = 32 <= length <= 1023 and len(block) >= 64
        assert (f1 - f0, g1 - g0) == ((1, 0) if qualifies else (0, 1)), (length, n_rec, pad)
        db.reset()
        rset.scan_into(db)
        L.check(L.lib().ss_device_sync(), "sync")
        assert np.array_equal(db.counts_rows(), want)
        slots = rset.read_back()
        back = [r for r in slots.split(b"\n") if r]
        assert sorted(back) == sorted(recs)
        assert len(slots) % 16 == 0 and slots.endswith(b"\n")
        if length >= 31:
            bits = 12
            while bits < 22 and (len(block) // 152) >> (bits + 2):
                bits += 1
            bins = [_locality_bin(r, bits) for r in back]
            assert bins == sorted(bins)
        rset.close()
    db.close()


def test_binning_falls_back_when_one_record_differs(L):
    """The count pass for records of one length CHECKS the slab: one record a base shorter (so that every later record is
    shifted), a newline in the middle of a record (the byte count still divides), a record one base longer at the very end,
    an empty record -- each sends the slab through the general passes, with the same result as ever.  And SS_ORDER_FIXED
    is honoured (read once per process: checked in a child)."""
    import torch
    rs = np.random.RandomState(77)
    lut = np.frombuffer(b"ACGT", np.uint8)
    g = lut[rs.randint(0, 4, size=50150)]
    base = [g[s:s + 150].tobytes() for s in rs.randint(0, 50000, size=9000)]
    kfa = b"".join(b">1\n" + g[i:i + 31].tobytes() + b"\n" for i in range(0, 50000, 5))
    db = L.KmerDB.from_text(kfa, 31, True)
    variants = {}
    v = list(base); v[4000] = v[4000][:-1]; variants["shorter"] = v
    v = list(base); v[4000] = v[4000][:75] + b"\n" + v[4000][76:]; variants["split"] = v
    v = list(base); v[-1] = v[-1] + b"A"; variants["longer_last"] = v
    v = list(base); v[8999] = v[8999][:149]; v.append(b"A"); variants["short_then_one"] = v
    v = list(base); v[100] = b""; variants["empty"] = v
    for name, recs in variants.items():
        block = b"\n".join(recs) + b"\n"
        db.reset()
        db.scan_flat(block)
        want = db.counts_rows().copy()
        d = torch.frombuffer(bytearray(block), dtype=torch.uint8).cuda()
        f0, g0 = _order_counters(L)
        rset = L.ReadSet.from_flat_dev(d.data_ptr(), d.numel(), order=True)
        f1, g1 = _order_counters(L)
        assert (f1 - f0, g1 - g0) == (0, 1), name
        db.reset()
        rset.scan_into(db)
        L.check(L.lib().ss_device_sync(), "sync")
        assert np.array_equal(db.counts_rows(), want), name
        back = [r for r in rset.read_back().split(b"\n") if r]
        assert sorted(back) == sorted(r for chunk in recs for r in chunk.split(b"\n") if r), name
        rset.close()
    db.close()


@pytest.mark.parametrize("shape", ["sampled", "dense", "repeats"])
def test_any_k_kernel_equals_tuned_kernel_at_31(L, shape):
    """ss_test_hook(4): the any-k kernel of the page index (scan_minik_kernel, one lane per start position) scans a k = 31
    table too -- the SAME index image, the same reads, through both kernels: equal counters slot for slot.  Sampled node
    sets (inline k-mers, no Bloom filter), a dense table with a Bloom filter (bucket references, super-k-mers), and a
    table full of repeats (several k-mers per minimizer offset: the `multi` buckets, low-complexity minimizers)."""
    import torch
    from oracle import oracle as orc
    if shape == "sampled":
        kfa, flat = _random_db_and_reads(501, 200000, 40000)
    elif shape == "dense":
        kfa, flat = _dense_case(seed=6, G=60000, n_reads=20000)
    else:
        rs = np.random.RandomState(9)
        unit = bytes(np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, size=40)])
        g = (unit * 40 + b"A" * 80 + b"ACAC" * 30 + unit[::-1] * 20)
        g = g + bytes(np.frombuffer(b"ACGT", np.uint8)[rs.randint(0, 4, size=5000)]) + g[:700]
        kfa = b"".join(b">1\n" + g[i:i + 31] + b"\n" for i in range(len(g) - 30))
        flat = b"\n".join(g[s:s + 150] for s in rs.randint(0, len(g) - 150, size=4000)) + b"\n"
    db = L.KmerDB.from_text(kfa, 31, True)
    assert db.info()["layout"] == 1
    fq = b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in flat.split(b"\n") if r)
    want, _ = orc.jellyfish_count(kfa, [fq], k=31, upper=True)
    d = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
    got = {}
    try:
        for hook in (0, 1):
            L.check(L.lib().ss_test_hook(4, hook), "ss_test_hook")
            for off in (0, 3):                                     # aligned and unaligned block
                t = torch.zeros(d.numel() + 16, dtype=torch.uint8, device="cuda")
                t[off:off + d.numel()] = d
                db.reset()
                torch.cuda.synchronize()
                db.scan_flat_dev(t.data_ptr() + off, d.numel(), torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                got[(hook, off)] = db.counts_rows().copy()
    finally:
        L.check(L.lib().ss_test_hook(4, 0), "ss_test_hook")
    for key, c in got.items():
        assert np.array_equal(c, want), (shape, key)
    db.close()


@pytest.mark.parametrize("k", [21, 25, 27])
def test_any_k_index_image_and_resident_reads(L, k, tmp_path):
    """k other than 31 on the page index through the rest of the ABI: the index image round trip (ss_db_export / ss_db_import),
    a binned resident read set, several tables in one call (a k = 25 table beside k = 31 ones), a dense table with its Bloom
    filter -- counts equal to the oracle's every time."""
    import torch
    from oracle import oracle as orc
    rs = np.random.RandomState(k)
    lut = np.frombuffer(b"ACGT", np.uint8)
    g = lut[rs.randint(0, 4, size=80000)].tobytes()
    kfa = b"".join(b">1\n" + g[i:i + k] + b"\n>1\n" + synth.revcomp(g[i:i + k]) + b"\n" for i in range(0, 70000))
    recs = []
    for s_ in rs.randint(0, 80000 - 150, size=15000):
        r = g[s_:s_ + int(rs.randint(k, 151))]
        recs.append(synth.revcomp(r) if rs.random_sample() < 0.5 else r)
    flat = b"\n".join(recs) + b"\n"
    fq = b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in recs)
    want, _ = orc.jellyfish_count(kfa, [fq], k=k, upper=True)
    db = L.KmerDB.from_text(kfa, k, True)
    info = db.info()
    assert info["layout"] == 1
    db.scan_flat(flat)
    assert np.array_equal(db.counts_rows(), want)
    img = str(tmp_path / "k.img")
    db.export(img)
    db2 = L.KmerDB.from_image(img)
    assert db2.info()["k"] == k
    d = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
    rset = L.ReadSet.from_flat_dev(d.data_ptr(), d.numel(), order=True)
    rset.scan_into(db2)
    L.check(L.lib().ss_device_sync(), "sync")
    assert np.array_equal(db2.counts_rows(), want)
    kfa31, _ = _random_db_and_reads(77, 30000, 10)
    db31 = L.KmerDB.from_text(kfa31, 31, True)
    db.reset()
    db2.reset()
    rset.scan_into_many([db31, db, db2])
    L.check(L.lib().ss_device_sync(), "sync")
    assert np.array_equal(db.counts_rows(), want) and np.array_equal(db2.counts_rows(), want)
    rset.close()
    for x in (db, db2, db31):
        x.close()


def _dense_case_k(k, seed=5, G=40000, n_reads=12000):
    """_dense_case at any k: EVERY k-mer of a genome, both orientations, reads at high coverage with errors."""
    rs = np.random.RandomState(seed)
    lut = np.frombuffer(b"ACGT", np.uint8)
    ga = lut[rs.randint(0, 4, size=G + k - 1)]
    g = ga.tobytes()
    kfa = b"".join(b">1\n" + g[i:i + k] + b"\n>1\n" + synth.revcomp(g[i:i + k]) + b"\n" for i in range(G))
    recs = []
    for s in rs.randint(0, G - 150, size=n_reads):
        r = ga[s:s + rs.randint(k, 151)].copy()
        m = rs.random_sample(r.size) < 0.01
        r[m] = lut[rs.randint(0, 4, size=int(m.sum()))]
        b = r.tobytes()
        recs.append(synth.revcomp(b) if rs.random_sample() < 0.5 else b)
    return kfa, b"\n".join(recs) + b"\n"


@pytest.mark.parametrize("k", [17, 20, 25, 30])
def test_tuned_kernel_at_run_time_k(L, k):
    """scan_mini_kernel with k at run time (KK = 0: the windows of k - 14 m-mers by a doubling network, the validity of k bases,
    flanks of k - 15 bases, the combining table and the solid buckets' stretches of d + k bases; at k = 17 / 20 its run queues
    overflow on every tile: the in-place path) against the oracle AND against the one-lane-per-position kernel on the same index
    (ss_test_hook(4, ...): 0 = the product's choice -- the run-queue kernel for flagged tables under binned reads at k >= 25, the
    per-position kernel otherwise --, 2 = always the per-position kernel, 3 = always the run-queue kernel): a dense table (every k-mer of a genome: buckets, solid
    runs), a table of repeats (several k-mers per offset), a sparse one (inline k-mers); plain scans in file order, binned
    resident reads with ss_db_expect_hits (the combining kernel), several tables of this k in one pass beside a k = 31 one."""
    import torch
    from oracle import oracle as orc
    rs = np.random.RandomState(100 + k)
    lut = np.frombuffer(b"ACGT", np.uint8)
    cases = {}
    cases["dense"] = _dense_case_k(k, seed=k, G=50000, n_reads=15000)
    unit = bytes(lut[rs.randint(0, 4, size=37)])
    g = unit * 40 + b"A" * 90 + b"ACAC" * 30 + b"ACG" * 40 + unit[::-1] * 20
    g = g + bytes(lut[rs.randint(0, 4, size=6000)]) + g[:800]
    cases["repeats"] = (b"".join(b">1\n" + g[i:i + k] + b"\n" for i in range(len(g) - k + 1)),
                        b"\n".join(g[s:s + 150] for s in rs.randint(0, len(g) - 150, size=5000)) + b"\n")
    cases["sparse"] = _random_db_and_reads(700 + k, 150000, 30000, k=k)
    dbs, wants, flats = {}, {}, {}
    for name, (kfa, flat) in cases.items():
        fq = b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in flat.split(b"\n") if r)
        wants[name], _ = orc.jellyfish_count(kfa, [fq], k=k, upper=True)
        dbs[name] = L.KmerDB.from_text(kfa, k, True)
        assert dbs[name].info()["layout"] == 1
        flats[name] = flat
    try:
        for name, db in dbs.items():
            d = torch.frombuffer(bytearray(flats[name]), dtype=torch.uint8).cuda()
            for hook in (0, 2, 3):                            # the product's choice, the per-position kernel, the run-queue kernel
                L.check(L.lib().ss_test_hook(4, hook), "ss_test_hook")
                for off in (0, 5):
                    tbuf = torch.zeros(d.numel() + 16, dtype=torch.uint8, device="cuda")
                    tbuf[off:off + d.numel()] = d
                    db.reset()
                    torch.cuda.synchronize()
                    db.scan_flat_dev(tbuf.data_ptr() + off, d.numel(), torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    assert np.array_equal(db.counts_rows(), wants[name]), (k, name, hook, off)
                # binned resident reads: unflagged (plain kernel), then flagged (hook 3 and, at k >= 25, the product: the combining kernel)
                rset = L.ReadSet.from_flat_dev(d.data_ptr(), d.numel(), order=True)
                for flag in (False, True):
                    db.expect_hits(flag)
                    db.reset()
                    rset.scan_into(db)
                    L.check(L.lib().ss_device_sync(), "sync")
                    assert np.array_equal(db.counts_rows(), wants[name]), (k, name, hook, "binned", flag)
                    db.scan_flat_dev(d.data_ptr(), d.numel(), None)              # file order under the flag: accumulates
                    L.check(L.lib().ss_device_sync(), "sync")
                    assert np.array_equal(db.counts_rows(), 2 * wants[name]), (k, name, hook, "file order, flagged", flag)
                rset.close()
            L.check(L.lib().ss_test_hook(4, 0), "ss_test_hook")
        # several tables in one pass: the three of this k (two flagged) + a k = 31 table, over the dense case's reads
        kfa31, _ = _random_db_and_reads(55, 20000, 10)
        db31 = L.KmerDB.from_text(kfa31, 31, True)
        d = torch.frombuffer(bytearray(flats["dense"]), dtype=torch.uint8).cuda()
        fq = b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in flats["dense"].split(b"\n") if r)
        want_on_dense = {name: orc.jellyfish_count(cases[name][0], [fq], k=k, upper=True)[0] for name in cases}
        want31, _ = orc.jellyfish_count(kfa31, [fq], k=31, upper=True)
        for binned in (True, False):
            rset = L.ReadSet.from_flat_dev(d.data_ptr(), d.numel(), order=binned)
            for all_flagged in (True, False):
                for name, db in dbs.items():
                    db.expect_hits(all_flagged or name == "dense")
                    db.reset()
                db31.reset()
                rset.scan_into_many([dbs["dense"], db31, dbs["repeats"], dbs["sparse"]])
                L.check(L.lib().ss_device_sync(), "sync")
                for name, db in dbs.items():
                    assert np.array_equal(db.counts_rows(), want_on_dense[name]), (k, name, "multi", binned, all_flagged)
                assert np.array_equal(db31.counts_rows(), want31)
            rset.close()
        db31.close()
    finally:
        L.check(L.lib().ss_test_hook(4, 0), "ss_test_hook")
        for db in dbs.values():
            db.close()

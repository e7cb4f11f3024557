#!/usr/bin/env python3
"""bench.py -- reads/s of the identification hot path on MI355X.

Workload (BASELINE.json configs[1]): synthetic E. coli-shaped database -- 823 leaf clusters =
1645 cluster-search-tree nodes, each node owning U[1000, 30000] k-mer rows (both orientations),
about 25 M rows in all -- and 10 M pairs = 20 M synthetic 150-bp reads of a 70/20/10 three-strain
mix (strand 50/50, 0.5 % substitutions), all resident in HBM before the timed region.

--db-shape sampled (default) draws every node's rows as a random 1.5-10 % sample of a longer stretch, both
orientations independently, rows scattered over kmer.fa -- what the reference's builder writes for a
node above its cap (Build_tree.py:590-591); --db-shape contiguous keeps every k-mer of a stretch.

One step = one pass of the hot path over the whole read batch on each GPU, over the sample as the
PRODUCT keeps it resident (records binned by the minimizer of their first k-mer at load time,
ss_reorder.hip; `prepare` reports that once-per-sample cost and the rates including it):
    reset counters -> encode+probe+count kernel over the resident read set -> harvest (non-zero
    counters to their node-list positions) -> [N > 1: RCCL exchange of the touched nodes' counts]
    -> per-node reductions (length / covered / outlier-cut sums for all 1645 nodes).
`value` = reads of all ranks / max-over-ranks step time.  `file_order` = the same steps over the
flat block in file order (--no-readset makes that the headline).  Weak scaling: every rank scans its
own 20 M-read shard (reads shard, the table is replicated).  `python bench.py --gpus N` starts the
N ranks itself; under torchrun it checks WORLD_SIZE == N; at N > 1 `check.parity_across_ranks`
compares the exchanged node statistics of a sample of every rank's reads with the oracle.

Extra objects on the JSON line:
  roofline     -- the scan kernel against the HBM roof: algorithmic bytes per launch
                  (150 B bases + 120 probes x 8 B per read) / its average duration
                  from HIP events on the launch stream.
  cpu_baseline -- the oracle's flat-stream counter (oracle/ss_oracle.c, OpenMP) on a bounded
                  sample of the same reads against the same table, on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
RANDOM_SECTOR_PEAK_G = 55.0    # measured: random 64-byte sector reads, G/s, footprint >= 64 MB (scripts/micro/randsec.hip)
READ_LEN = 150
K = 31
BYTES_PER_READ = READ_LEN + (READ_LEN - K + 1) * 8   # 1110 B algorithmic


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def _kmer_keys(torch, codes, start, dev):
    """device-convention keys (A0 C1 T2 G3, first base LSB) and oracle-convention keys (A0 C1 G2 T3, first base
    MSB) of the k-mers starting at `start`, and of their reverse complements"""
    n = start.numel()
    key = torch.zeros(n, dtype=torch.int64, device=dev)
    rc = torch.zeros(n, dtype=torch.int64, device=dev)
    okey = torch.zeros(n, dtype=torch.int64, device=dev)
    orc_ = torch.zeros(n, dtype=torch.int64, device=dev)
    to_or = torch.tensor([0, 1, 3, 2], device=dev, dtype=torch.int64)
    for j in range(K):
        cj = codes[start + j].to(torch.int64)
        key |= cj << (2 * j)
        rc |= (cj ^ 2) << (2 * (K - 1 - j))
        oj = to_or[cj]
        okey |= oj << (2 * (K - 1 - j))
        orc_ |= (3 - oj) << (2 * j)
    return key, rc, okey, orc_


def make_db(torch, dev, n_leaves, seed, lo_sites=500, hi_sites=15000, shape="contiguous", hit_frac=0.05):
    """Node-private random sequences -> k-mer rows.  Two shapes of the node sets:

    contiguous  every k-mer of the node's stretch, forward and reverse complement adjacent in kmer.fa (a node
                whose unique set is below the builder's cap keeps all of it: Build_tree.py:586-596)
    sampled     what the builder writes for a node ABOVE its cap (Build_tree.py:590-591:
                `kmer_t = set(random.sample(kmer_t, maxsize))`): a uniform random subset -- density 1-10 % --
                of the (k-mer, orientation) entries of a stretch 10-100x longer; forward and reverse-complement
                entries are separate ids there (Build_tree.py:101-110) and are drawn independently; kmer.fa
                is written in set order (Build_tree.py:677-684), so a node's rows are scattered over the file.

    Returns keys (device-convention uint64, numpy), oracle-convention keys, node row lists (rows, row_off),
    the stretch code tensor + offsets (the reads are cut from them) and the node count."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n_nodes = 2 * n_leaves - 1
    rs = np.random.RandomState(seed)
    sites = rs.randint(lo_sites, hi_sites + 1, size=n_nodes).astype(np.int64)   # rows per orientation
    if shape == "contiguous":
        stretch = sites
    else:
        dens = np.minimum(1.0, rs.uniform(0.3, 2.0, size=n_nodes) * max(hit_frac, 0.025))
        if shape == "mixed":                       # half of the nodes below the builder's cap: all their k-mers
            dens[rs.random_sample(n_nodes) < 0.5] = 1.0
        stretch = np.ceil(sites / dens).astype(np.int64)
    seq_len = stretch + K - 1
    seq_off = np.concatenate([[0], np.cumsum(seq_len)])
    total = int(seq_off[-1])
    codes = torch.randint(0, 4, (total,), generator=g, device=dev, dtype=torch.uint8)
    st_off = np.concatenate([[0], np.cumsum(stretch)])
    n_st = int(st_off[-1])
    if shape == "contiguous":
        site_node = np.repeat(np.arange(n_nodes), sites)
        site_pos = np.arange(n_st) - np.repeat(st_off[:-1], sites)
        start = torch.from_numpy(seq_off[site_node] + site_pos).to(dev)
        key, rc, okey, orc_ = _kmer_keys(torch, codes, start, dev)
        keys = torch.stack([key, rc], 1).reshape(-1).cpu().numpy().view(np.uint64)
        okeys = torch.stack([okey, orc_], 1).reshape(-1).cpu().numpy().view(np.uint64)
        row_off = np.concatenate([[0], np.cumsum(2 * sites)]).astype(np.uint64)
        rows = np.arange(keys.size, dtype=np.uint32)
    else:
        # Bernoulli(density) per (site, orientation): a uniform random subset, ~sites[i] rows per orientation
        node_of_site = torch.from_numpy(np.repeat(np.arange(n_nodes), stretch)).to(dev)
        d_site = torch.from_numpy(dens.astype(np.float32)).to(dev)[node_of_site]
        base = torch.from_numpy(seq_off[:-1] - st_off[:-1]).to(dev)[node_of_site] + torch.arange(n_st, device=dev)
        parts_k, parts_o, parts_n = [], [], []
        for orient in (0, 1):
            sel = torch.nonzero(torch.rand(n_st, generator=g, device=dev) < d_site).squeeze(1)
            key, rc, okey, orc_ = _kmer_keys(torch, codes, base[sel], dev)
            parts_k.append(rc if orient else key)
            parts_o.append(orc_ if orient else okey)
            parts_n.append(node_of_site[sel])
        k_all, o_all, n_all = torch.cat(parts_k), torch.cat(parts_o), torch.cat(parts_n)
        order = torch.argsort(n_all, stable=True)                       # node-major
        perm = torch.randperm(k_all.numel(), generator=g, device=dev)   # node-major position -> kmer.fa row
        keys_t = torch.empty_like(k_all)
        okeys_t = torch.empty_like(o_all)
        keys_t[perm] = k_all[order]
        okeys_t[perm] = o_all[order]
        keys = keys_t.cpu().numpy().view(np.uint64)
        okeys = okeys_t.cpu().numpy().view(np.uint64)
        rows = perm.cpu().numpy().astype(np.uint32)
        per_node = np.bincount(n_all.cpu().numpy(), minlength=n_nodes)
        row_off = np.concatenate([[0], np.cumsum(per_node)]).astype(np.uint64)
        del node_of_site, d_site, base, k_all, o_all, n_all, keys_t, okeys_t
    # balanced binary tree in heap order: node 0 root, children 2i+1, 2i+2; leaves are the last n_leaves
    return dict(keys=keys, okeys=okeys, rows=rows, row_off=row_off, sites=sites, seq_off=seq_off, codes=codes,
                n_nodes=n_nodes, shape=shape)


def path_genome(torch, dev, db, leaf, hit_frac, g):
    """A strain genome of the leaf `leaf` (heap index): the node sequences on its root->leaf path (these are the reads' database
    hits) + private filler so that about `hit_frac` of the read k-mers are database k-mers, as for a 5 Mb genome against its
    ~1e5 path k-mers; the core pieces are interleaved into the filler so that the hits are spread over the genome."""
    path = []
    i = int(leaf)
    while True:
        path.append(i)
        if i == 0:
            break
        i = (i - 1) // 2
    parts = [db["codes"][int(db["seq_off"][p]):int(db["seq_off"][p + 1])] for p in path[::-1]]
    core = torch.cat(parts)
    n_db = sum(int(db["sites"][p]) for p in path)                 # database k-mers per orientation on the path
    n_fill = max(0, int(n_db / hit_frac) - int(core.numel()))
    filler = torch.randint(0, 4, (n_fill,), generator=g, device=dev, dtype=torch.uint8)
    chunks = list(torch.tensor_split(filler, len(parts)))
    return torch.cat([x for pair in zip(chunks, parts) for x in pair])


def reads_of(torch, dev, genomes, counts, g):
    """counts[i] reads of READ_LEN bases from genomes[i] (code tensors 0..3): uniform starts, strand 50/50, 0.5 % substitutions;
    the strains interleaved (a FASTQ is not sorted by source genome).  -> flat uint8 block, one record per READ_LEN + 1 bytes."""
    n_reads = int(sum(int(c) for c in counts))
    asc = torch.tensor([65, 67, 84, 71], dtype=torch.uint8, device=dev)      # device codes 0..3 -> A C T G
    out = torch.empty(n_reads * (READ_LEN + 1), dtype=torch.uint8, device=dev)
    view = out.view(n_reads, READ_LEN + 1)
    ar = torch.arange(READ_LEN, device=dev)
    row = 0
    chunk = 1 << 20
    for gi, cnt in enumerate(counts):
        gen = genomes[gi]
        done = 0
        while done < cnt:
            m = int(min(chunk, cnt - done))
            st = torch.randint(0, gen.numel() - READ_LEN, (m,), generator=g, device=dev)
            c = gen[st[:, None] + ar[None, :]]
            err = torch.rand((m, READ_LEN), generator=g, device=dev) < 0.005
            c = torch.where(err, torch.randint(0, 4, (m, READ_LEN), generator=g, device=dev, dtype=torch.uint8), c)
            rev = torch.rand((m,), generator=g, device=dev) < 0.5
            c = torch.where(rev[:, None], c.flip(1) ^ 2, c)
            view[row:row + m, :READ_LEN] = asc[c.long()]
            row += m
            done += m
    view[:, READ_LEN] = 10
    perm = torch.randperm(n_reads, generator=g, device=dev)
    return view[perm].contiguous().view(-1)


def make_reads(torch, dev, db, n_reads, seed, hit_frac, mix=(0.7, 0.2, 0.1)):
    """Reads of a three-strain mix (path_genome of three random leaves, reads_of)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rs = np.random.RandomState(seed)
    n_nodes = db["n_nodes"]
    leaves = rs.choice(np.arange(n_nodes // 2, n_nodes), size=len(mix), replace=False)
    genomes = [path_genome(torch, dev, db, leaf, hit_frac, g) for leaf in leaves]
    counts = (np.array(mix) * n_reads).astype(np.int64)
    counts[0] += n_reads - counts.sum()
    return reads_of(torch, dev, genomes, counts, g)


def heap_to_id(h, C):
    """make_db numbers nodes in heap order (0 root, children 2h+1, 2h+2); Build_tree.py numbers leaves
    1..C, the root C+1 and internal nodes after it, every parent before its internal children."""
    return C + 1 + h if h < C - 1 else h - (C - 1) + 1


def write_tree_files(db_spec, C, tdir, multi=None):
    """The small text files of a Tree_database (tree_structure.txt, node_length.txt, reconstructed_nodes.txt,
    hclsMap_95_recls.txt; Build_tree.py:494-526,664-673) for the synthetic tree -- what the host walk reads.
    multi: {leaf id: [strain names]} -- clusters of several strains (no 4th column in tree_structure.txt)."""
    multi = multi or {}
    os.makedirs(os.path.join(tdir, "overlapping_info"), exist_ok=True)
    n_nodes = db_spec["n_nodes"]
    per_node = np.diff(db_spec["row_off"].astype(np.int64))
    ids = [heap_to_id(h, C) for h in range(n_nodes)]
    with open(os.path.join(tdir, "tree_structure.txt"), "w") as f, open(os.path.join(tdir, "node_length.txt"), "w") as g:
        for i in sorted(ids):
            h = i - (C + 1) if i > C else i - 1 + (C - 1)
            par = "N" if h == 0 else str(heap_to_id((h - 1) // 2, C))
            ch = "N" if h >= C - 1 else "%d %d" % tuple(sorted((heap_to_id(2 * h + 1, C), heap_to_id(2 * h + 2, C))))
            f.write("%d\t%s\t%s\t%s\n" % (i, par, ch, "strain_%d" % i if h >= C - 1 and i not in multi else ""))
            g.write("%d\t%d\n" % (i, int(per_node[h])))
    open(os.path.join(tdir, "reconstructed_nodes.txt"), "w").close()
    with open(os.path.join(tdir, "hclsMap_95_recls.txt"), "w") as f:
        for leaf in range(1, C + 1):
            names = multi.get(leaf) or ["strain_%d" % leaf]
            f.write("%d\t%d\t%s\n" % (leaf, len(names), ",".join(names)))


def write_fastq(reads_dev, n_reads, path, noisy_quality_seed=None):
    """Four-line FASTQ of the block's reads; quality 'I' throughout, or (for the .gz leg, where the quality line is
    half of what the inflater decodes) Phred values that fall off along the read with per-base noise."""
    reads = reads_dev.view(n_reads, READ_LEN + 1)[:, :READ_LEN].cpu().numpy()
    rec = np.empty((n_reads, 2 * READ_LEN + 7), np.uint8)
    rec[:, 0:2] = np.frombuffer(b"@r", np.uint8); rec[:, 2] = 10
    rec[:, 3:3 + READ_LEN] = reads
    rec[:, 3 + READ_LEN] = 10; rec[:, 4 + READ_LEN] = ord("+"); rec[:, 5 + READ_LEN] = 10
    if noisy_quality_seed is None:
        rec[:, 6 + READ_LEN:6 + 2 * READ_LEN] = ord("I")
    else:
        rs = np.random.RandomState(noisy_quality_seed)
        q = 38 - np.abs(rs.normal(0, 4, size=(n_reads, READ_LEN))).astype(np.int64) - (np.arange(READ_LEN) // 30)
        rec[:, 6 + READ_LEN:6 + 2 * READ_LEN] = (np.clip(q, 2, 40) + 33).astype(np.uint8)
    rec[:, 6 + 2 * READ_LEN] = 10
    rec.tofile(path)


def measure_gz_ingest(reads, n_pair, base):
    """A pair of .fastq.gz files (gzip -6, n_pair reads each) -> resident read set, with the device path (inflate +
    FASTQ extraction on the GPU, the default) and with the host inflaters (SS_GZ_GPU=0).  Outside the timed region."""
    import subprocess
    from strainscan_amd import _lib
    paths = []
    for f in range(2):
        p = os.path.join(base, "gz_%d.fq" % (f + 1))
        write_fastq(reads[f * n_pair * (READ_LEN + 1): (f + 1) * n_pair * (READ_LEN + 1)], n_pair, p, noisy_quality_seed=77 + f)
        paths.append(p)
    t0 = time.perf_counter()
    procs = [subprocess.Popen(["gzip", "-6", "-f", p]) for p in paths]
    if any(pr.wait() != 0 for pr in procs):
        return None
    gz = [p + ".gz" for p in paths]
    out = dict(reads=2 * n_pair, files=2, gz_mb=round(sum(os.path.getsize(p) for p in gz) / 1e6, 1), gzip_level=6,
               compress_s=round(time.perf_counter() - t0, 1), host_cpus=int(_lib.lib().ss_host_cpus()))
    prev = os.environ.get("SS_GZ_GPU")
    _lib.warm_up(gz=2)       # (as the CLI does on its warm-up thread when it is given .gz files: the pinned upload buffers)
    try:
        for mode, key in (("1", "device_ms"), ("0", "host_inflaters_ms")):
            os.environ["SS_GZ_GPU"] = mode
            every = []
            for _ in range(7 if mode == "1" else 3):
                t0 = time.perf_counter()
                rs = _lib.ReadSet(gz)
                _lib.check(_lib.lib().ss_device_sync(), "sync")
                dt = time.perf_counter() - t0
                n_rec = rs.info()["n_records"]
                rs.close()
                if n_rec != 2 * n_pair:
                    return None
                every.append(round(dt * 1e3, 1))
            out[key] = round(float(np.median(every)), 1)           # the MEDIAN (round 4 reported the best of seven)
            out[key + "_best"] = min(every)
            out[key + "_all"] = every
    finally:
        if prev is None:
            os.environ.pop("SS_GZ_GPU", None)
        else:
            os.environ["SS_GZ_GPU"] = prev
    out["m_reads_per_s_device"] = round(2 * n_pair / out["device_ms"] / 1e3, 1)
    out["m_reads_per_s_host_inflaters"] = round(2 * n_pair / out["host_inflaters_ms"] / 1e3, 1)
    out["note"] = ("file -> resident flat blocks, MEDIAN of 7 (device) / 3 (host) loads with every load listed, page cache warm; device = ss_ginflate.hip + ss_fastq_dev.hip (the "
                   "default; pinned upload buffers made beforehand as the CLI's warm-up thread does), host = the threaded two-pass "
                   "inflater on this box's CPUs + parse threads")
    return out


class _StatsProvider:
    """What the host walk (strainscan_amd.cst.Walk) asks of the device, answered from the bench's node statistics."""

    def __init__(self, st_np, C):
        self.st, self.C = st_np, C

    def node_stat(self, node_id):
        h = node_id - (self.C + 1) if node_id > self.C else node_id - 1 + (self.C - 1)
        s = self.st[h]
        return int(s["length"]), int(s["n_kept"]), int(s["sum_kept"])


def measure_phases(torch, dev, args, db, nodes, db_spec, reads, st_np, kern_ms, harvest_ms, reduce_ms, stream):
    """SURVEY 8(d)'s phases, OUTSIDE the timed region of `value` (rank 0, one GPU): FASTQ text -> HBM of a bounded
    sample (the product's ingest: parse threads || PCIe), the same bytes as one pinned host-to-device copy, the device
    phases from the timed steps, the host tree walk on the step's node statistics; end-to-end reads/s from text."""
    import shutil
    import tempfile
    from strainscan_amd import _lib, cst, identify
    n_s = int(min(args.reads, args.phase_reads))
    base = tempfile.mkdtemp(prefix="ss_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    ph = {}
    try:
        fq = os.path.join(base, "sample.fq")
        write_fastq(reads[: n_s * (READ_LEN + 1)], n_s, fq)
        text_bytes = os.path.getsize(fq)
        _lib.ReadSet([fq]).close()                               # first touch of the pinned buffers, page cache
        t0 = time.perf_counter()
        rs = _lib.ReadSet([fq])
        _lib.check(_lib.lib().ss_device_sync(), "sync")
        ingest_s = time.perf_counter() - t0
        db.reset(stream)
        t0 = time.perf_counter()
        rs.scan_into(db, stream)
        torch.cuda.synchronize()
        scan_sample_s = time.perf_counter() - t0
        rs.close()
        flat = torch.empty(n_s * (READ_LEN + 1), dtype=torch.uint8).pin_memory()
        dst = torch.empty_like(flat, device=dev)
        dst.copy_(flat, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dst.copy_(flat, non_blocking=True)
        torch.cuda.synchronize()
        h2d_s = time.perf_counter() - t0
        tdir = os.path.join(base, "Tree_database")
        write_tree_files(db_spec, args.leaves, tdir)
        for _ in range(2):                                       # second run: modules imported, files in the page cache
            t0 = time.perf_counter()
            res = cst.Walk(_StatsProvider(st_np, args.leaves), tdir, [0.1, 0.4, 1], identify._PARAMS, out=lambda *a: None).run()
            walk_s = time.perf_counter() - t0
        # end to end as ONE timed run over the sample: FASTQ text -> resident read set (parse threads || PCIe, binned) -> reset ->
        # scan -> harvest -> node reductions -> statistics on the host -> tree walk on THOSE statistics
        e2e = []
        stats_s = torch.zeros(db_spec["n_nodes"] * 32, dtype=torch.uint8, device=dev)
        for _ in range(4):                                       # (the first run is a warm-up; the median of the other three is reported)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rs = _lib.ReadSet([fq])
            db.reset(stream)
            rs.scan_into(db, stream)
            nodes.harvest_dev(db, stream)
            nodes.reduce_touched_dev(stats_s.data_ptr(), stream)
            st_s = stats_s.cpu().numpy().view(_lib.NODE_STAT_DTYPE)
            res_s = cst.Walk(_StatsProvider(st_s, args.leaves), tdir, [0.1, 0.4, 1], identify._PARAMS, out=lambda *a: None).run()
            e2e.append(time.perf_counter() - t0)
            rs.close()
        ph = dict(sample_reads=n_s, fastq_text_gb=round(text_bytes / 1e9, 3),
                  parse_and_h2d_ms=round(ingest_s * 1e3, 2), h2d_ms=round(h2d_s * 1e3, 2),
                  parse_note="the product's ingest parses on host threads while earlier chunks cross PCIe, then bins the records "
                             "(ss_reorder.hip): parse_and_h2d_ms is all of that; h2d_ms = the same flat bytes as one pinned copy",
                  kernel_ms=round(kern_ms, 3), gather_ms=round(harvest_ms, 3), node_reduce_ms=round(reduce_ms, 3),
                  allreduce_ms=None, l1_host_ms=round(walk_s * 1e3, 2), scan_of_sample_ms=round(scan_sample_s * 1e3, 3),
                  clusters_found=len(res), clusters_found_in_sample=len(res_s),
                  e2e_ms=round(float(np.median(e2e[1:])) * 1e3, 2), e2e_ms_all=[round(x * 1e3, 2) for x in e2e[1:]],
                  e2e_reads_per_s=round(n_s / float(np.median(e2e[1:])), 1),
                  e2e_note="whole timed runs over the sample (MEDIAN of three after a warm-up run): FASTQ text in the page cache -> "
                           "resident read set (parse threads || PCIe, binned) -> scan -> harvest -> node reductions -> host tree walk; "
                           "one GPU, in a process that has the database index on the device (a fresh `strainscan` process: cli_e2e)")
        if args.gz_reads > 0:
            ph["gz_ingest"] = measure_gz_ingest(reads, int(min(args.gz_reads, n_s) // 2), base)
    finally:
        shutil.rmtree(base, ignore_errors=True)
    return ph


def measure_config3(torch, dev, args, stream):
    """BASELINE configs[3] (one large cluster; the reference re-runs jellyfish over ALL reads for every identified cluster,
    Vote_Strain_L2_Lasso_new_sp.py:354-372, then solves it, identify_strains_L2_Enet_Pscan_new_sp.py:177-478), OUTSIDE the
    timed region of `value`:
      cluster_scan  a 10 M-row cluster table = every k-mer of a 5 Mb genome, both orientations, and 20 M resident reads of that
                    genome (600-fold, 0.5 % substitutions): scan kernel in file order and binned (ss_reorder.hip), with the
                    hits of a tile added up in LDS (ss_db_expect_hits, the product's setting) and without
      l2_solve      K = 5 M k-mers x S = 300 strains, three strains present: detect_core from the device image, phases;
                    abundances against the oracle on a sub-sample of the rows"""
    from strainscan_amd import _lib
    from oracle import oracle as orc
    out = {}
    # ---- cluster scan -------------------------------------------------------------------------------------------------
    G, n_reads = args.cluster_genome, args.reads
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    genome = torch.randint(0, 4, (G + 200,), generator=g, device=dev, dtype=torch.uint8)
    key, rc, okey, orc_ = _kmer_keys(torch, genome, torch.arange(0, G, device=dev), dev)
    keys = torch.stack([key, rc], 1).reshape(-1).cpu().numpy().view(np.uint64)
    okeys = torch.stack([okey, orc_], 1).reshape(-1).cpu().numpy().view(np.uint64)
    del key, rc, okey, orc_
    asc = torch.tensor([65, 67, 84, 71], dtype=torch.uint8, device=dev)
    reads = torch.empty((n_reads, READ_LEN + 1), dtype=torch.uint8, device=dev)
    ar = torch.arange(READ_LEN, device=dev)
    st = torch.randint(0, G, (n_reads,), generator=g, device=dev)
    for lo in range(0, n_reads, 1 << 20):
        s_ = st[lo:lo + (1 << 20)]
        c = genome[s_[:, None] + ar[None, :]]
        err = torch.rand(c.shape, generator=g, device=dev) < 0.005
        c = torch.where(err, torch.randint(0, 4, c.shape, generator=g, device=dev, dtype=torch.uint8), c)
        rev = torch.rand((s_.numel(),), generator=g, device=dev) < 0.5
        c = torch.where(rev[:, None], c.flip(1) ^ 2, c)
        reads[lo:lo + (1 << 20), :READ_LEN] = asc[c.long()]
    reads[:, READ_LEN] = 10
    flat = reads.view(-1)

    def kernel_ms(fn, db, reps=3):
        ts = []
        for _ in range(reps + 1):
            db.reset(stream)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts[1:]))

    def fracs(ms, hits):
        return dict(frac=round(n_reads * BYTES_PER_READ / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    frac_with_hit_bytes=round((n_reads * BYTES_PER_READ + 8.0 * hits) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))

    db = _lib.KmerDB(keys, np.ones(keys.size, np.uint8), K, True).expect_hits()
    t_file = kernel_ms(lambda: db.scan_flat_dev(flat.data_ptr(), flat.numel(), stream), db)
    want = db.counts_rows()
    hits = int(want.astype(np.int64).sum())
    rs = _lib.ReadSet.from_flat_dev(flat.data_ptr(), flat.numel(), order=True)
    t_bin = kernel_ms(lambda: rs.scan_into(db, stream), db)
    equal = bool(np.array_equal(db.counts_rows(), want))
    db.expect_hits(False)
    t_bin_plain = kernel_ms(lambda: rs.scan_into(db, stream), db)
    equal = equal and bool(np.array_equal(db.counts_rows(), want))
    probe_plain = db.probe_info()
    db.expect_hits(True)
    # three identified clusters: their tables in ONE pass over the reads (ss_scan_reads_multi) against a scan per table, as
    # the reference's loop does (Vote_Strain_L2_Lasso_new_sp.py:295-296); the two other tables are other genomes' k-mers
    others = []
    for sd in (4, 5):
        g2 = torch.Generator(device=dev)
        g2.manual_seed(sd)
        gen2 = torch.randint(0, 4, (G + 200,), generator=g2, device=dev, dtype=torch.uint8)
        k2, r2, _, _ = _kmer_keys(torch, gen2, torch.arange(0, G, device=dev), dev)
        kk = torch.stack([k2, r2], 1).reshape(-1).cpu().numpy().view(np.uint64)
        others.append(_lib.KmerDB(kk, np.ones(kk.size, np.uint8), K, True).expect_hits())
        del gen2, k2, r2
    three = [db] + others

    def reset_all():
        for d_ in three:
            d_.reset(stream)

    def timed(fn, reps=3):
        ts = []
        for _ in range(reps + 1):
            reset_all()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts[1:]))

    t_sep = timed(lambda: [rs.scan_into(d_, stream) for d_ in three])
    t_one = timed(lambda: rs.scan_into_many(three, stream))
    equal3 = bool(np.array_equal(db.counts_rows(), want)) and not others[0].counts_rows().any()
    for d_ in others:
        d_.close()
    # the oracle on a sub-sample (checker only)
    n_s = min(n_reads, 200_000)
    threads = max(1, min(orc.lib().orc_omp_threads(), int(_lib.lib().ss_host_cpus())))
    got = orc.count_flat(okeys, K, flat[: n_s * (READ_LEN + 1)].cpu().numpy(), threads)
    db.reset(stream)
    sub = _lib.ReadSet.from_flat_dev(flat.data_ptr(), n_s * (READ_LEN + 1), order=True)
    sub.scan_into(db, stream)
    torch.cuda.synchronize()
    parity = bool(np.array_equal(db.counts_rows(), got))
    sub.close()
    out["cluster_scan"] = dict(
        workload="cluster table of %d rows (every 31-mer of a %.0f Mb genome, both orientations), %d reads of that genome, %d-fold"
                 % (keys.size, G / 1e6, n_reads, n_reads * READ_LEN // G),
        hits=hits, hits_per_read=round(hits / n_reads, 1),
        file_order=dict(kernel_ms=round(t_file, 3), **fracs(t_file, hits)),
        binned=dict(kernel_ms=round(t_bin, 3), **fracs(t_bin, hits), m_reads_per_s=round(n_reads / t_bin / 1e3, 1)),
        binned_unflagged=dict(kernel_ms=round(t_bin_plain, 3), **fracs(t_bin_plain, hits), probe=probe_plain,
                              note="the same table without ss_db_expect_hits: the read set's first 8192 tiles report their found runs and "
                                   "the scan picks the combining kernel itself (ss_db_probe_info)"),
        three_tables=dict(one_pass_ms=round(t_one, 3), scan_per_table_ms=round(t_sep, 3), one_table_ms=round(t_bin, 3),
                          one_pass_over_one_table=round(t_one / t_bin, 3), counts_equal=equal3,
                          note="the cluster's table + two other genomes' tables of the same size, the same binned reads"),
        counters_added_per_s_binned=round(hits / (t_bin * 1e-3) / 1e9, 1), counters_unit="G hits/s",
        counts_equal_across_orders=equal, parity_on_sample=parity, parity_sample="first %d reads vs oracle orc_count_flat" % n_s,
        bound="VALU issue (3.3 G wave instructions per launch = 5.5 ms) and 148 M atomic requests of the chip's 27 G (instruction, "
              "line) per second (profiles/r04_cluster_scan2_pmc_summary.txt, r04_atomics_micro_40MB.txt)")
    rs.close()
    db.close()
    del reads, flat, genome, st
    torch.cuda.empty_cache()

    # ---- layer-2 solve ------------------------------------------------------------------------------------------------
    import contextlib
    import io
    import scipy.sparse as sp
    from strainscan_amd import identify_strains_L2_Enet_Pscan_new_sp as m
    from strainscan_amd import l2 as L2
    Kc, S, NSEG = args.l2_rows, args.l2_strains, 64
    rs_ = np.random.RandomState(5)
    pres = rs_.random_sample((S, NSEG)) < 0.35
    seg = rs_.randint(0, NSEG, size=Kc)
    depths = {3 % S: 30.0, 57 % S: 11.0, 120 % S: 5.0}
    lam = np.zeros(Kc)
    for s_i, d in depths.items():
        lam += pres[s_i, seg] * d
    y = rs_.poisson(lam).astype(np.int64)
    y[y == 1] = 0
    ids = ["S%03d" % i for i in range(S)]
    W = ((Kc + 31) // 32 + 3) & ~3
    seg_d = torch.from_numpy(seg).to(dev)
    pres_d = torch.from_numpy(pres).to(dev)
    wts = (1 << torch.arange(32, device=dev, dtype=torch.int64))
    planes = np.zeros(S * W, np.uint32)
    pad = W * 32 - Kc
    for s_i in range(S):
        b = pres_d[s_i][seg_d]
        b = torch.cat([b, torch.zeros(pad, dtype=torch.bool, device=dev)]).view(W, 32).to(torch.int64)
        planes[s_i * W:(s_i + 1) * W] = (b * wts).sum(1).cpu().numpy().astype(np.uint32)
    om = sp.csr_matrix(np.ones((Kc, 1), np.int8))
    npp = float(np.median(y[y != 0]) * 1000)
    walls, trace = [], {}
    for _ in range(7):
        img = L2.ClusterImage.from_planes(planes, Kc, S)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            res = m.detect_core(None, om, ids, y.copy(), K, 0, npp, npp, 0.9, [1], 0, 40, 0, 0, trace=trace, img=img)
        walls.append((time.perf_counter() - t0) * 1e3)
        img.close()
    t0 = time.perf_counter()
    L2.shuffle_split_test_bits(trace["n_rows"], 20, 0.5, 0)
    split_ms = (time.perf_counter() - t0) * 1e3
    # four clusters of this size solved at once, as vote_strain_L2_batch does on its "ss-l2" threads once cluster_counts_many
    # has every cluster's counts (Vote_Strain_L2_Lasso_new_sp.py:295-296 is a serial loop): the same bit planes, four samples
    from concurrent.futures import ThreadPoolExecutor
    ys4 = [y] + [np.where((v := rs_.poisson(lam * f)) == 1, 0, v).astype(np.int64) for f in (0.8, 1.3, 0.6)]
    npp4 = [float(np.median(v[v != 0]) * 1000) for v in ys4]

    def solve_one(i, img_):
        return m.detect_core(None, om, ids, ys4[i].copy(), K, 0, npp4[i], npp4[i], 0.9, [1], 0, 40, 0, 0, img=img_)

    walls4, res4 = [], None
    with contextlib.redirect_stdout(io.StringIO()), ThreadPoolExecutor(max_workers=4, thread_name_prefix="ss-l2") as pool4:
        for _ in range(5):                      # (the redirection is process-wide: once, around the threads)
            imgs4 = [L2.ClusterImage.from_planes(planes, Kc, S) for _ in range(4)]      # resident images, as for the one cluster above
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res4 = list(pool4.map(solve_one, range(4), imgs4))
            walls4.append((time.perf_counter() - t0) * 1e3)
            for im in imgs4:
                im.close()
        imgs4 = [L2.ClusterImage.from_planes(planes, Kc, S) for _ in range(4)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one_by_one = [solve_one(i, imgs4[i]) for i in range(4)]
        serial4_ms = (time.perf_counter() - t0) * 1e3
        for im in imgs4:
            im.close()
    same4 = all([dict(a) for a in r_a] == [dict(b) for b in r_b] for r_a, r_b in zip(res4, one_by_one))
    # sub-sample of the rows through the product and through the oracle
    Ks = min(Kc, args.l2_check_rows)
    Xs = sp.csr_matrix(pres[:, seg[:Ks]].T.astype(np.int8))
    ys = y[:Ks]
    tr2 = {}
    with contextlib.redirect_stdout(io.StringIO()):
        r2 = m.detect_core(Xs, sp.csr_matrix(np.ones((Ks, 1), np.int8)), ids, ys.copy(), K, 0, npp, npp, 0.9, [1], 0, 40, 0, 0, trace=tr2)
    cols, names, scov, sval, fsrc, depth = orc.prescan_packed(Xs, ys, ys, ids, 40 * K, 0, 0, 0)
    keep = (ys >= 0) & (ys <= npp)
    Xd = Xs[:, cols].toarray()[keep]
    al, mse = orc.enet_cv(Xd, ys[keep])
    a_, _, _ = orc.lasso_mpm(al, mse)
    coef = orc.enet_fit(Xd, ys[keep], a_)
    rel = coef / coef.sum()
    got = np.array([float(r2[0].get(n, 0.0)) for n in names])
    tm = trace.get("timing_ms", {})
    out["l2_solve"] = dict(
        workload="K = %d k-mers x S = %d strains (bit planes resident), strains at 30 / 11 / 5 fold" % (Kc, S),
        wall_ms=round(sorted(walls)[len(walls) // 2], 2), wall_ms_best=round(min(walls), 2), wall_ms_all=[round(w, 2) for w in walls],
        wall_note="median of the calls (the first one warms buffers up); phases_ms are the last call's",
        phases_ms={k_: round(v, 2) for k_, v in tm.items()},
        shuffle_split_generator_ms=round(split_ms, 2),
        four_clusters=dict(wall_ms=round(sorted(walls4[1:])[len(walls4[1:]) // 2], 2), wall_ms_all=[round(w, 2) for w in walls4],
                           one_by_one_ms=round(serial4_ms, 2), equal_to_one_by_one=bool(same4),
                           note="four samples of the same %d x %d cluster (four resident images) solved at once on four host threads, "
                                "against the same four one after the other; wall_ms = the median of the rounds after the first" % (Kc, S)),
        selected=list(res[0].keys()), rel=[round(float(v), 6) for v in res[0].values()],
        abundance_max_abs_diff=float(np.abs(got - rel).max()),
        prescan_equal=bool(names == list(r2[2].keys()) and {k_: int(v) for k_, v in r2[3].items()} == {k_: int(v) for k_, v in sval.items()}),
        check="the first %d rows through detect_core and through oracle.prescan_packed + enet_cv + lasso_mpm + enet_fit" % Ks)
    return out


def parity_across_ranks(torch, dist, dev, args, db, nodes, db_spec, reads, stream, rank, world, ssdist, n_sample=100_000):
    """N > 1: an independent correctness signal for the sharded path, after the timed region.  Every rank runs the product's
    step (binned resident set -> scan -> harvest -> exchange of the touched nodes -> node reductions) over its FIRST
    `n_sample` reads; the ranks all-gather those blocks (15 MB each); rank 0 counts the concatenation with the oracle
    (orc_count_flat: the checker, never the thing measured), reduces every node with orc_match_node and compares all
    node statistics with the exchanged ones.  -> dict on rank 0 (None elsewhere)."""
    from strainscan_amd import _lib
    n_s = int(min(args.reads, n_sample))
    block = reads[: n_s * (READ_LEN + 1)]
    rs = _lib.ReadSet.from_flat_dev(block.data_ptr(), block.numel(), order=True)
    stats = torch.zeros(db_spec["n_nodes"] * 32, dtype=torch.uint8, device=dev)
    state = dict(nodes.__dict__)
    ok = False
    for _ in range(3):                                # the first exchange of a sample may only size the buffer
        db.reset(stream)
        rs.scan_into(db, stream)
        nodes.harvest_dev(db, stream)
        pe = ssdist.exchange_touched(nodes, stream=stream)
        nodes.reduce_touched_dev(stats.data_ptr(), stream)
        torch.cuda.synchronize()
        if pe.complete():
            ok = True
            break
    rs.close()
    nodes.__dict__.update({k: v for k, v in state.items() if k == "_pack_cap"})     # the timed sample's buffer size stays
    gathered = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(gathered, block.contiguous())
    out = None
    if rank == 0:
        from oracle import oracle as orc
        t0 = time.perf_counter()
        flat = torch.cat(gathered).cpu().numpy()
        threads = max(1, min(orc.lib().orc_omp_threads(), int(_lib.lib().ss_host_cpus())))
        counts = orc.count_flat(db_spec["okeys"], K, flat, threads)
        valid = np.ones(counts.size, np.uint8)
        st = stats.cpu().numpy().view(_lib.NODE_STAT_DTYPE)
        rows, off = db_spec["rows"].astype(np.int64), db_spec["row_off"].astype(np.int64)
        bad = []
        for h in range(db_spec["n_nodes"]):
            o = orc.match_node(counts, valid, rows[off[h]:off[h + 1]])
            got = (int(st[h]["length"]), int(st[h]["n_pos"]), int(st[h]["n_kept"]), int(st[h]["sum_kept"]))
            want = (o["length"], o["n_pos"], o["n_kept"], o["sum_kept"])
            if got != want or (o["n_pos"] and int(st[h]["median2"]) != int(round(2 * o["median"]))):
                bad.append(h)
        out = dict(ok=bool(ok and not bad), reads_per_rank=n_s, ranks=world, nodes_compared=int(db_spec["n_nodes"]),
                   nodes_with_hits=int((st["n_pos"] > 0).sum()), total_hits=int(counts.astype(np.int64).sum()),
                   nodes_differing=bad[:10], exchange_complete=bool(ok), oracle_s=round(time.perf_counter() - t0, 2),
                   what="every rank: the product's step over its first %d reads; rank 0: oracle orc_count_flat + orc_match_node "
                        "over the all-gathered blocks of all ranks vs the exchanged node statistics" % n_s)
    del gathered
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=0,
                    help="reads per GPU; default: 20 M (configs[1]: 10 M pairs on one GPU) at --gpus 1, 25 M at --gpus N > 1 "
                         "(configs[2]: 200 M reads sharded over 8 GPUs = 25 M per GPU)")
    ap.add_argument("--leaves", type=int, default=823)
    ap.add_argument("--hit-frac", type=float, default=0.05)
    ap.add_argument("--db-shape", choices=("sampled", "contiguous", "mixed"), default="sampled",
                    help="node k-mer sets: a random 1-10 %% sample of a longer stretch, rows scattered over kmer.fa "
                         "(what Build_tree.py:590-591 writes for a node above its cap), every k-mer of a stretch, or half "
                         "of the nodes each way (mixed)")
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="0 = sized for ~15 s of CPU work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-phases", action="store_true", help="skip the untimed phase breakdown (text -> HBM, walk)")
    ap.add_argument("--no-readset", action="store_true", help="time the flat block in FILE order as the headline (no binned resident read set; `value` up to round 4)")
    ap.add_argument("--phase-reads", type=int, default=20_000_000, help="reads of the FASTQ sample written for the phase breakdown and "
                    "the end-to-end rate (default: the config's 20 M)")
    ap.add_argument("--gz-reads", type=int, default=1_000_000, help="reads of the .fastq.gz pair of the phase breakdown (0 = skip)")
    ap.add_argument("--no-config3", action="store_true", help="skip the cluster_scan / l2_solve blocks (BASELINE configs[3])")
    ap.add_argument("--cluster-genome", type=int, default=5_000_000, help="bases of the cluster_scan block's genome (rows = 2x)")
    ap.add_argument("--l2-rows", type=int, default=5_000_000)
    ap.add_argument("--l2-strains", type=int, default=300)
    ap.add_argument("--l2-check-rows", type=int, default=400_000)
    ap.add_argument("--no-file-order", action="store_true", help="skip the second timing over the block in file order (profiling runs: "
                    "every launch of the scan kernel is then a launch of the headline step)")
    ap.add_argument("--no-cli-e2e", action="store_true", help="skip the cli_e2e block (the whole CLI with layer 2 on the path, reduced size)")
    ap.add_argument("--calib-stream", action="store_true",
                    help="PMC calibration: every base is 'N' (the kernel only streams the block: known bytes)")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU, RCCL
    rendezvous on 127.0.0.1) BEFORE this process has touched torch or HIP, relay rank 0's JSON line and
    exit with the worst return code.  Nothing is exec'ed: the parent only waits."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].stdout.read()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.reads <= 0:
        args.reads = 20_000_000 if args.gpus == 1 else 25_000_000
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner on the C stdout of every rank
    # (flushed at exit): keep the real stdout aside for the JSON line and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("SS_BENCH_WORKER_STUB"):     # tests/test_abi_and_host.py: argument handling up to here, no GPU
        if rank == 0:
            os.write(real_stdout, (json.dumps(dict(stub=True, n_gpus=world, rank=rank, local_rank=local,
                                                   master=os.environ.get("MASTER_ADDR"), argv=list(argv),
                                                   reads_per_gpu=args.reads)) + "\n").encode())
        return

    import torch
    import torch.distributed as dist
    from strainscan_amd import _lib

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path to time")
    # SS_BENCH_SHARE_GPU=1 (tests on a one-GPU box): every rank uses device 0 and the group runs over gloo, which stages
    # GPU tensors itself (RCCL refuses two ranks on one device)
    share_gpu = bool(os.environ.get("SS_BENCH_SHARE_GPU"))
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.check(_lib.lib().ss_set_device(local), "ss_set_device")
    # SS_BENCH_FORCE_EXCHANGE=1: run the N > 1 code path (RCCL exchange of the touched nodes) in a one-rank group, to
    # test it on a single-GPU box
    self_group = world == 1 and bool(os.environ.get("SS_BENCH_FORCE_EXCHANGE") or os.environ.get("SS_BENCH_FORCE_PIPELINE"))
    if world > 1 or self_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:          # a one-rank group of its own (a launcher always sets the port): any free port,
            import socket                            # so that two such jobs on one box do not meet
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group of %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
        world = dist.get_world_size()

    t0 = time.time()
    db_spec = make_db(torch, dev, args.leaves, seed=20231013, shape=args.db_shape, hit_frac=args.hit_frac)
    n_rows = db_spec["keys"].size
    index_how = "built"
    if world > 1:
        # as the product does (strainscan_amd/db.py rank0_first): rank 0 builds the index on the host's CPUs and exports
        # the image, the others wait and import it -- not N host-side builds competing for the same cores
        img_path = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp",
                                "ss_bench_index_%s.bin" % os.environ.get("MASTER_PORT", "0"))
        if rank == 0:
            db = _lib.KmerDB(db_spec["keys"], np.ones(n_rows, np.uint8), K, True)
            db.export(img_path)
        dist.barrier()
        if rank != 0:
            db = _lib.KmerDB.from_image(img_path)
            index_how = "imported from rank 0's image"
        dist.barrier()
        if rank == 0:
            os.unlink(img_path)
    else:
        db = _lib.KmerDB(db_spec["keys"], np.ones(n_rows, np.uint8), K, True)
    info = db.info()
    rows = db_spec["rows"]
    nodes = _lib.NodeSet.__new__(_lib.NodeSet)
    import ctypes as C
    h = C.c_void_p()
    _lib.check(_lib.lib().ss_nodes_create(_lib.ptr(rows), _lib.ptr(db_spec["row_off"]), db_spec["n_nodes"],
                                          C.byref(h)), "ss_nodes_create")
    nodes._h, nodes.n_nodes, nodes.n_rows_total = h, db_spec["n_nodes"], int(db_spec["row_off"][-1])
    layout = os.environ.get("SS_LAYOUT", "mini")
    log("[bench] db: %d rows, %d distinct, %d slots (%s layout), %.2f GB on device (%.1f s)" % (
        n_rows, info["n_distinct"], info["capacity"], layout, info["device_bytes"] / 1e9, time.time() - t0))
    t0 = time.time()
    reads = make_reads(torch, dev, db_spec, args.reads, seed=2 + rank, hit_frac=args.hit_frac)
    if args.calib_stream:
        reads.view(args.reads, READ_LEN + 1)[:, :READ_LEN] = 78      # 'N'
    torch.cuda.synchronize()
    log("[bench] reads: %d x %d bp = %.2f GB in HBM (%.1f s)" % (args.reads, READ_LEN, reads.numel() / 1e9,
                                                                 time.time() - t0))

    # One step = reset the counters -> scan kernel over the sample AS THE PRODUCT KEEPS IT RESIDENT (ss_reads: records binned by
    # the minimizer of their first k-mer when the sample is loaded, ss_reorder.hip -- strainscan_amd/db.py resident_reads, what
    # every scan of identify_cluster / vote_strain_L2 runs over) -> harvest (one streaming pass over the counters: the non-zero
    # ones go to their node-list positions, their nodes are flagged) -> [N > 1: RCCL exchange of the touched nodes: flags
    # MAX-all-reduced, their segments packed, SUM-all-reduced, unpacked] -> per-node reductions.  The binning belongs to the
    # LOAD of a sample (once, whatever the number of scans): timed before the timed region and reported as `prepare`, with
    # the rate a sample of one, two and three scans sees.  The same steps over the block in FILE order (what `value` was up
    # to round 4) are timed afterwards and reported as `file_order`.
    from strainscan_amd import dist as ssdist
    nodes.bind(db)
    stats = torch.zeros(db_spec["n_nodes"] * 32, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    exchange = world > 1 or self_group
    packed = dict(n=0, cap=0, pending=[])
    binned = not args.no_readset and not args.calib_stream

    rs_loc, prep, prep_parts = None, [], []
    if binned:
        rs_all = []
        for _ in range(5):                          # first call: first touch of 3 GB of fresh device memory
            # (the five sets stay alive until all are made: freeing a 3 GB slab right before the next one is allocated --
            #  which no sample does -- made every other call wait 80-240 ms for the driver on some boxes)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rs_all.append(_lib.ReadSet.from_flat_dev(reads.data_ptr(), reads.numel(), order=True))
            torch.cuda.synchronize()
            prep.append((time.perf_counter() - t1) * 1e3)
            parts = np.zeros(3)
            _lib.check(_lib.lib().ss_reads_order_timing(_lib.ptr(parts)), "ss_reads_order_timing")
            prep_parts.append(parts)
        rs_loc = rs_all.pop()
        for r_ in rs_all:
            r_.close()

    def make_step(scan, ev, out_stats, before=None, after=None):
        def step(i=None):
            if before is not None:
                before()
            db.reset(stream)
            if i is not None:
                ev[i][0].record()
            scan()
            if i is not None:
                ev[i][1].record()
            nodes.harvest_dev(db, stream)
            if i is not None:
                ev[i][2].record()
            if exchange:
                # no host round trip inside: the packed buffer's size was decided by the previous exchange of this node set
                packed["pending"].append(ssdist.exchange_touched(nodes, stream=stream))
            if i is not None:
                ev[i][3].record()
            nodes.reduce_touched_dev(out_stats.data_ptr(), stream)
            if i is not None:
                ev[i][4].record()
            if after is not None:
                after()
        return step

    def settle_exchanges():
        """After a synchronisation: did every exchange carry all its counts?  (The first one of a node set sizes the buffer;
        NodeSet.harvest repeats such a scan's harvest + exchange, here the steps are simply run again.)"""
        ok = True
        for pe in packed["pending"]:
            ok = pe.complete() and ok
            packed["n"], packed["cap"] = pe.total(), pe.cap
        packed["pending"] = []
        return ok

    def run_timed(scan, out_stats, before=None, after=None):
        """W warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; the MAX over the ranks.
        -> (seconds, mean ms of scan kernel / harvest / exchange / node reductions from HIP events on the launch stream)"""
        ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(5)) for _ in range(args.steps)]
        step = make_step(scan, ev, out_stats, before, after)
        if exchange:
            step()
            torch.cuda.synchronize()
            settle_exchanges()                       # learns the buffer size for this sample
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if exchange and not settle_exchanges():
            raise SystemExit("bench.py: the exchange buffer did not settle during warm-up")
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        if exchange and not settle_exchanges():
            raise SystemExit("bench.py: an exchange inside the timed region did not carry all counts")
        return dt, [float(np.mean([e[j].elapsed_time(e[j + 1]) for e in ev])) for j in range(4)]

    def scan_file():
        db.scan_flat_dev(reads.data_ptr(), reads.numel(), stream)

    def scan_binned():
        rs_loc.scan_into(db, stream)

    # THE HEADLINE STEP (round 6): everything a sample's reads go through from the moment they are resident in file order --
    # the binning of the records (ss_reorder.hip: what ss_reads_load does once per sample), the tree scan of the binned set,
    # harvest, [exchange,] node reductions.  One scan per sample is the WORST case for the binning (configs[1]'s flow makes
    # two or more: tree scan + the identified clusters' scan, with -b two more); the rate over an already binned set -- `value`
    # of round 5, the limit for many scans -- is `resident_binned`, the steps in file order without any binning `file_order`.
    sample = {}

    def bin_sample():
        sample["rs"] = _lib.ReadSet.from_flat_dev(reads.data_ptr(), reads.numel(), order=True)

    def scan_sample():
        sample["rs"].scan_into(db, stream)

    def drop_sample():
        sample.pop("rs").close()                     # (waits for the step's kernels: ss_reads_destroy's lifetime rule; the slab is kept for the next step)

    resident_binned = None
    if binned:
        dt, (kern_ms, harvest_ms, exch_ms, reduce_ms) = run_timed(scan_sample, stats, bin_sample, drop_sample)
        stats_b = torch.zeros_like(stats)
        dt_b, (k_b, h_b, x_b, r_b) = run_timed(scan_binned, stats_b)
        torch.cuda.synchronize()
        resident_binned = dict(what="the same steps over a read set that was binned BEFORE the clock started (`value` of round 5): what every scan "
                                    "after a sample's first costs, the limit for many scans per sample",
                               ms_per_step=round(dt_b / args.steps * 1e3, 3), value=round(args.reads * world * args.steps / dt_b / 1e6, 3),
                               unit="M reads/s", node_stats_equal=bool(torch.equal(stats, stats_b)),
                               step_breakdown_ms=dict(scan_kernel=round(k_b, 3), harvest=round(h_b, 3),
                                                      exchange=round(x_b, 3) if exchange else None, node_reduce=round(r_b, 3)))
    else:
        dt, (kern_ms, harvest_ms, exch_ms, reduce_ms) = run_timed(scan_file, stats)
    ms_per_step = dt / args.steps * 1e3
    reads_per_s = args.reads * world * args.steps / dt
    tail_ms = exch_ms + reduce_ms
    st_np = stats.cpu().numpy().view(_lib.NODE_STAT_DTYPE)
    # the counters of the last step are still in the table (a step resets them at its start): whole-table checksum,
    # and the harvest path against the row-gather path it replaces (ss_counts_rows_dev + ss_nodes_reduce_dev)
    counts_rows = torch.zeros(n_rows, dtype=torch.int32, device=dev)
    db.counts_rows_dev(counts_rows.data_ptr(), stream)
    hits = int(counts_rows.to(torch.int64).sum().item())
    stats2 = torch.zeros_like(stats)
    if exchange:
        ssdist.allreduce_counts(counts_rows)
    nodes.reduce_dev(counts_rows.data_ptr(), db.row_valid_dev, stats2.data_ptr(), stream)
    torch.cuda.synchronize()
    harvest_equals_gather = bool(torch.equal(stats, stats2))

    def pmc_entry(key):
        """Counter-derived figures come from separate rocprofv3 --pmc passes of THIS command (scripts/gpu_round.sh writes
        profiles/pmc_traffic.json, one entry per database shape, hit fraction and read order, with the commit they were taken
        at): PMC collection serialises kernels and cannot run inside the timed region."""
        path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(path) and args.reads == 20_000_000 and args.leaves == 823:
            with open(path) as f:
                return json.load(f).get(key, {})
        return {}

    def roofline_of(k_ms, pmc, pmc_key, bytes_per_record):
        """The scan kernel against the HBM roof by SURVEY 8(d)'s model (1110 algorithmic bytes per read), with the measured
        side beside it.  The counters describe THIS kernel only while it still runs as it did when they were collected: if
        the live kernel time has moved by more than 5 % from the one recorded with them they are not reported."""
        achieved = args.reads * BYTES_PER_READ / (k_ms * 1e-3) / 1e9
        pmc_ms = pmc.get("kernel_ms_at_collection")
        stale = bool(pmc) and pmc_ms is not None and abs(k_ms - pmc_ms) > 0.05 * pmc_ms
        if stale:
            log("[bench] profiles/pmc_traffic.json[%s] was collected at %.3f ms per launch, the kernel now takes %.3f ms: counters not reported"
                % (pmc_key, pmc_ms, k_ms))
            pmc = dict(source=pmc.get("source"), commit=pmc.get("commit"))
        traffic = pmc.get("traffic_gb_per_launch")
        compulsory_gb = (args.reads * bytes_per_record + 8.0 * hits) / 1e9      # every base once + 4 B read + 4 B write per hit
        frac = achieved / HBM_PEAK_GBS
        hbm_meas = round(traffic / (k_ms * 1e-3) / HBM_PEAK_GBS, 4) if traffic else None
        rf = dict(bound="hbm", kernel="scan_mini_kernel" if layout == "mini" else "scan_kernel",
                  achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(frac, 5), traffic=traffic,
                  traffic_unit="GB per launch", kernel_ms=round(k_ms, 3), bytes_per_read=BYTES_PER_READ,
                  frac_algorithmic=round(frac, 5), hbm_frac_measured=hbm_meas, compulsory_gb=round(compulsory_gb, 3),
                  traffic_over_compulsory=(round(traffic / compulsory_gb, 2) if traffic else None),
                  valu_busy=pmc.get("valu_busy"), valu_insts_per_tile=pmc.get("valu_insts_per_tile"),
                  random_sectors=dict(
                      note="what bounds the lookups of a scan in file order: random 64-byte sectors, ~55 G/s on this chip from "
                           "any footprint beyond the L2 (scripts/micro/randsec.hip, profiles/r02_randsec.txt)",
                      peak_gsectors_s=RANDOM_SECTOR_PEAK_G, read_requests_per_launch=pmc.get("rdreq_per_launch"),
                      achieved_gsectors_s=(round(pmc["rdreq_per_launch"] / (k_ms * 1e-3) / 1e9, 1) if pmc.get("rdreq_per_launch") else None)),
                  traffic_source=(dict(file="profiles/pmc_traffic.json", key=pmc_key, summary=pmc.get("source"), commit=pmc.get("commit"),
                                       kernel_ms_at_collection=pmc_ms, stale=stale) if pmc else None))
        # SURVEY 8(d) prices a read at 120 probes x 8 B; a minimizer index answers the ~9 k-mers of a run with ONE lookup,
        # so on a fast shape the model's bytes exceed what any kernel would move and `frac` stops being a fraction of a roof
        # what the counters say bounds the kernel: with measured HBM use below half of the model's fraction the kernel does not sit on
        # the HBM roof at all -- it sits on VALU issue (valu_busy 0.93 at collection) -- and says so; frac stays SURVEY 8(d)'s model
        if hbm_meas is not None and hbm_meas < 0.5 * frac:
            rf["bound"] = "valu"
            rf["bound_note"] = ("hbm_frac_measured %.2f < 0.5 x frac %.2f: the minimizer index answers ~9 probes with one sector, so the kernel is bound by "
                                "VALU issue (valu_busy, valu_insts_per_tile), not by HBM; achieved / peak / frac are SURVEY 8(d)'s algorithmic-byte "
                                "model against the HBM peak" % (hbm_meas, frac))
        rf["model_saturated"] = bool(frac > 0.9 and (hbm_meas is None or hbm_meas < 0.6 * frac))
        if rf["model_saturated"]:
            rf["model_note"] = ("8(d)'s 1110 B per read is saturated by the minimizer index on this shape: the work is done (counts checked "
                                "in this run) with far fewer bytes; what bounds the kernel is VALU issue (valu_busy), measured HBM use is "
                                "hbm_frac_measured")
        return rf

    pmc_base = "%s:%s:%g" % (layout, args.db_shape, args.hit_frac)
    head_key = pmc_base + (":binned" if binned else "")
    roofline = roofline_of(kern_ms, pmc_entry(head_key), head_key, READ_LEN + (2 if binned else 1))
    roofline["read_order"] = "binned (the product's resident read set)" if binned else "file"

    prepare, file_order = None, None
    if binned:
        prep_ms = float(np.median(prep))
        parts_med = np.median(np.array(prep_parts), axis=0)
        step_s = dt_b / args.steps                   # a step over the binned resident set; + prepare / n for a sample of n scans

        def with_prepare(n_scans, ms=None):
            return round(args.reads * world / (step_s + (prep_ms if ms is None else ms) * 1e-3 / n_scans) / 1e6, 1)

        kern_only = float(parts_med[[0, 2]].sum())

        # (the MEDIAN of five is reported, the best and all five beside it: the call allocates the new 3 GB slab and frees 0.2 GB
        #  of scratch, and on a box whose host is busy with other tenants one such driver call now and then takes 60-150 ms)
        prepare = dict(what="binning of the resident records by the minimizer of their first k-mer, ~4 records per bin (ss_reorder.hip)",
                       charged="once per sample; INSIDE every timed step of the headline (a step = a sample scanned once), measured on its own here",
                       ms=round(prep_ms, 2), ms_best=round(float(np.min(prep)), 2), ms_first_call=round(prep[0], 2), ms_all=[round(x, 2) for x in prep],
                       ms_kernels=round(float(parts_med[[0, 2]].sum()), 2),
                       breakdown_ms=dict(zip(("count_and_prefix", "slab_allocation", "place"), [round(float(x), 2) for x in parts_med])),
                       scans_per_sample=dict(note="the tree scan, + one scan per group of <= 4 identified multi-strain clusters "
                                                  "(ss_scan_reads_multi), + 2 more with -b (identify_low_depth.py:119,124)",
                                             all_clusters_single_strain=1, one_to_four_multi_strain_clusters=2, low_depth_b=3),
                       m_reads_per_s_including_prepare={"1_scan": with_prepare(1), "2_scans": with_prepare(2), "3_scans": with_prepare(3),
                                                        "note": "resident_binned's step + prepare.ms / n; `value` is the measured 1-scan case"},
                       m_reads_per_s_including_prepare_kernels={"1_scan": with_prepare(1, kern_only), "2_scans": with_prepare(2, kern_only),
                                                                "3_scans": with_prepare(3, kern_only)},
                       driver_allocation_ms=round(float(parts_med[1]), 2), driver_allocation_slow=bool(parts_med[1] > 5.0),
                       driver_note="ms is wall time: the binning's two kernels (ms_kernels) + the driver's hipMalloc of the 3 GB output slab, which takes "
                                   "0.3 ms on most boxes and 60-150 ms on some (every call then: freshly freed device memory is handed out slowly); "
                                   "m_reads_per_s_including_prepare uses ms, ..._kernels the kernels alone",
                       policy="always (SS_READS_ORDER=file keeps the file order): binning pays from the SECOND scan of a sample on; a "
                              "sample scanned once loses prepare.ms - (file_order.ms_per_step - ms_per_step), beside a text ingest of "
                              "~80 ms for the same reads")
    if binned and not args.no_file_order:
        stats_f = torch.zeros_like(stats)
        dt_f, (k_f, h_f, x_f, r_f) = run_timed(scan_file, stats_f)
        torch.cuda.synchronize()
        file_order = dict(what="the same steps over the flat block in FILE order (`value` up to round 4; SS_READS_ORDER=file)",
                          ms_per_step=round(dt_f / args.steps * 1e3, 3), value=round(args.reads * world * args.steps / dt_f / 1e6, 3),
                          unit="M reads/s", node_stats_equal=bool(torch.equal(stats, stats_f)),
                          roofline=roofline_of(k_f, pmc_entry(pmc_base), pmc_base, READ_LEN + 1),
                          step_breakdown_ms=dict(scan_kernel=round(k_f, 3), harvest=round(h_f, 3),
                                                 exchange=round(x_f, 3) if exchange else None, node_reduce=round(r_f, 3)))
    parity_ranks = None
    if world > 1 or self_group:                  # (a one-rank group, SS_BENCH_FORCE_EXCHANGE: the same collectives over RCCL)
        parity_ranks = parity_across_ranks(torch, dist, dev, args, db, nodes, db_spec, reads, stream, rank, world, ssdist)
    if rs_loc is not None:
        rs_loc.close()
        rs_loc = None
    # what decides the END-TO-END rate of an N-GPU node (not the scan): the FASTQ parse threads each rank takes out of the CPUs
    # the ranks share (ss_ingest_threads: min(20, usable CPUs / LOCAL_WORLD_SIZE); ~21 M reads/s of plain text per thread)
    import ctypes
    nthr = ctypes.c_int()
    _lib.check(_lib.lib().ss_ingest_threads(ctypes.byref(nthr)), "ss_ingest_threads")
    host = dict(cpus_usable=int(_lib.lib().ss_host_cpus()), local_world=int(os.environ.get("LOCAL_WORLD_SIZE", "1")),
                parse_threads_per_rank=[int(nthr.value)])
    if world > 1:
        got = [None] * world
        dist.all_gather_object(got, int(nthr.value))
        host["parse_threads_per_rank"] = [int(x) for x in got]

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported at N = 1 only
        from oracle import oracle as orc
        # the CPUs this process may really use (a container's cgroup quota can be far below the hardware threads:
        # 16 CPUs on the 256-thread GPU box; more busy threads than that only get the group throttled)
        hw_threads = orc.lib().orc_omp_threads()
        threads = max(1, min(hw_threads, int(_lib.lib().ss_host_cpus())))
        flat = reads[: 20000 * (READ_LEN + 1)].cpu().numpy()
        t1 = time.perf_counter()
        orc.count_flat(db_spec["okeys"], K, flat, threads)      # includes the table build
        t_small = time.perf_counter() - t1
        t1 = time.perf_counter()
        orc.count_flat(db_spec["okeys"], K, flat[: 151 * 100], threads)
        t_build = time.perf_counter() - t1                       # ~ table build only
        rate = 20000 / max(1e-6, t_small - t_build)
        n_s = args.cpu_sample_reads or int(min(args.reads, max(20000, rate * 15)))
        flat = reads[: n_s * (READ_LEN + 1)].cpu().numpy()
        t1 = time.perf_counter()
        got = orc.count_flat(db_spec["okeys"], K, flat, threads)
        t_cpu = time.perf_counter() - t1 - t_build
        cpu = dict(value=round(n_s / t_cpu / 1e6, 4), unit="M reads/s", cores=threads, hardware_threads=hw_threads, kind="port",
                   sample="first %d reads of the rank-0 batch vs the same %d-row table; oracle/ss_oracle.c "
                          "orc_count_flat (OpenMP), table build excluded" % (n_s, n_rows))
        # the reference runs jellyfish with a fixed -t 8 (identify.py:82): the same counter on 8 threads, smaller sample
        if threads > 8:
            n8 = int(max(20000, min(n_s, n_s * 8.0 / threads * 0.5)))
            t1 = time.perf_counter()
            orc.count_flat(db_spec["okeys"], K, flat[: n8 * (READ_LEN + 1)], 8)
            t8 = time.perf_counter() - t1 - t_build
            cpu["value_8_threads"] = round(n8 / max(1e-6, t8) / 1e6, 4)
            cpu["sample_8_threads"] = "first %d reads" % n8
        # and use it as a checker on that sample (never the other way round)
        db.reset(stream)
        if binned:                                   # through what the product scans: the binned resident set of the sample
            sub = _lib.ReadSet.from_flat_dev(reads.data_ptr(), n_s * (READ_LEN + 1), order=True)
            sub.scan_into(db, stream)
            torch.cuda.synchronize()
            sub.close()
        else:
            db.scan_flat_dev(reads.data_ptr(), n_s * (READ_LEN + 1), stream)
            torch.cuda.synchronize()
        chk = db.counts_rows()
        cpu["parity_on_sample"] = bool(np.array_equal(chk, got))

    phases = None
    if rank == 0 and world == 1 and not args.no_phases and not args.calib_stream:
        phases = measure_phases(torch, dev, args, db, nodes, db_spec, reads, st_np, kern_ms, harvest_ms, reduce_ms, stream)
    elif rank == 0:
        phases = dict(kernel_ms=round(kern_ms, 3), gather_ms=round(harvest_ms, 3), allreduce_ms=round(exch_ms, 3),
                      node_reduce_ms=round(reduce_ms, 3), exchanged_counts=packed["n"], exchange_buffer_counts=packed["cap"],
                      exchange_bytes_per_rank=4 * (packed["cap"] + int(db_spec["n_nodes"])),
                      allreduce_note="flags MAX-all-reduce + pack + SUM-all-reduce of the touched nodes' counts + unpack, "
                                     "no host synchronisation inside (dist.exchange_touched)",
                      index=index_how)
    config3 = None
    if rank == 0 and world == 1 and not args.no_config3 and not args.calib_stream:
        del reads
        torch.cuda.empty_cache()
        config3 = measure_config3(torch, dev, args, stream)
    cli_e2e = None
    if rank == 0 and world == 1 and not args.no_cli_e2e and not args.calib_stream:
        # what a user runs, reduced to a few seconds: fresh `strainscan` processes on a 63-cluster tree with three multi-strain
        # clusters (layer 2 on the path), 2 M reads from text; the full configs[3] shape: scripts/bench_cli_l2.py, profiles/r05_cli_l2_*
        torch.cuda.empty_cache()
        from scripts import bench_cli_l2
        t1 = time.perf_counter()
        try:
            c = bench_cli_l2.run(2_000_000, [(400_000, 60), (200_000, 40), (100_000, 30)], "text", 63, per_cluster=False)
            cli_e2e = dict(workload="63-cluster tree, three multi-strain clusters with layer-2 k-mer sets (%s), %d reads from a FASTQ text pair"
                                    % (", ".join(c["clusters"]), c["n_reads"]),
                           fresh_process=[dict(label=r_["label"], wall_s=r_["wall_s"], phases_s=r_["phases_s"], rc=r_["rc"]) for r_ in c["cli_fresh_process"]],
                           in_process=c.get("in_process"), all_expected_strains_reported=c.get("all_expected_strains_reported"),
                           strains_in_sample=c.get("strains_in_sample"), block_s=round(time.perf_counter() - t1, 1),
                           full_size="scripts/bench_cli_l2.py (202 clusters / 1627 strains, clusters of 5 M x 300 / 2 M x 120 / 1 M x 60, 20-50 M reads, "
                                     "text and .gz): profiles/r05_cli_l2_*.json")
        except Exception as e:                      # noqa: B902 -- the headline does not depend on this block
            cli_e2e = dict(error="%s: %s" % (type(e).__name__, e))
    if rank == 0:
        out = dict(metric="M reads/sec vs 1433-strain E. coli DB", value=round(reads_per_s / 1e6, 3),
                   unit="M reads/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(ms_per_step, 3), higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="u64", data="synthetic",
                   config=dict(workload=("E. coli 1433-strain/823-cluster DB, 10M synthetic 150 bp PE reads (20M reads), 1xMI355X "
                                         "[BASELINE configs[1]]" if world == 1 and args.reads == 20_000_000 else
                                         "E. coli 1433-strain DB, %dM reads sharded %dxMI355X (%dM per GPU), RCCL all-reduce of per-node "
                                         "hit counts [BASELINE configs[2]: 200M reads over 8 GPUs = 25M per GPU]"
                                         % (args.reads * world // 1_000_000, world, args.reads // 1_000_000)),
                               db_rows=int(n_rows), tree_nodes=int(db_spec["n_nodes"]), reads_per_gpu=args.reads,
                               read_len=READ_LEN, k=K, hit_frac=args.hit_frac, db_shape=args.db_shape,
                               table_slots=int(info["capacity"]), minimizer_buckets=int(info.get("n_buckets", 0)),
                               index=dict(pages=info.get("n_dir"), bucket_slots=info.get("n_mslots"), inline_kmers=info.get("n_inline"),
                                          filter_bits=info.get("filter_bits"), device_gb=round(info["device_bytes"] / 1e9, 3)),
                               table_layout=layout,
                               read_order=("binned resident read set (what the product scans: strainscan_amd/db.py resident_reads); the binning "
                                           "itself is inside every timed step" if binned else "file order (flat block)"),
                               step=("binning of the file-order records + tree scan + harvest%s + node reductions: a sample scanned ONCE, everything "
                                     "included (resident_binned: the same without the binning; file_order: no binning at all)"
                                     % (" + exchange" if exchange else "") if binned else "tree scan + harvest + node reductions in file order"),
                               parallelism="reads sharded x%d, table replicated, all-reduce of the touched nodes' hit counts" % world),
                   roofline=roofline, cpu_baseline=cpu, phases=phases, prepare=prepare, resident_binned=resident_binned,
                   file_order=file_order, host=host,
                   cluster_scan=(config3 or {}).get("cluster_scan"), l2_solve=(config3 or {}).get("l2_solve"), cli_e2e=cli_e2e,
                   e2e_reads_per_s=(phases or {}).get("e2e_reads_per_s"),
                   check=dict(total_hits=hits, nodes_with_hits=int((st_np["n_pos"] > 0).sum()),
                              harvest_equals_gather=harvest_equals_gather, exchanged_counts=packed["n"],
                              parity_across_ranks=parity_ranks),
                   step_breakdown_ms=dict(binning=(round(ms_per_step - kern_ms - harvest_ms - exch_ms - reduce_ms, 3) if binned else None),
                                          scan_kernel=round(kern_ms, 3), harvest=round(harvest_ms, 3),
                                          exchange=round(exch_ms, 3) if exchange else None, node_reduce=round(reduce_ms, 3)))
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1 or self_group:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

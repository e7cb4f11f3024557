"""The N > 1 path on CPU: two gloo ranks shard the read blocks, count their shard (with the
oracle standing in for the device scan), and the product's all-reduce must reproduce the
single-process counts bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, kfa_path, fq_path, out_dir):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", init_method="file://" + os.path.join(out_dir, "store_%d" % port), rank=rank, world_size=world)
    from strainscan_amd import _lib
    from strainscan_amd import dist as sdist
    from oracle import oracle as orc
    kfa = open(kfa_path, "rb").read()
    rows = kfa.split(b"\n")[1::2]
    keys = np.array([orc.encode_kmer(r.decode()) for r in rows], np.uint64)
    assert sdist.is_distributed() and sdist.rank_world() == (rank, world)
    # product code: the flat-block reader (small blocks so that both ranks get several) + round robin
    blocks = list(_lib.read_flat_blocks([fq_path], cap=8192, overlap=30))
    mine = list(sdist.rank_blocks(blocks, rank, world))
    assert len(mine) in (len(blocks) // world, len(blocks) // world + 1)
    counts = np.zeros(len(keys), np.uint32)
    for blk, _ in mine:
        counts += orc.count_flat(keys, 31, blk, threads=1)
    # make the unsigned wrap visible: push one counter over 2^31 on every rank
    counts[0] += np.uint32(0x90000000)
    t = torch.from_numpy(counts.view(np.int32).copy())
    sdist.allreduce_counts(t)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), t.numpy().view(np.uint32))
    dist.destroy_process_group()


def test_two_rank_allreduce_matches_single(tmp_path, l1_dbs, l1_reads):
    from oracle import oracle as orc
    info = l1_dbs["A"]
    kfa_path = os.path.join(info["db_dir"], "Tree_database", "kmer.fa")
    fq_path, reads = l1_reads["A_mix3"]
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, kfa_path, fq_path, str(tmp_path)), nprocs=2, join=True)
    kfa = open(kfa_path, "rb").read()
    want, valid = orc.jellyfish_count(kfa, [reads], k=31, upper=True)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1)
    want = want.copy()
    want[0] = np.uint32((int(want[0]) + 2 * 0x90000000) % (1 << 32))     # uint32 wrap of the injected offsets
    # rows that are not valid (duplicates of an earlier row) still count in the flat counter:
    assert np.array_equal(r0[valid == 1], want[valid == 1])


def _share_worker(rank, world, port, gz_path, plain_path, out_dir):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_WORLD_SIZE"] = str(world)
    dist.init_process_group("gloo", init_method="file://" + os.path.join(out_dir, "store_%d" % port), rank=rank, world_size=world)
    from strainscan_amd import dist as sdist
    use, cleanup = sdist.share_inflated([gz_path, plain_path, ""])
    assert use[1] == plain_path and use[2] == ""
    assert use[0] != gz_path and os.path.exists(use[0])            # the same tmpfs file on both ranks
    with open(use[0], "rb") as f:
        text = f.read()
    with open(os.path.join(out_dir, "share%d.txt" % rank), "w") as f:
        f.write("%s %d %d\n" % (use[0], len(text), __import__("zlib").crc32(text)))
    cleanup()
    dist.barrier()
    assert not os.path.exists(use[0])                              # rank 0 removed it after everyone had read it
    os.environ["SS_GZ_SHARE"] = "0"
    assert sdist.share_inflated([gz_path])[0] == [gz_path]
    os.environ["SS_GZ_SHARE"] = "1"
    # load_agreed: every rank loads under the strict policy -- first with the file's inflation shared between the ranks
    # (range mode), then every rank inflating the whole file on its GPU --; if ANY rank was declined, all of them go on to
    # the next way, at last (or with SS_GZ_GPU=0) to the shared plain text: no rank keeps what it loaded alone
    from strainscan_amd import _lib
    calls = []

    def load(paths):
        calls.append(list(paths))
        if len(calls) <= decline_first and rank == 1:              # rank 1's device path declines
            raise _lib.SSError(_lib.SS_EAGAIN, "test")
        return ("loaded", len(calls))

    dropped = []
    os.environ["SS_GZ_GPU"] = "1"
    decline_first = 1                                              # the shared inflation fails on rank 1: whole-file device path for all
    got = sdist.load_agreed([gz_path, plain_path], load, discard=dropped.append)
    assert got == ("loaded", 2) and calls == [[gz_path, plain_path]] * 2
    assert dropped == ([] if rank == 1 else [("loaded", 1)])       # rank 0 gave up what it had loaded alone
    calls.clear()
    dropped.clear()
    decline_first = 2                                              # ... that one too: the host inflaters, one shared inflate
    got = sdist.load_agreed([gz_path, plain_path], load, discard=dropped.append)
    assert got == ("loaded", 3) and calls[0] == calls[1] == [gz_path, plain_path] and calls[2][0] != gz_path and calls[2][1] == plain_path
    assert dropped == ([] if rank == 1 else [("loaded", 1), ("loaded", 2)])
    dist.barrier()                                                 # (rank 0 removes the shared text behind load_agreed's own barrier)
    assert not os.path.exists(calls[2][0])
    calls.clear()
    os.environ["SS_GZ_RANGE"] = "0"                                # without the shared inflation: two ways left
    decline_first = 1
    got = sdist.load_agreed([gz_path, plain_path], load, discard=dropped.append)
    assert got == ("loaded", 2) and calls[1][0] != gz_path
    os.environ.pop("SS_GZ_RANGE")
    decline_first = 0
    calls.clear()
    assert sdist.load_agreed([gz_path], lambda ps: ("ok", list(ps))) == ("ok", [gz_path])        # nobody declined: one attempt
    os.environ["SS_GZ_GPU"] = "0"                                  # host inflaters asked for: shared text at once
    got = sdist.load_agreed([gz_path], lambda ps: list(ps))
    assert got != [gz_path] and got[0].startswith("/dev/shm/")
    assert sdist.load_agreed([plain_path], lambda ps: list(ps)) == [plain_path]                   # no .gz: no collective
    os.environ["SS_GZ_GPU"] = "1"
    with pytest.raises(ValueError):                                # a failure on one rank is raised on both
        def bad(ps):
            if rank == 0:
                raise ValueError("rank 0 only")
            return 1
        try:
            sdist.load_agreed([gz_path], bad)
        except RuntimeError as e:
            assert rank == 1 and "another rank" in str(e)
            raise ValueError("told")
    dist.destroy_process_group()


@pytest.mark.skipif(not os.path.isdir("/dev/shm"), reason="no tmpfs")
def test_two_ranks_share_one_inflate(tmp_path, monkeypatch):
    """dist.share_inflated, for .gz inputs that go through the host inflaters: rank 0 inflates a .gz once into /dev/shm, both
    ranks get the same plain file with the right bytes, plain inputs pass through, cleanup removes the file.
    dist.load_agreed: the ranks agree on ONE inflate path per load (strict device attempt, MIN-all-reduce, host path for
    all when any rank was declined); an exception on one rank reaches all."""
    import gzip
    import zlib
    monkeypatch.setenv("SS_GZ_GPU", "0")
    rs = np.random.RandomState(4)
    lut = np.frombuffer(b"ACGT", np.uint8)
    text = b"".join(b"@r%d\n" % i + lut[rs.randint(0, 4, size=150)].tobytes() + b"\n+\n" + b"I" * 150 + b"\n" for i in range(20000))
    gz = tmp_path / "s.fq.gz"
    gz.write_bytes(gzip.compress(text, 6))
    plain = tmp_path / "p.fq"
    plain.write_bytes(text[:5000])
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_share_worker, args=(2, port, str(gz), str(plain), str(tmp_path)), nprocs=2, join=True)
    a = (tmp_path / "share0.txt").read_text().split()
    b = (tmp_path / "share1.txt").read_text().split()
    assert a == b and int(a[1]) == len(text) and int(a[2]) == zlib.crc32(text)


class _NumpyNodes:
    """What ss_nodes_* gives dist.exchange_touched, on numpy arrays: a dense node-major buffer, touched flags, packing of
    the touched nodes' segments in node order.  The semantics of ss_nodes.hip (pack_offsets_kernel / pack_copy_kernel)."""

    def __init__(self, offsets, val, state=None):
        self._state = {} if state is None else state       # what the product keeps on the NodeSet: the learnt buffer size
        self.offsets = np.asarray(offsets, np.int64)
        self.n_nodes = self.offsets.size - 1
        self.val = val
        self.touched = np.zeros(self.n_nodes, np.int32)
        for j in range(self.n_nodes):
            self.touched[j] = int(val[self.offsets[j]:self.offsets[j + 1]].any())

    def flags_get(self, t, stream):
        t[:self.n_nodes] = torch.from_numpy(self.touched)

    def flags_set(self, t, stream):
        self.touched = t[:self.n_nodes].numpy().copy()

    def _segments(self):
        return [(int(self.offsets[j]), int(self.offsets[j + 1])) for j in range(self.n_nodes) if self.touched[j]]

    def pack(self, t, stream):
        seg = self._segments()
        n = sum(b - a for a, b in seg)
        if t is not None:
            assert t.numel() >= n
            o = 0
            for a, b in seg:
                t[o:o + b - a] = torch.from_numpy(self.val[a:b].view(np.int32))
                o += b - a
        return n

    def unpack(self, t, stream):
        o = 0
        for a, b in self._segments():
            self.val[a:b] = t[o:o + b - a].numpy().view(np.uint32)
            o += b - a

    # capped forms (ss_nodes_pack_capped_dev / ss_nodes_unpack_capped_dev): the first `cap` packed counts travel
    def pack_capped(self, t, cap, total_t, stream):
        o = 0
        for a, b in self._segments():
            m = max(0, min(b - a, cap - o))
            t[o:o + m] = torch.from_numpy(self.val[a:a + m].view(np.int32))
            o += b - a
        total_t[0] = o

    def unpack_capped(self, t, cap, stream):
        o = 0
        for a, b in self._segments():
            m = max(0, min(b - a, cap - o))
            self.val[a:a + m] = t[o:o + m].numpy().view(np.uint32)
            o += b - a

    @property
    def n_positions(self):
        return int(self.offsets[-1])

    @property
    def state(self):
        return self._state


def _exchange_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", init_method="file://" + os.path.join(out_dir, "store_%d" % port), rank=rank, world_size=world)
    from strainscan_amd import dist as sdist
    rs = np.random.RandomState(7)                            # the same tree on every rank ...
    lens = rs.randint(0, 400, size=40)
    lens[[3, 17]] = 0                                        # ... with empty nodes
    offsets = np.concatenate([[0], np.cumsum(lens)])
    rr = np.random.RandomState(100 + rank)                   # ... different hits per rank
    val = np.zeros(int(offsets[-1]), np.uint32)
    hot = rr.choice(40, size=6, replace=False)               # this rank has hits in six nodes
    for j in hot:
        a, b = offsets[j], offsets[j + 1]
        if b > a:
            idx = rr.randint(a, b, size=max(1, (b - a) // 3))
            val[idx] += rr.randint(1, 50, size=idx.size).astype(np.uint32)
    if offsets[1] > 0:
        val[0] += np.uint32(0x90000000)                      # unsigned wrap across ranks
    np.save(os.path.join(out_dir, "val%d.npy" % rank), val)
    ex = _NumpyNodes(offsets, val.copy())
    sdist.PACK_MIN = 64                                      # the first buffer is too small on purpose
    pend = sdist.exchange_touched(None, device="cpu", ex=ex)
    rounds = 1
    assert pend.cap == 64 and not pend.complete()            # ... every rank sees that from the same total ...
    while not pend.complete():                               # ... and repeats as NodeSet.harvest does: harvest again, exchange
        ex = _NumpyNodes(offsets, val.copy(), ex.state)
        pend = sdist.exchange_touched(None, device="cpu", ex=ex)
        rounds += 1
    assert rounds == 2 and pend.cap >= pend.total()
    n = pend.total()
    np.save(os.path.join(out_dir, "sum%d.npy" % rank), ex.val)
    np.save(os.path.join(out_dir, "flags%d.npy" % rank), ex.touched)
    with open(os.path.join(out_dir, "n%d.txt" % rank), "w") as f:
        f.write(str(n))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_touched_sums_the_union(tmp_path, world):
    """dist.exchange_touched (the collective of a sharded tree scan): flags MAX-all-reduced, the union's segments packed,
    SUM-all-reduced (uint32 wrap included), unpacked -- every rank ends with the global counts in every node any rank
    touched, and only those nodes travel."""
    port = 33500 + (os.getpid() % 2000) + world
    mp.spawn(_exchange_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    vals = [np.load(tmp_path / ("val%d.npy" % r)) for r in range(world)]
    want = np.zeros_like(vals[0])
    for v in vals:
        want = want + v                                      # uint32 arithmetic wraps
    rs = np.random.RandomState(7)
    lens = rs.randint(0, 400, size=40)
    lens[[3, 17]] = 0
    offsets = np.concatenate([[0], np.cumsum(lens)])
    union = np.array([int(any(v[offsets[j]:offsets[j + 1]].any() for v in vals)) for j in range(40)])
    n_want = int(sum(lens[j] for j in range(40) if union[j]))
    assert 0 < n_want < offsets[-1]
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("sum%d.npy" % r)), want)
        assert np.array_equal(np.load(tmp_path / ("flags%d.npy" % r)), union)
        assert int((tmp_path / ("n%d.txt" % r)).read_text()) == n_want


def _chain_worker(rank, world, port, out_dir):
    import ctypes as C
    import json
    import time
    sys.path.insert(0, REPO)
    os.environ["SS_GZ_CHAIN_TIMEOUT"] = "1.5"
    dist.init_process_group("gloo", init_method="file://" + os.path.join(out_dir, "store_%d" % port), rank=rank, world_size=world)
    from strainscan_amd import dist as sdist
    chain = sdist._gz_chain(rank, world)
    n = 4096
    buf = (C.c_uint8 * n)()
    res = {}
    # a message travels: the owner of slice 0 (rank 0) sends, the owner of slice 1 (rank 1) receives
    if rank == 0:
        for i in range(n):
            buf[i] = (i * 7 + 3) & 0xFF
        res["send"] = chain(C.addressof(buf), n, 0, 1, None)
    elif rank == 1:
        res["recv"] = chain(C.addressof(buf), n, 1, 0, None)
        res["payload_ok"] = all(buf[i] == ((i * 7 + 3) & 0xFF) for i in range(n))
    dist.barrier()
    # the peer is gone: rank 1 waits for the message of slice 1 + world, which rank 0 never sends -- the call must come back
    # with a failure after the deadline, not sit in recv
    if rank == 1:
        t0 = time.monotonic()
        res["late"] = chain(C.addressof(buf), n, 1 + world, 0, None)
        res["waited"] = time.monotonic() - t0
        res["failures"] = list(sdist.CHAIN_FAILURES)
    else:
        time.sleep(4)           # alive, but silent (a peer that has gone away altogether fails the receive at once)
    with open(os.path.join(out_dir, "chain%d.json" % rank), "w") as f:
        json.dump(res, f)
    os._exit(0)             # (a receive is still posted on rank 1: no orderly shutdown of the group)


def test_chain_calls_are_bounded(tmp_path):
    """dist._gz_chain (range mode's point-to-point chain): a message arrives intact; a receive whose sender never sends
    ends after SS_GZ_CHAIN_TIMEOUT with a failure -- the library then declines and passes status -1 on -- instead of sitting
    in recv until the backend's own timeout."""
    import json
    port = 29500 + (os.getpid() % 2000) + 7
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_chain_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    r0 = json.load(open(tmp_path / "chain0.json"))
    r1 = json.load(open(tmp_path / "chain1.json"))
    assert r0["send"] == 0 and r1["recv"] == 0 and r1["payload_ok"] is True
    assert r1["late"] == 1 and 1.4 <= r1["waited"] < 10 and r1["failures"] == [[3, 0]]


def _rank0_first_worker(rank, world, port, out_dir, fail_rank):
    import json
    sys.path.insert(0, REPO)
    os.environ["SS_IMAGE_CACHE"] = os.path.join(out_dir, "cache")
    dist.init_process_group("gloo", init_method="file://" + os.path.join(out_dir, "store_%d" % port), rank=rank, world_size=world)
    from strainscan_amd import db as ssdb
    calls = []

    def fn():
        calls.append(rank)
        if rank == fail_rank:
            raise OSError("cannot read the database on rank %d" % rank)
        return "image"

    res = dict(rank=rank)
    try:
        res["out"] = ssdb.rank0_first(fn)
    except BaseException as e:          # noqa: B902
        res["error"] = type(e).__name__ + ": " + str(e)
    res["calls"] = len(calls)
    # whatever happened, the ranks are still in step: a collective that follows completes on all of them
    t = torch.ones(1)
    dist.all_reduce(t)
    res["after"] = float(t.item())
    with open(os.path.join(out_dir, "r0f_%d_%d.json" % (fail_rank, rank)), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [0, 2, -1])
def test_rank0_first_failure_is_raised_on_every_rank(tmp_path, fail_rank):
    """db.rank0_first (rank 0 builds and exports the image, the others import it): a failure on ANY rank -- rank 0 while it
    builds, another one while it imports -- is raised on all of them, and nobody is left in a collective alone."""
    import json
    port = 29500 + (os.getpid() % 2000) + 11 + fail_rank
    mp.spawn(_rank0_first_worker, args=(3, port, str(tmp_path), fail_rank), nprocs=3, join=True)
    res = [json.load(open(tmp_path / ("r0f_%d_%d.json" % (fail_rank, r)))) for r in range(3)]
    assert all(r["after"] == 3.0 for r in res)
    if fail_rank < 0:
        assert all(r.get("out") == "image" and r["calls"] == 1 for r in res)
    else:
        assert all("error" in r for r in res), res
        assert "cannot read the database" in res[fail_rank]["error"]
        # rank 0 failed: the others never started; another rank failed: everybody had run its turn
        assert [r["calls"] for r in res] == ([1, 0, 0] if fail_rank == 0 else [1, 1, 1])

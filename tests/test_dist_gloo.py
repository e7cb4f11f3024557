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
    dist.init_process_group("gloo", rank=rank, world_size=world)
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

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


def _share_worker(rank, world, port, gz_path, plain_path, out_dir):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_WORLD_SIZE"] = str(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    dist.destroy_process_group()


@pytest.mark.skipif(not os.path.isdir("/dev/shm"), reason="no tmpfs")
def test_two_ranks_share_one_inflate(tmp_path):
    """dist.share_inflated: rank 0 inflates a .gz once into /dev/shm, both ranks get the same plain file with the right
    bytes, plain inputs pass through, cleanup removes the file."""
    import gzip
    import zlib
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

"""Multi-GPU identification: reads shard, the k-mer table is replicated, hit counts are summed.

One process per GPU (torchrun / torch.distributed, backend "nccl" = RCCL over xGMI).  k-mers never
span reads, so the scan partitions by read block with no data-path exchange; the
collectives sum hit counts.  Tree scan: the counts of the nodes WITH hits only (exchange_touched:
touched flags MAX-all-reduced, those nodes' segments packed and SUM-all-reduced -- ~1-3 MB where the
uint32[n_rows] vector of an E. coli table is 100 MB); the full row vector is all-reduced only when the
walk asks for single rows (adjust_profile's Poisson branch, identify.py:203-218) and for layer-2 scans,
whose y vector is every row of a (much smaller) cluster table.  Integer sums commute, so the results are
bit-identical to a single-GPU scan whatever the block-to-rank assignment.  Every rank runs the (tiny,
sequential) tree walk redundantly; rank 0 writes the reports.
"""
import os

import numpy as np

from . import _lib

CHUNK = 1 << 26      # elements per all-reduce call (256 MiB of int32): bounded staging, few large collectives


def is_distributed():
    """A process group exists and has more than one rank.  A group can only have been created through
    torch.distributed, so if that module is not loaded the answer is no -- WITHOUT importing torch (2.7 s in a
    process whose whole identification takes 0.13 s)."""
    import sys
    dist = sys.modules.get("torch.distributed")
    if dist is None:
        return False
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank_world():
    if not is_distributed():
        return 0, 1
    import torch.distributed as dist
    return dist.get_rank(), dist.get_world_size()


def rank_blocks(blocks, rank, world):
    """Round-robin assignment of flat base blocks (any iterable) to ranks."""
    for i, b in enumerate(blocks):
        if i % world == rank:
            yield b


def allreduce_counts(t, group=None):
    """In-place sum over ranks of a hit-count vector.  `t` is a torch tensor (int32 on the GPU for
    RCCL, or on the CPU for gloo) holding uint32 bit patterns: two's-complement addition IS uint32
    addition modulo 2^32, so int32 SUM is exact for the unsigned counts."""
    import torch
    import torch.distributed as dist
    assert t.dtype == torch.int32 and t.is_contiguous()
    if dist.get_world_size(group) == 1:
        return t
    flat = t.view(-1)
    for lo in range(0, flat.numel(), CHUNK):
        dist.all_reduce(flat[lo:lo + CHUNK], op=dist.ReduceOp.SUM, group=group)
    return t


class _NodeExchange:
    """What exchange_touched needs from a node set: the touched flags out and in, the touched nodes' segments packed
    and unpacked (device pointers; strainscan_amd._lib.NodeSet over ss_nodes_*; tests substitute a numpy double)."""

    def __init__(self, nodes):
        self.nodes = nodes
        self.n_nodes = nodes.n_nodes

    def flags_get(self, t, stream):
        _lib.check(_lib.lib().ss_nodes_touched_get_dev(self.nodes._h, t.data_ptr(), stream), "ss_nodes_touched_get_dev")

    def flags_set(self, t, stream):
        _lib.check(_lib.lib().ss_nodes_touched_set_dev(self.nodes._h, t.data_ptr(), stream), "ss_nodes_touched_set_dev")

    def pack(self, t, stream):
        import ctypes as C
        n = C.c_uint64()
        _lib.check(_lib.lib().ss_nodes_pack_dev(self.nodes._h, t.data_ptr() if t is not None else None,
                                                t.numel() if t is not None else 0, C.byref(n), stream), "ss_nodes_pack_dev")
        return int(n.value)

    def unpack(self, t, stream):
        _lib.check(_lib.lib().ss_nodes_unpack_dev(self.nodes._h, t.data_ptr(), stream), "ss_nodes_unpack_dev")

    # capped forms: no host round trip; the packed size stays on the device until the statistics are read
    def pack_capped(self, t, cap, total_t, stream):
        _lib.check(_lib.lib().ss_nodes_pack_capped_dev(self.nodes._h, t.data_ptr(), cap, total_t.data_ptr(), stream),
                   "ss_nodes_pack_capped_dev")

    def unpack_capped(self, t, cap, stream):
        _lib.check(_lib.lib().ss_nodes_unpack_capped_dev(self.nodes._h, t.data_ptr(), cap, stream), "ss_nodes_unpack_capped_dev")

    @property
    def n_positions(self):
        return int(self.nodes.n_rows_total)

    @property
    def state(self):
        """Where the buffer size learnt from earlier scans lives (the node set itself)."""
        return self.nodes.__dict__


_PACK_BUF = {}
PACK_MIN = 1 << 18        # elements: 1 MB -- below this a smaller all-reduce is no faster


def exchange_device():
    """Tensors of the exchange live on the GPU (RCCL; gloo takes GPU tensors too and stages them itself)."""
    return "cuda"


class PackedExchange:
    """What exchange_touched returns: the buffer size it used and, on the device, the size the packed counts needed.
    complete() reads that number (call it once the stream has been synchronised -- the caller does that anyway to
    read the node statistics) and sizes the next exchange of this node set: about twice the last total."""

    def __init__(self, cap, total_t, state, n_positions):
        self.cap, self._total_t, self._state, self._n_positions = cap, total_t, state, n_positions
        self._total = None

    def total(self):
        if self._total is None:
            self._total = int(self._total_t.item()) if self._total_t is not None else 0
            want = min(max(PACK_MIN, 2 * self._total), max(1, self._n_positions))
            if self._total > self.cap or 4 * want <= self.cap:      # grow at once, shrink only when far too large
                self._state["_pack_cap"] = want
        return self._total

    def complete(self):
        return self.total() <= self.cap

    def __int__(self):
        return self.total()


def exchange_touched(nodes, group=None, device="cuda", stream=None, ex=None):
    """The collective of a sharded tree scan, between ss_nodes_harvest_dev and ss_nodes_reduce_touched_dev: every rank
    has harvested ITS reads' counts into the node-major buffer.  (1) MAX-all-reduce of the touched flags (4 bytes per
    node) -> every rank knows the union of the nodes with hits; (2) the segments of those nodes, packed in node order
    (same layout on every rank), are SUM-all-reduced and unpacked.  Bytes per rank per scan: 4 * n_nodes + 4 * (rows
    of the touched nodes, times the slack of the buffer: <= 2) -- a few MB for a three-strain sample against an E. coli
    tree of 25 M rows, where the full row vector is 100 MB.  Integer sums: bit-identical to a single-GPU scan.

    Nothing here waits for the host: the buffer has a size decided BEFORE the scan (twice what the previous exchange
    of this node set needed, 1 MB at least, never more than all list positions), the device packs what fits and leaves
    the real total in a device word.  -> PackedExchange; its complete() tells the caller -- when it reads the node
    statistics -- whether everything travelled (every rank computes the same total from the same flags, so all ranks
    agree); if not, the caller harvests again and calls this again, now with the larger buffer (NodeSet.harvest)."""
    import torch
    import torch.distributed as dist
    ex = ex or _NodeExchange(nodes)
    if stream is None and device != "cpu":
        stream = torch.cuda.current_stream().cuda_stream
    flags = torch.empty(max(1, ex.n_nodes), dtype=torch.int32, device=device)
    ex.flags_get(flags, stream)
    dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
    ex.flags_set(flags, stream)
    n_pos = ex.n_positions
    if n_pos == 0:
        return PackedExchange(0, None, ex.state, 0)
    cap = min(int(ex.state.get("_pack_cap", PACK_MIN)), n_pos)
    buf = _PACK_BUF.get(device)
    if buf is None or buf.numel() < cap:
        buf = torch.zeros(max(cap, 1 << 20), dtype=torch.int32, device=device)
        _PACK_BUF.clear()
        _PACK_BUF[device] = buf
    total_t = torch.zeros(1, dtype=torch.int64, device=device)
    ex.pack_capped(buf, cap, total_t, stream)
    allreduce_counts(buf[:cap], group)
    ex.unpack_capped(buf, cap, stream)
    return PackedExchange(cap, total_t, ex.state, n_pos)


def allreduce_table(kdb):
    """Sum the per-rank hit counts of `kdb` over all ranks and load the global vector back."""
    import torch
    t = torch.empty(kdb.n_rows, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib().ss_device_sync(), "ss_device_sync")
    kdb.counts_rows_dev(t.data_ptr(), stream)
    allreduce_counts(t)
    kdb.load_counts_rows_dev(t.data_ptr(), stream)
    torch.cuda.synchronize()


def scan_files_sharded(kdb, paths, allreduce=True):
    """Scan this rank's share of the reads into `kdb` (ss_scan_files_shard: a rank parses, copies and scans only its share
    of every file; .gz inputs: load_agreed); with `allreduce` the row counts are then summed over the ranks and loaded
    back (layer-2 scans need every row; the tree scan exchanges the touched nodes instead, exchange_touched).
    Returns this rank's (n_records, n_bases)."""
    rank, world = rank_world()

    def scan(use):
        kdb.reset()                       # (a second attempt starts from zero: the first may have counted some files)
        return kdb.scan_files(use, rank, world)

    nrec, nb = load_agreed([p for p in paths if p], scan)
    if world > 1 and allreduce:
        allreduce_table(kdb)
    return nrec, nb


def _gz_chain(rank, world):
    """The callback of ss_gz_set_range: direction 1 = send the message to the owner of the next slice, 0 = receive the one
    for `slice` from the owner of the slice before (slice s belongs to rank s mod world).  Called on the loader's worker
    thread while this thread waits inside the library.  NCCL moves device tensors, gloo host tensors."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    on_dev = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_dev else torch.device("cpu")

    # Every wait is BOUNDED (SS_GZ_CHAIN_TIMEOUT seconds, default 60): a peer that never sends -- it crashed, or left without
    # serving the chain -- must not leave this rank in a blocking recv on the loader thread until the backend's own timeout
    # (10 minutes for NCCL).  After the deadline the call fails, the library declines, passes status -1 on to the ranks behind
    # this one, and load_agreed moves all ranks to the whole-file path; range mode is not tried again in this process group
    # (the abandoned receive is still posted).
    deadline = float(os.environ.get("SS_GZ_CHAIN_TIMEOUT", "60"))

    def wait(work):
        import threading
        import time
        if on_dev:                           # NCCL: the work's event is polled (wait() would only block the stream)
            t_end = time.monotonic() + deadline
            while not work.is_completed():
                if time.monotonic() > t_end:
                    return False
                time.sleep(0.0005)
            work.wait()                      # (completed: raises here if the operation failed)
            return True
        # gloo: a point-to-point work only completes inside wait() (is_completed() never turns true on its own), and a wait
        # WITH a timeout tears the connection to the peer down when it runs out -- the collectives that follow would fail.
        # So the blocking wait runs on a helper thread and THIS thread waits for it, bounded; after the deadline the helper
        # is left behind with its posted receive (a daemon thread: it ends with the process)
        done, failed = threading.Event(), []

        def block():
            try:
                work.wait()
            except BaseException as e:      # noqa: B902
                failed.append(e)
            done.set()

        threading.Thread(target=block, name="ss-gz-chain-wait", daemon=True).start()
        return done.wait(deadline) and not failed

    def chain(msg, nbytes, slice_, direction, _user):
        try:
            if on_dev:
                torch.cuda.set_device(dev)          # (the current device is a property of the thread: this is the loader's)
            host = torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(msg)))
            if direction == 1:
                out = host.to(dev) if on_dev else host.clone()
                ok = wait(dist.isend(out, (slice_ + 1) % world))
            else:
                box = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                ok = wait(dist.irecv(box, (slice_ - 1) % world))
                if ok:
                    host.copy_(box.cpu() if on_dev else box)
            if not ok:
                CHAIN_FAILURES.append((slice_, direction))
            return 0 if ok else 1
        except BaseException:               # noqa: B902 -- the library breaks the chain off and declines
            if os.environ.get("SS_INGEST_TRACE"):
                import traceback
                traceback.print_exc()
            CHAIN_FAILURES.append((slice_, direction))
            return 1

    return _lib.GZ_CHAIN_FN(chain)


_P2P_OK = {}
CHAIN_FAILURES = []       # (slice, direction) of every chain call that failed or ran out of time in this process


def _p2p_works():
    """Point-to-point traffic between neighbouring ranks, tried ONCE per process group before the first shared inflation
    (a ring: every rank sends a word to the next and receives one from the one before; the outcomes are MIN-all-reduced):
    range mode's chain hangs where send / recv do not work, so it is only used where they have been seen to."""
    import torch
    import torch.distributed as dist
    key = (dist.get_backend(), dist.get_world_size(), dist.get_rank())
    if key not in _P2P_OK:
        rank, world = dist.get_rank(), dist.get_world_size()
        on_dev = dist.get_backend() == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if on_dev else torch.device("cpu")
        agreed = False
        for attempt in range(2):            # (the first point-to-point traffic of a group sets its connections up: one more try,
            ok = 1                          #  taken by all ranks together -- they all see the same MIN)
            try:
                out = torch.full((1,), rank + 1000 * attempt, dtype=torch.int32, device=dev)
                box = torch.empty(1, dtype=torch.int32, device=dev)
                reqs = [dist.isend(out, (rank + 1) % world), dist.irecv(box, (rank - 1) % world)]
                for r in reqs:
                    r.wait()
                ok = int(int(box.item()) == (rank - 1) % world + 1000 * attempt)
            except BaseException:           # noqa: B902
                ok = 0
            t = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            agreed = bool(int(t.item()))
            if agreed:
                break
        _P2P_OK[key] = agreed
    return _P2P_OK[key]


def _is_gz(p):
    try:
        with open(p, "rb") as f:
            return f.read(2) == b"\x1f\x8b"
    except OSError:
        return False


def load_agreed(paths, load, discard=None):
    """load(paths) under torch.distributed when some inputs are .gz.  The device path (ss_ginflate.hip + ss_fastq_dev.hip)
    gives a rank its blocks of 4096 records, the host inflaters give it parse chunks of the text, and the device path may
    decline a file for reasons of one rank's own (free memory, a HIP error): if the ranks took different paths for a file,
    reads would be counted twice or not at all, silently.  So: every rank loads under the STRICT policy (the device or
    SS_EAGAIN, nothing loaded), the outcomes are MIN-all-reduced, and unless every rank succeeded all of them load again
    with the host inflaters -- rank 0 inflating each .gz once into /dev/shm for all (share_inflated).  An exception on one
    rank is raised on all of them (no rank is left waiting at a collective).  `discard(obj)` releases what a successful
    first attempt of THIS rank produced when another rank was declined."""
    paths = list(paths)
    if not is_distributed() or not any(_is_gz(p) for p in paths if p):
        return load(paths)
    import torch
    import torch.distributed as dist
    dev = "cuda" if torch.cuda.is_available() else "cpu"

    def agree(status):
        t = torch.tensor([status], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    err = None
    rank, world = rank_world()
    if os.environ.get("SS_GZ_GPU", "1") != "0" and os.environ.get("SS_GZ_RANGE", "1") != "0" and _p2p_works():
        # first: the ranks SHARE every file's inflation (ss_gz_set_range: slices of the deflate data round-robin, a chain of
        # small point-to-point messages hands each slice what lies in front of it); strict as well -- a rank that cannot take
        # part serves the chain, reports SS_EAGAIN, and everybody moves on to the next way
        obj, status = None, 1
        chain = _gz_chain(rank, world)
        try:
            _lib.check(_lib.lib().ss_gz_set_range(rank, world, 0, chain, None), "ss_gz_set_range")
            with _lib.gz_policy(1):
                obj = load(paths)
        except _lib.SSError as e:
            status, err = (0, None) if e.code == _lib.SS_EAGAIN else (-1, e)
        except BaseException as e:          # noqa: B902 -- re-raised below, after the other ranks have been told
            status, err = -1, e
        finally:
            _lib.lib().ss_gz_set_range(0, 1, 0, _lib.NO_CHAIN, None)
        agreed = agree(status)
        # a chain call that failed or timed out on ANY rank: no more shared inflations in this process group (a receive may
        # still be posted on some rank; the next file's messages must not meet it)
        if agree(0 if CHAIN_FAILURES else 1) == 0:
            _P2P_OK[(dist.get_backend(), dist.get_world_size(), dist.get_rank())] = False
        if agreed == 1:
            return obj
        if obj is not None and discard is not None:
            discard(obj)
        if agreed < 0:
            raise err if err is not None else RuntimeError("another rank failed while loading the reads")
        err = None
    if os.environ.get("SS_GZ_GPU", "1") != "0":
        obj, status = None, 1
        try:
            with _lib.gz_policy(1):
                obj = load(paths)
        except _lib.SSError as e:
            status, err = (0, None) if e.code == _lib.SS_EAGAIN else (-1, e)
        except BaseException as e:          # noqa: B902 -- re-raised below, after the other ranks have been told
            status, err = -1, e
        agreed = agree(status)
        if agreed == 1:
            return obj
        if obj is not None and discard is not None:
            discard(obj)
        if agreed < 0:
            raise err if err is not None else RuntimeError("another rank failed while loading the reads")
    use, cleanup = share_inflated(paths)
    obj, status = None, 1
    try:
        with _lib.gz_policy(2):
            obj = load(use)
    except BaseException as e:              # noqa: B902
        status, err = -1, e
    try:
        agreed = agree(status)
    finally:
        cleanup()
    if agreed < 0:
        raise err if err is not None else RuntimeError("another rank failed while loading the reads")
    return obj


def share_inflated(paths, shm_dir="/dev/shm"):
    """.gz inputs that go through the HOST inflaters under torch.distributed on ONE node (SS_GZ_GPU=0, or the device path
    declined one of them on some rank: load_agreed): rank 0 inflates each of them once into a plain file on tmpfs
    (ss_gz_inflate_to_file: the threaded inflater writes through a shared mapping), the names are broadcast, every
    rank then parses only its share of the plain text -- instead of every rank inflating the whole file before it
    can pick its share (a gzip member has no entry points).  -> (paths to read, cleanup): call cleanup() when the
    reads are loaded (a barrier, then rank 0 removes the files).  Not distributed, several nodes, SS_GZ_SHARE=0,
    no room on tmpfs, or an input that cannot be inflated whole: the original paths come back."""
    import ctypes as C
    paths = list(paths)
    if not is_distributed() or os.environ.get("SS_GZ_SHARE", "1") == "0":
        return paths, (lambda: None)
    import torch.distributed as dist
    rank, world = rank_world()
    if int(os.environ.get("LOCAL_WORLD_SIZE", world)) != world or not os.path.isdir(shm_dir):
        return paths, (lambda: None)
    names = [None] * len(paths)
    job_dir = None
    if rank == 0:
        import tempfile
        try:
            for i, p in enumerate(paths):
                if not p or not _is_gz(p):
                    continue
                fs = os.statvfs(shm_dir)
                if fs.f_bavail * fs.f_frsize < 8 * os.path.getsize(p):
                    continue
                if job_dir is None:
                    job_dir = tempfile.mkdtemp(prefix="ss_inflate_", dir=shm_dir)       # 0700, unpredictable name
                out = os.path.join(job_dir, "%d_%s.txt" % (i, os.path.basename(p)))
                n = C.c_uint64()
                if _lib.lib().ss_gz_inflate_to_file(os.fsencode(p), os.fsencode(out), 0, C.byref(n)) == _lib.SS_OK:
                    os.chmod(out, 0o644)
                    names[i] = out
            if job_dir is not None:
                os.chmod(job_dir, 0o755)          # the other ranks of the job read the files
        except Exception:                         # whatever happens on rank 0, the broadcast below must take place
            names = [None] * len(paths)
    box = [names, job_dir]
    dist.broadcast_object_list(box, src=0)
    names, job_dir = box

    def cleanup():
        dist.barrier()
        if rank == 0 and job_dir:
            import shutil
            shutil.rmtree(job_dir, ignore_errors=True)

    return [names[i] or p for i, p in enumerate(paths)], cleanup


def init_from_env():
    """torchrun environment -> process group on this rank's GPU (no-op for a single process)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local)
    _lib.check(_lib.lib().ss_set_device(local), "ss_set_device")
    if not dist.is_initialized():
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return dist.get_rank(), dist.get_world_size()

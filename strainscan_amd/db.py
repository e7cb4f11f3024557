"""On-disk StrainScan databases -> device images.

Tree_database/ (written by the reference's library/Build_tree.py:494-526,648-698):
    kmer.fa                 `>1\\n<31-mer>\\n` per row
    kmers/<id>              space separated 0-based rows of kmer.fa, one line
    node_length.txt         `id \\t len`
    reconstructed_nodes.txt one id per line
    overlapping_info/<leaf>, <leaf>_supple
Kmer_Sets_L2/Kmer_Sets/C<id>/ is handled by strainscan_amd/Vote_Strain_L2_Lasso_new_sp.py (cluster image cache there).

A TreeImage owns the device k-mer table of kmer.fa and the node row lists; scans accumulate in
it.  Images are cached per (directory, key mode) for the life of the process so that the cutoff
ladder of StrainScan.py:196-216 (which calls identify_cluster up to twice) parses and uploads a
database once.
"""
import os
import threading

import numpy as np

from . import _lib

L1_K = 31  # the reference hard-codes `-m 31` for the tree scan (identify.py:82,86)


class CountsView:
    """`match_results` of library/identify.py:96-101 without materialising a dict: a read-only
    mapping row -> count over the per-row arrays (only valid rows are keys)."""

    def __init__(self, counts, valid):
        self.counts = counts
        self.valid = valid

    def __getitem__(self, row):
        if not self.valid[row]:
            raise KeyError(row)
        return int(self.counts[row])

    def __contains__(self, row):
        return 0 <= row < self.valid.size and bool(self.valid[row])

    def __len__(self):
        return int(self.valid.sum())

    def keys(self):
        return np.nonzero(self.valid)[0]

    def __iter__(self):
        return iter(self.keys().tolist())

    def items(self):
        k = self.keys()
        return zip(k.tolist(), self.counts[k].tolist())

    def get(self, row, default=None):
        return int(self.counts[row]) if row in self else default


_READS = {}          # one resident read set at a time: key -> _lib.ReadSet
_READS_LOCK = threading.Lock()
RESIDENT_LIMIT_BYTES = int(float(os.environ.get("SS_READS_RESIDENT_GB", "160")) * 1e9)


def _reads_key(paths, rank, world):
    return tuple((os.path.abspath(p), os.path.getmtime(p), os.path.getsize(p)) for p in paths if p) + (rank, world)


def resident_reads(paths):
    """The sample's reads as flat base blocks in HBM (parsed and copied over PCIe once), or None
    when they would not fit the budget (then every scan streams the files again).  Under
    torch.distributed each rank keeps only its shard of the blocks."""
    from . import dist
    rank, world = dist.rank_world()
    total = sum(os.path.getsize(p) for p in paths if p)
    gz = any(str(p).endswith(".gz") for p in paths if p)
    if (total * (4 if gz else 1)) / world > RESIDENT_LIMIT_BYTES or RESIDENT_LIMIT_BYTES <= 0:
        return None
    key = _reads_key(paths, rank, world)
    with _READS_LOCK:                     # layer 2 asks from several threads; the set is parsed once
        rs = _READS.get(key)
        if rs is None:
            for old in _READS.values():
                old.close()
            _READS.clear()
            # (.gz under torch.distributed: all ranks take the same inflate path for a file, dist.load_agreed)
            rs = dist.load_agreed([p for p in paths if p], lambda use: _lib.ReadSet(use, rank, world), discard=lambda r: r.close())
            _READS[key] = rs
    return rs


def prefetch_reads(paths):
    """Start loading the sample's reads (parse || PCIe, binning) on a worker thread WHILE the caller loads the database
    image -- both are native calls that leave the interpreter lock alone, and neither needs the other: the cached
    identify_cluster() call on 16 M reads is image 0.10 s + reads 0.07 s one after the other.  One process only (under
    torch.distributed both sides run collectives, whose order must be the same on every rank); a failure here is raised
    again by the load that follows."""
    from . import dist
    if dist.is_distributed():
        return None
    ps = [p for p in paths if p]
    if not ps or not all(os.path.exists(p) for p in ps):
        return None

    def work():
        try:
            resident_reads(paths)
        except BaseException:               # noqa: B902 -- the caller's own resident_reads() raises it again
            pass

    th = threading.Thread(target=work, name="ss-prefetch-reads")
    th.start()
    return th


def scan_into(kdb, paths, allreduce=True):
    """Count kdb's k-mers in the reads of `paths`: resident blocks when possible, streaming
    otherwise.  Under torch.distributed every rank scans its share of the reads; with `allreduce` the row counts are
    then summed over the ranks (RCCL) and loaded back, without it the table keeps this rank's counts (the tree scan:
    TreeImage exchanges the touched nodes' counts instead, dist.exchange_touched)."""
    from . import dist
    kdb.reset()
    rs = resident_reads(paths)
    if rs is not None:
        rs.scan_into(kdb)
        if allreduce and dist.is_distributed():
            dist.allreduce_table(kdb)
    elif dist.is_distributed():
        dist.scan_files_sharded(kdb, paths, allreduce=allreduce)
    else:
        kdb.scan_files([p for p in paths if p])


INDEX_EVENTS = {"built": 0, "imported": 0}       # device indexes built from keys / imported from an image, this process


def rank0_first(fn):
    """Several ranks and an image cache: rank 0 runs fn() first -- it parses / builds and EXPORTS whatever image is
    missing -- the others wait at a barrier and then run the same fn(), which now finds the image and imports it:
    one host-side index build per node instead of one per rank (an E. coli tree: ~1 s of all the box's CPUs each).
    The barrier is unconditional (every rank takes it once per call), so the ranks cannot disagree about it; a rank
    that finds no image after all (cache not writable) simply builds its own."""
    from . import dist
    if not dist.is_distributed() or not _cache_dir():
        return fn()
    import torch
    import torch.distributed as td
    rank, _ = dist.rank_world()
    dev = "cuda" if torch.cuda.is_available() and td.get_backend() == "nccl" else "cpu"

    def agree(ok):
        """MIN over the ranks of a success flag: doubles as the barrier, and a rank that failed is seen by all (the others
        would otherwise go on into the scan's collectives with one rank missing, and hang there)."""
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        td.all_reduce(t, op=td.ReduceOp.MIN)
        return bool(int(t.item()))

    out, err = None, None
    if rank == 0:
        try:
            out = fn()
            wait_cache_writes()
        except Exception as e:              # raised below, after the other ranks have been told; KeyboardInterrupt /
            err = e                         # SystemExit leave at once (a rank on its way out must not enter a collective)
    if not agree(err is None):              # rank 0's turn
        raise err if err is not None else RuntimeError("rank 0 failed while building the database image")
    if rank != 0:
        try:
            out = fn()
        except Exception as e:
            err = e
    if not agree(err is None):              # everybody else's
        raise err if err is not None else RuntimeError("another rank failed while loading the database image")
    return out


def fasta_index(path, k, upper_keys):
    """Device index of a k-mer FASTA (a cluster's all_kmer.fasta): imported from the image cache when there is
    one for this file (path, size, mtime, k, key convention), else parsed, built and exported.  Only the
    minimizer-paged layout (17 <= k <= 31) has an image; a flat table (k <= 16) is built every time."""
    return rank0_first(lambda: _fasta_index(path, k, upper_keys))


def _fasta_index(path, k, upper_keys):
    cdir = _cache_dir()
    img = None
    st = os.stat(path)                                   # (a cluster without its k-mer set: FileNotFoundError, as the reference's open() raises)
    if cdir and 17 <= int(k) <= 31:
        tag = cache_tag("%s|%d|%d|%d|%d" % (os.path.realpath(path), st.st_size, st.st_mtime_ns, int(k),
                                                 int(upper_keys)))
        img = os.path.join(cdir, "index_%s.bin" % tag)
        if os.path.exists(img):
            try:
                kdb = _lib.KmerDB.from_image(img)
                INDEX_EVENTS["imported"] += 1
                return kdb
            except RuntimeError:
                pass
    kdb = _lib.KmerDB.from_fasta(path, int(k), upper_keys=upper_keys)
    INDEX_EVENTS["built"] += 1
    if img:
        try:
            _export_image(kdb, cdir, img)
        except (RuntimeError, OSError):
            pass
    return kdb


def _export_image(kdb, cdir, img):
    os.makedirs(cdir, exist_ok=True)
    tmp = _cache_tmp(img)
    try:
        kdb.export(tmp)
        os.replace(tmp, img)
    except BaseException:
        _unlink_quiet(tmp)
        raise


def _export_image_later(kdb, cdir, img):
    """The 0.9 GB image of an E. coli tree takes 0.25 s to read back and write: on a worker thread (like the tree cache),
    while the reads are scanned -- it only reads the index arrays, scans only add to the counters.  The handle must outlive
    the writer: wait_cache_writes() runs before an image is dropped (tree_image, clear_cache, TreeImage.__del__)."""

    def work():
        try:
            _export_image(kdb, cdir, img)
        except (RuntimeError, OSError):
            pass

    with _CACHE_WRITERS_LOCK:
        old = _CACHE_WRITERS.get(img)
    if old is not None:
        old.join()
    th = threading.Thread(target=work, name="ss-export-image")
    with _CACHE_WRITERS_LOCK:
        _CACHE_WRITERS[img] = th
    th.start()


def cache_tag(text):
    """Name of a cache entry from the string that identifies it (path | size | mtime | ...): 128 bits of a multiplicative
    mix over the bytes, two lanes with different odd multipliers (non-linear in the input, unlike a CRC: the tag is the only
    identity of an entry).  Not hashlib: importing it costs a fresh CLI process 0.05-0.1 s (OpenSSL) for what is a file name;
    the strings are ~150 bytes."""
    b = text.encode()
    M = (1 << 64) - 1
    h1, h2 = 0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F
    b += b"\x80" + b"\0" * (-(len(b) + 1) % 8)
    for i in range(0, len(b), 8):
        w = int.from_bytes(b[i:i + 8], "little")
        h1 = ((h1 ^ w) * 0xFF51AFD7ED558CCD) & M
        h1 ^= h1 >> 32
        h2 = ((h2 + w) * 0xD6E8FEB86659FD93) & M
        h2 ^= h2 >> 29
    h1 = ((h1 ^ len(text)) * 0xC4CEB9FE1A85EC53) & M
    h2 = ((h2 ^ h1) * 0x9FB21C651E98DF25) & M
    return "%016x%016x" % (h1 ^ (h1 >> 33), h2 ^ (h2 >> 31))


def _cache_dir():
    d = os.environ.get("SS_IMAGE_CACHE", os.path.join(os.path.expanduser("~"), ".cache", "strainscan_amd"))
    return None if d in ("", "0", "off") else d


class TreeArrays:
    """kmer.fa + kmers/<id> as arrays.  keys u64[n] / flags u8[n]: the encoded rows of kmer.fa (only needed
    to BUILD the device index); ids: node ids ascending; lists[i]: rows of node ids[i] in FILE order
    (adjust_profile indexes them); urows / uoffs: the same lists de-duplicated and sorted, back to back
    (what the device-side NodeSet holds: set(map(int, ...)) at identify.py:118)."""

    def __init__(self, keys, flags, ids, rows, offs, urows, uoffs):
        self.keys, self.flags, self.ids = keys, flags, ids
        self.rows, self.offs, self.urows, self.uoffs = rows, offs, urows, uoffs
        self.lists = [rows[offs[i]:offs[i + 1]] for i in range(len(ids))]


_TREE_MAGIC = b"SSTREE02"


def _pad64(n):
    return (n + 63) & ~63


def _write_tree_cache(path, t):
    """One raw file, arrays at 64-byte aligned offsets (read back through one memory map: no decompression,
    no checksum pass, and the 8-byte keys are only paged in when the device index has to be rebuilt)."""
    same = t.urows is t.rows
    arrays = [np.asarray(t.ids, np.int64), np.asarray(t.offs, np.int64), np.asarray(t.rows, np.uint32),
              np.asarray(t.uoffs, np.int64), np.zeros(0, np.uint32) if same else np.asarray(t.urows, np.uint32),
              np.asarray(t.flags, np.uint8), np.asarray(t.keys, np.uint64)]
    hdr = np.array([t.keys.size, len(t.ids), t.rows.size, 0 if same else t.urows.size, int(same), 0], np.uint64)
    tmp = _cache_tmp(path)
    try:
        with open(tmp, "wb") as f:
            f.write(_TREE_MAGIC)
            f.write(hdr.tobytes())
            pos = 8 + hdr.nbytes
            for a in arrays:
                f.write(b"\0" * (_pad64(pos) - pos))
                pos = _pad64(pos)
                f.write(memoryview(np.ascontiguousarray(a)).cast("B"))
                pos += a.nbytes
        os.replace(tmp, path)
    except BaseException:
        _unlink_quiet(tmp)
        raise


def _cache_tmp(path):
    """A temp file of its own in the cache directory for every writer (thread or process): `path.<pid>.tmp` was shared
    by two writer threads of one process, the second truncating what the first was still writing.  The finished
    file reaches `path` through os.replace, so a reader sees the old image, none, or a complete new one."""
    import tempfile
    fd, tmp = tempfile.mkstemp(dir=os.path.dirname(path), prefix=os.path.basename(path) + ".", suffix=".tmp")
    os.close(fd)
    return tmp


def _unlink_quiet(p):
    try:
        os.unlink(p)
    except OSError:
        pass


_CACHE_WRITERS = {}                  # cache path -> thread writing it
_CACHE_WRITERS_LOCK = threading.Lock()


def _write_tree_cache_later(cdir, path, t):
    """The tree cache is written on a worker thread (0.2 s for an E. coli tree) while the first scan runs.  At most
    one writer per path: load_tree() of the same database joins it before it looks for the file (`-b 1` on a new
    database loads the tree twice, identify_low_depth.py:119 then identify.py:402); the interpreter waits for the
    writers at exit, wait_cache_writes() before that."""

    def work():
        try:
            os.makedirs(cdir, exist_ok=True)
            _write_tree_cache(path, t)
        except OSError:
            pass

    with _CACHE_WRITERS_LOCK:
        old = _CACHE_WRITERS.get(path)
    if old is not None:
        old.join()
    th = threading.Thread(target=work)
    with _CACHE_WRITERS_LOCK:
        _CACHE_WRITERS[path] = th
    th.start()


def _join_cache_writer(path):
    with _CACHE_WRITERS_LOCK:
        th = _CACHE_WRITERS.get(path)
    if th is not None:
        th.join()
        with _CACHE_WRITERS_LOCK:
            if _CACHE_WRITERS.get(path) is th:
                del _CACHE_WRITERS[path]


def wait_cache_writes():
    while True:
        with _CACHE_WRITERS_LOCK:
            paths = list(_CACHE_WRITERS)
        if not paths:
            return
        for p in paths:
            _join_cache_writer(p)


def _read_tree_cache(path):
    size = os.path.getsize(path)
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    if size < 56 or bytes(mm[:8]) != _TREE_MAGIC:
        raise ValueError("not a tree cache")
    n_rows, n_ids, n_total, n_utotal, same, _ = (int(x) for x in np.frombuffer(mm[8:56], np.uint64))
    pos = [56]

    def take(dtype, n):
        o = _pad64(pos[0])
        nb = n * np.dtype(dtype).itemsize
        if o + nb > size:
            raise ValueError("truncated tree cache")
        pos[0] = o + nb
        return mm[o:o + nb].view(dtype)

    ids = take(np.int64, n_ids).tolist()
    offs = np.array(take(np.int64, n_ids + 1))
    rows = take(np.uint32, n_total)
    uoffs = np.array(take(np.int64, n_ids + 1))
    urows = take(np.uint32, n_utotal)
    flags = take(np.uint8, n_rows)
    keys = take(np.uint64, n_rows)
    if pos[0] != size or offs[-1] != n_total or uoffs[-1] != (n_total if same else n_utotal):
        raise ValueError("inconsistent tree cache")
    return TreeArrays(keys, flags, ids, rows, offs, rows if same else urows, uoffs)


def _node_lists_native(kdir, ids, n_rows):
    """The same through ss_node_lists_parse (the host's threads: 40 ms for the 1645 files of an E. coli tree where numpy
    takes 0.7 s); None when a file holds anything but row numbers of kmer.fa -- the loop below then raises what it raises."""
    ida = np.ascontiguousarray(ids, np.int64)
    counts = np.zeros(ida.size, np.uint64)
    bad = _lib.C.c_uint32()
    if _lib.lib().ss_node_lists_parse(os.fsencode(kdir), _lib.ptr(ida), ida.size, int(n_rows), _lib.ptr(counts), None, None,
                                      _lib.C.byref(bad)) != _lib.SS_OK:
        return None
    offs = np.zeros(ida.size + 1, np.int64)
    np.cumsum(counts.astype(np.int64), out=offs[1:])
    rows = np.empty(int(offs[-1]), np.uint32)
    uoffs = offs.astype(np.uint64)
    if _lib.lib().ss_node_lists_parse(os.fsencode(kdir), _lib.ptr(ida), ida.size, int(n_rows), _lib.ptr(counts), _lib.ptr(uoffs),
                                      _lib.ptr(rows), _lib.C.byref(bad)) != _lib.SS_OK:
        return None
    # set(map(int, ...)) of identify.py:118: lists that are already strictly increasing (the builder writes them so) are
    # their own de-duplicated form
    inc = rows[1:] > rows[:-1] if rows.size > 1 else np.ones(0, bool)
    if rows.size > 1:
        inner = offs[1:-1]
        inc[inner[(inner > 0) & (inner < rows.size)] - 1] = True
    if inc.all():
        return rows, offs, rows, offs
    ul = [np.unique(rows[offs[i]:offs[i + 1]]) for i in range(ida.size)]
    uo = np.zeros(ida.size + 1, np.int64)
    for i, r in enumerate(ul):
        uo[i + 1] = uo[i] + r.size
    return rows, offs, (np.concatenate(ul) if ul else np.zeros(0, np.uint32)), uo


def _node_lists(db_dir, ids, n_rows):
    """kmers/<id> (one line of row numbers each, identify.py:116-118) -> rows in file order, their offsets, and the
    de-duplicated sorted form the node statistics use."""
    kdir = os.path.join(db_dir, "kmers")
    native = _node_lists_native(kdir, ids, n_rows)
    if native is not None:
        return native
    lists = []
    for i in ids:
        with open(os.path.join(kdir, str(i)), "rb") as f:
            first = f.readline()
        r = (np.array(first.split(), dtype=np.int64) if len(first) < 4096 else np.fromstring(first, dtype=np.int64, sep=" "))
        if r.size and (r.min() < 0 or r.max() >= max(1, n_rows)):
            raise ValueError("%s/kmers/%d lists a row outside kmer.fa" % (db_dir, i))
        lists.append(r.astype(np.uint32))
    offs = np.zeros(len(ids) + 1, np.int64)
    for i, r in enumerate(lists):
        offs[i + 1] = offs[i] + r.size
    rows = np.concatenate(lists) if lists else np.zeros(0, np.uint32)
    # set(map(int, ...)) of identify.py:118: lists that are already strictly increasing (the builder writes them
    # so) are their own de-duplicated form
    inc = rows[1:] > rows[:-1] if rows.size > 1 else np.ones(0, bool)
    if rows.size > 1:
        inc[offs[1:-1][(offs[1:-1] > 0) & (offs[1:-1] < rows.size)] - 1] = True
    if inc.all():
        urows, uoffs = rows, offs
    else:
        ul = [np.unique(r) for r in lists]
        uoffs = np.zeros(len(ids) + 1, np.int64)
        for i, r in enumerate(ul):
            uoffs[i + 1] = uoffs[i] + r.size
        urows = np.concatenate(ul) if ul else np.zeros(0, np.uint32)
    return rows, offs, urows, uoffs


def load_tree(db_dir, k=L1_K, with_keys=None):
    """kmer.fa + kmers/<id> -> TreeArrays.  The text/directory parse (tens of millions of tokens) is done
    once per database: a binary image is kept under SS_IMAGE_CACHE (default ~/.cache/strainscan_amd), keyed
    by the database path, size and mtime of kmer.fa.

    `with_keys(keys, flags)`: called on this thread as soon as kmer.fa is encoded -- the caller builds the device index
    there (native code, the interpreter lock is free) WHILE a worker thread parses the 1645 node files (numpy, 0.7 s
    for an E. coli tree); its result comes back as the second element of the returned pair."""
    fa = os.path.join(db_dir, "kmer.fa")
    st = os.stat(fa)
    kdir = os.path.join(db_dir, "kmers")
    tag = cache_tag("%s|%d|%d|%d|%d" % (os.path.realpath(db_dir), st.st_size, st.st_mtime_ns, k,
                                             os.stat(kdir).st_mtime_ns))
    cdir = _cache_dir()
    path = os.path.join(cdir, "tree_%s.bin" % tag) if cdir else None
    if path:
        _join_cache_writer(path)          # an earlier load_tree of this database may still be writing it
    if path and os.path.exists(path):
        try:
            t = _read_tree_cache(path)
            return (t, with_keys(t.keys, t.flags)) if with_keys else t
        except (ValueError, OSError):
            pass
    n = _lib.C.c_uint64()
    _lib.check(_lib.lib().ss_kmerfa_count_rows(os.fsencode(fa), _lib.C.byref(n)), "ss_kmerfa_count_rows(%s)" % fa)
    keys = np.empty(n.value, np.uint64)
    flags = np.empty(n.value, np.uint8)
    _lib.check(_lib.lib().ss_kmerfa_encode(os.fsencode(fa), int(k), n.value, _lib.ptr(keys), _lib.ptr(flags), 0),
               "ss_kmerfa_encode(%s)" % fa)
    ids = sorted(int(f) for f in os.listdir(kdir) if f.isdigit())
    extra = None
    if with_keys:
        import threading
        box = {}

        def work():
            try:
                box["lists"] = _node_lists(db_dir, ids, n.value)
            except BaseException as e:  # noqa: B902 -- handed to the caller's thread
                box["err"] = e

        th = threading.Thread(target=work)
        th.start()
        try:
            extra = with_keys(keys, flags)
        finally:
            th.join()
        if "err" in box:
            raise box["err"]
        rows, offs, urows, uoffs = box["lists"]
    else:
        rows, offs, urows, uoffs = _node_lists(db_dir, ids, n.value)
    t = TreeArrays(keys, flags, ids, rows, offs, urows, uoffs)
    if path:
        _write_tree_cache_later(cdir, path, t)
    return (t, extra) if with_keys else t


def load_tree_text(db_dir, k=L1_K):
    """-> (keys u64[n], flags u8[n], node ids, node row lists in FILE order); see load_tree."""
    t = load_tree(db_dir, k)
    return t.keys, t.flags, t.ids, t.lists


class TreeImage:
    def __init__(self, db_dir, upper_keys=True):
        self.db_dir = db_dir
        self.upper_keys = upper_keys
        t, self.kdb = rank0_first(lambda: load_tree(db_dir, L1_K, with_keys=lambda keys, flags:
                                                    self._index(db_dir, keys, flags, upper_keys)))
        self.node_ids = list(t.ids)
        self.node_rows = dict(zip(t.ids, t.lists))   # id -> rows in FILE order (adjust_profile indexes it)
        self.node_index = {i: j for j, i in enumerate(self.node_ids)}
        self.nodes = _lib.NodeSet.from_sorted(t.urows, t.uoffs)
        self._scanned = None          # key of the inputs whose counts are in the table
        self._counts = None
        self._stats = None
        self._rows_global = True      # False: several ranks, the table holds THIS rank's counts only

    def __del__(self):
        try:
            wait_cache_writes()      # the index may still be on its way to the image cache
        except Exception:            # noqa: B902 -- interpreter shutdown
            pass

    @staticmethod
    def _index(db_dir, keys, flags, upper_keys):
        """Device index of kmer.fa: imported from the image cache when present, else built and exported."""
        cdir = _cache_dir()
        path = None
        if cdir:
            st = os.stat(os.path.join(db_dir, "kmer.fa"))
            tag = cache_tag("%s|%d|%d|%d|%d" % (os.path.realpath(db_dir), st.st_size, st.st_mtime_ns, L1_K,
                                                     int(upper_keys)))
            path = os.path.join(cdir, "index_%s.bin" % tag)
            if os.path.exists(path):
                try:
                    kdb = _lib.KmerDB.from_image(path)
                    if kdb.n_rows == keys.size:
                        INDEX_EVENTS["imported"] += 1
                        return kdb
                    kdb.close()
                except RuntimeError:
                    pass
        kdb = _lib.KmerDB(keys, flags, L1_K, upper_keys)
        INDEX_EVENTS["built"] += 1
        if path:
            _export_image_later(kdb, cdir, path)
        return kdb

    # -- scanning ---------------------------------------------------------------------------
    def scan(self, paths):
        """Count the table's k-mers in the given FASTA/FASTQ(.gz) files (replaces the
        `jellyfish count` + `dump -c` pair, identify.py:82-87)."""
        key = tuple((os.path.abspath(p), os.path.getmtime(p), os.path.getsize(p)) for p in paths if p)
        if self._scanned == key:
            return
        from . import dist
        scan_into(self.kdb, paths, allreduce=False)
        self._rows_global = not dist.is_distributed()
        self._scanned = key
        self._counts = None
        self._stats = None

    def mark_external(self, tag):
        """The table now holds counts produced elsewhere (multi-GPU: every rank scanned a shard,
        the row vectors were all-reduced and loaded back with KmerDB.load_counts_rows_dev)."""
        self._scanned = ("external", tag)
        self._counts = None
        self._stats = None
        self._rows_global = True

    @property
    def is_external(self):
        return bool(self._scanned) and self._scanned[0] == "external"

    def _global_rows(self):
        """Several ranks: single rows are asked for (adjust_profile's Poisson branch, identify.py:203-218; the `remain`
        set, :181-189) -> now the whole row vector is summed over the ranks and loaded back.  A collective: every rank
        runs the same walk on the same node statistics, so all of them get here at the same point."""
        if not self._rows_global:
            from . import dist
            self.node_stats()                     # while the table still holds this rank's own counts
            dist.allreduce_table(self.kdb)
            self._rows_global = True

    @property
    def counts(self):
        if self._counts is None:
            self._global_rows()
            self._counts = self.kdb.counts_rows()
        return self._counts

    @property
    def valid(self):
        return self.kdb.row_valid

    def match_results(self):
        return CountsView(self.counts, self.valid)

    def node_stats(self):
        """match_node + del_outlier for every node in one launch (identify.py:106-127)."""
        if self._stats is None:
            # harvest path (ss_nodes.hip): one streaming pass over the counters, reductions over the nodes with hits;
            # several ranks: their counts are exchanged in between (a few MB instead of the row vector)
            from . import dist
            between = None
            if not self._rows_global:
                between = lambda ns: dist.exchange_touched(ns, device=dist.exchange_device())    # noqa: E731
            self._stats = self.nodes.harvest(self.kdb, between)
        return self._stats

    def rows_stat(self, rows):
        """Ad-hoc row set (adjust_profile's `remain`, identify.py:181-189)."""
        self._global_rows()
        return _lib.rows_reduce(self.kdb, rows)


_CACHE = {}


def tree_image(db_dir, upper_keys=True):
    key = (os.path.realpath(db_dir), bool(upper_keys), os.path.getmtime(os.path.join(db_dir, "kmer.fa")))
    img = _CACHE.get(key)
    if img is None:
        img = TreeImage(db_dir, upper_keys)
        if _CACHE:
            wait_cache_writes()      # (an image that is still being exported must not be dropped)
        _CACHE.clear()               # one database image at a time on the device
        _CACHE[key] = img
    return img


def clear_cache():
    wait_cache_writes()
    _CACHE.clear()
    for rs in _READS.values():
        rs.close()
    _READS.clear()

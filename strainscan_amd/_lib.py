"""ctypes binding of libstrainscan_hip.so (the C ABI declared in include/strainscan_hip.h).

Fails loudly: a missing library is an ImportError-like RuntimeError, a missing GPU surfaces as
SSError(SS_ENODEV) from the first call that needs the device.  Nothing here falls back to a CPU
implementation.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SS_LIB") or os.path.join(_HERE, "lib", "libstrainscan_hip.so")

SS_OK, SS_EINVAL, SS_ENOMEM, SS_EIO, SS_EHIP, SS_ENODEV, SS_EKEY, SS_ERANGE = 0, -22, -12, -5, -1000, -19, -2, -34
SS_EAGAIN = -11
GZ_CHAIN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p)      # ss_gz_chain_fn (strainscan_hip.h)
NO_CHAIN = C.cast(None, GZ_CHAIN_FN)
ROW_VALID, ROW_LOWER = 1, 2


def cli_clock(what):
    """SS_CLI_TRACE=1: seconds since the interpreter started, at the milestones of a run (stderr).  scripts/bench_cli_l2.py
    reads these lines to say where a fresh `strainscan` process spends its time."""
    if os.environ.get("SS_CLI_TRACE"):
        import sys
        import time
        try:
            import psutil
            t0 = psutil.Process().create_time()
        except Exception:                   # noqa: B902
            t0 = time.time()
        sys.stderr.write("[cli] %-44s %.3f s after process start\n" % (what, time.time() - t0))


class SSError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        lib = _lib
        msg = lib.ss_strerror(code).decode() if lib is not None else str(code)
        if code == SS_EHIP and lib is not None:
            msg += ": " + lib.ss_last_error().decode()
        super().__init__("%s failed: %s (%d)" % (where, msg, code))


class NodeStat(C.Structure):
    _fields_ = [("length", C.c_uint32), ("n_pos", C.c_uint32), ("n_kept", C.c_uint32), ("reserved", C.c_uint32),
                ("sum_kept", C.c_uint64), ("median2", C.c_uint64)]


NODE_STAT_DTYPE = np.dtype([("length", "<u4"), ("n_pos", "<u4"), ("n_kept", "<u4"), ("reserved", "<u4"),
                            ("sum_kept", "<u8"), ("median2", "<u8")])

_lib = None

u64, u32, i32, vp, cp = C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_char_p
P = C.POINTER

# name -> (restype, argtypes); kept in one table so tests can check it against the header
SIGNATURES = {
    "ss_version": (i32, []),
    "ss_strerror": (cp, [i32]),
    "ss_last_error": (cp, []),
    "ss_device_count": (i32, [P(i32)]),
    "ss_set_device": (i32, [i32]),
    "ss_device_sync": (i32, []),
    "ss_stream_sync": (i32, [vp]),
    "ss_dev_alloc": (i32, [P(vp), u64]),
    "ss_dev_alloc_async": (i32, [P(vp), u64]),
    "ss_dev_free_async": (i32, [vp]),
    "ss_dev_free": (i32, [vp]),
    "ss_memcpy_h2d": (i32, [vp, vp, u64, vp]),
    "ss_memcpy_d2h": (i32, [vp, vp, u64, vp]),
    "ss_memset_dev": (i32, [vp, i32, u64, vp]),
    "ss_revcomp": (i32, [cp, cp, u64]),
    "ss_shuffle_split_bits": (i32, [u64, i32, u64, u32, vp]),
    "ss_gz_inflate": (i32, [cp, i32, i32, P(vp), P(u64)]),
    "ss_gz_inflate_gpu": (i32, [cp, P(vp), P(u64)]),
    "ss_gz_gpu_counters": (i32, [P(u64), P(u64)]),
    "ss_gz_gpu_release": (i32, []),
    "ss_dev_big_blocks": (i32, [P(u64)]),
    "ss_dev_big_release": (i32, []),
    "ss_gz_set_policy": (i32, [i32]),
    "ss_gz_set_range": (i32, [i32, i32, u64, GZ_CHAIN_FN, vp]),
    "ss_gz_range_counters": (i32, [P(u64), P(u64)]),
    "ss_test_hook": (i32, [i32, C.c_longlong]),
    "ss_gz_free": (None, [vp]),
    "ss_gz_inflate_to_file": (i32, [cp, cp, i32, P(u64)]),
    "ss_host_cpus": (i32, []),
    "ss_revcomp_dev": (i32, [vp, vp, u64, u64, vp]),
    "ss_kmerfa_count_rows": (i32, [cp, P(u64)]),
    "ss_node_lists_parse": (i32, [cp, vp, C.c_uint32, u64, vp, vp, vp, P(C.c_uint32)]),
    "ss_kmerfa_encode": (i32, [cp, i32, u64, vp, vp, i32]),
    "ss_kmerfa_encode_mem": (i32, [cp, u64, i32, u64, vp, vp]),
    "ss_encode_kmer": (i32, [cp, i32, P(u64)]),
    "ss_db_build": (i32, [vp, vp, u64, i32, i32, P(vp)]),
    "ss_db_destroy": (i32, [vp]),
    "ss_db_export": (i32, [vp, cp]),
    "ss_db_import": (i32, [cp, P(vp)]),
    "ss_db_info": (i32, [vp, P(u64), P(u64), P(u64), P(i32)]),
    "ss_db_row_valid": (i32, [vp, vp]),
    "ss_db_row_valid_dev": (vp, [vp]),
    "ss_db_device_bytes": (u64, [vp]),
    "ss_db_index_info": (i32, [vp, vp]),
    "ss_db_expect_hits": (i32, [vp, i32]),
    "ss_db_probe_info": (i32, [vp, vp]),
    "ss_scan_reads_multi": (i32, [vp, i32, vp, vp]),
    "ss_scan_reset": (i32, [vp, vp]),
    "ss_scan_flat_dev": (i32, [vp, vp, u64, vp]),
    "ss_scan_flat_host": (i32, [vp, cp, u64]),
    "ss_scan_files": (i32, [vp, P(cp), i32, P(u64), P(u64)]),
    "ss_scan_files_shard": (i32, [vp, P(cp), i32, i32, i32, P(u64), P(u64)]),
    "ss_counts_rows_dev": (i32, [vp, vp, vp]),
    "ss_counts_rows": (i32, [vp, vp]),
    "ss_counts_load_rows_dev": (i32, [vp, vp, vp]),
    "ss_scan_kernel_launches": (u64, [vp]),
    "ss_ingest_warm_up": (i32, []),
    "ss_ingest_threads": (i32, [P(i32)]),
    "ss_gz_warm_up": (i32, [i32]),
    "ss_reads_load": (i32, [P(cp), i32, i32, i32, P(vp)]),
    "ss_reads_from_flat_dev": (i32, [vp, u64, i32, P(vp)]),
    "ss_reads_read_back": (i32, [vp, vp, u64, P(u64)]),
    "ss_reads_destroy": (i32, [vp]),
    "ss_reads_info": (i32, [vp, P(u64), P(u64), P(u64), P(u64)]),
    "ss_scan_reads": (i32, [vp, vp, vp]),
    "ss_reads_order_timing": (i32, [vp]),
    "ss_reads_order_counters": (i32, [P(u64)]),
    "ss_fastx_to_flat": (i32, [cp, u64, vp, P(u64), P(u64)]),
    "ss_reader_open": (i32, [P(cp), i32, P(vp)]),
    "ss_reader_set_overlap": (i32, [vp, i32]),
    "ss_reader_next": (i32, [vp, vp, u64, P(u64), P(u64)]),
    "ss_reader_close": (i32, [vp]),
    "ss_nodes_create": (i32, [vp, vp, u32, P(vp)]),
    "ss_nodes_destroy": (i32, [vp]),
    "ss_nodes_reduce_dev": (i32, [vp, vp, vp, vp, vp]),
    "ss_nodes_reduce": (i32, [vp, vp, vp]),
    "ss_nodes_bind": (i32, [vp, vp]),
    "ss_nodes_harvest_dev": (i32, [vp, vp, vp]),
    "ss_nodes_reduce_touched_dev": (i32, [vp, vp, vp]),
    "ss_nodes_touched_get_dev": (i32, [vp, vp, vp]),
    "ss_nodes_touched_set_dev": (i32, [vp, vp, vp]),
    "ss_nodes_pack_dev": (i32, [vp, vp, u64, P(u64), vp]),
    "ss_nodes_unpack_dev": (i32, [vp, vp, vp]),
    "ss_nodes_pack_capped_dev": (i32, [vp, vp, u64, vp, vp]),
    "ss_nodes_unpack_capped_dev": (i32, [vp, vp, u64, vp]),
    "ss_nodes_clear_dev": (i32, [vp, vp]),
    "ss_rows_reduce": (i32, [vp, vp, u64, P(NodeStat)]),
    "ss_l2_create": (i32, [vp, vp, u64, u32, P(vp)]),
    "ss_l2_create_dev": (i32, [vp, vp, u64, u32, P(vp)]),
    "ss_npz_member_dev": (i32, [C.c_char_p, u64, u64, C.c_uint32, u64, i32, P(vp), P(u64), P(vp)]),
    "ss_npz_member_done": (i32, [vp]),
    "ss_crc32_repeat": (i32, [C.c_uint32, i32, u64, P(C.c_uint32)]),
    "ss_l2_create_planes": (i32, [vp, u64, u32, P(vp)]),
    "ss_l2_export_planes": (i32, [vp, vp]),
    "ss_l2_destroy": (i32, [vp]),
    "ss_l2_info": (i32, [vp, P(u64), P(u32), P(u64)]),
    "ss_l2_popc2": (i32, [vp, vp, vp, vp, vp]),
    "ss_l2_andnot_col": (i32, [vp, u32, vp]),
    "ss_l2_set_overlap": (i32, [vp, vp, vp, vp, u32]),
    "ss_l2_prepare": (i32, [vp, vp, vp, C.c_double, C.c_double, C.c_double, vp, vp, vp, vp, vp, vp, vp]),
    "ss_l2_import": (i32, [C.c_char_p, u64, C.c_uint32, u64, u64, u64, u64, u64, C.c_uint32, vp]),
    "ss_l2_fold": (i32, [vp, vp, vp, u64, vp]),
    "ss_l2_fold_train": (i32, [vp, vp, vp, u64, i32, vp]),
    "ss_split_dev_start": (i32, [u64, i32, u64, C.c_uint32, vp]),
    "ss_split_dev_wait": (i32, [vp, vp, vp]),
    "ss_split_dev_free": (i32, [vp]),
    "ss_l2_count_keep": (i32, [vp, u64, C.c_double, C.c_double, C.c_double, vp]),
    "ss_l2_quantile_sums": (i32, [vp, vp, vp, u32, C.c_double, C.c_double, vp, vp, vp, vp, vp]),
    "ss_l2_pattern_stats": (i32, [vp, vp, i32, vp, vp, i32, vp]),
    "ss_enet_path_gram": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, i32, C.c_double, i32, C.c_double, i32, vp, vp,
                                vp, vp, vp]),
    "ss_enet_cd": (i32, [vp, vp, u64, i32, C.c_double, C.c_double, i32, C.c_double, i32, vp, vp, vp]),
}


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (same soname as
    /opt/rocm's); whichever is loaded first serves both.  If this library were loaded first it
    would pull /opt/rocm's runtime and a later `import torch` (multi-GPU plumbing, bench.py)
    then fails with "No HIP GPUs are available".  So when torch is installed, map ITS runtime
    first; without torch the RUNPATH of libstrainscan_hip.so finds /opt/rocm's."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


_LIB_LOCK = __import__("threading").Lock()


def lib():
    """The loaded library; raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _LIB_LOCK:
        return _load()


def warm_up(ingest=False, gz=0):
    """Load the library and start the HIP runtime (0.1-0.2 s in a fresh process); `ingest`: also the one-time costs of the
    first read of plain-text files (pinned parse buffers, streams: another 0.1 s); `gz`: the pinned upload buffers of that
    many .gz inputs -- the CLI calls this on a worker thread while the interpreter is still importing modules and parsing
    arguments."""
    try:
        if device_count() > 0 and ingest:
            lib().ss_ingest_warm_up()
        if device_count() > 0 and gz:
            lib().ss_gz_warm_up(int(gz))

    except Exception:                       # noqa: B902 -- whoever needs the GPU next gets the real error
        pass


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "strainscan_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C strainscan_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = library older than the binding
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, where):
    if rc != SS_OK:
        if rc == SS_EKEY:
            raise KeyError("%s: a database k-mer has no owning row (identify_low_mem.py:88)" % where)
        raise SSError(rc, where)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class gz_policy:
    """with gz_policy(1): ...  -- who inflates .gz inputs inside the block (ss_gz_set_policy: 1 = the device or SS_EAGAIN,
    2 = the host inflaters); back to 0 (device, then host for what it declines) afterwards."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        check(lib().ss_gz_set_policy(self.mode), "ss_gz_set_policy")
        return self

    def __exit__(self, *exc):
        lib().ss_gz_set_policy(0)
        return False


def device_count():
    n = C.c_int(0)
    rc = lib().ss_device_count(C.byref(n))
    return n.value if rc == SS_OK else 0


def require_gpu():
    if device_count() < 1:
        raise SSError(SS_ENODEV, "strainscan_amd (no MI355X visible; there is no CPU fallback)")


# -------------------------------------------------------------------------------------------------
class KmerDB:
    """Device k-mer table built from a k-mer FASTA (`--if` argument of the reference's jellyfish
    calls, library/identify.py:82-86) with per-row bookkeeping of identify.py:90-101."""

    def __init__(self, keys, flags, k=31, upper_keys=True):
        require_gpu()
        keys = np.ascontiguousarray(keys, np.uint64)
        flags = np.ascontiguousarray(flags, np.uint8)
        assert keys.shape == flags.shape
        h = C.c_void_p()
        check(lib().ss_db_build(ptr(keys), ptr(flags), keys.size, int(k), int(upper_keys), C.byref(h)),
              "ss_db_build")
        self._h = h
        self.k = int(k)
        self.n_rows = int(keys.size)
        self._row_valid = None

    @classmethod
    def from_image(cls, path):
        """A previously exported index image (ss_db_export); raises SSError if it is not usable."""
        require_gpu()
        h = C.c_void_p()
        check(lib().ss_db_import(os.fsencode(path), C.byref(h)), "ss_db_import(%s)" % path)
        self = cls.__new__(cls)
        self._h = h
        self._row_valid = None
        i = self.info()
        self.k, self.n_rows = i["k"], i["n_rows"]
        return self

    def export(self, path):
        check(lib().ss_db_export(self._h, os.fsencode(path)), "ss_db_export(%s)" % path)

    @classmethod
    def from_fasta(cls, path, k=31, upper_keys=True, threads=0):
        n = C.c_uint64()
        check(lib().ss_kmerfa_count_rows(os.fsencode(path), C.byref(n)), "ss_kmerfa_count_rows(%s)" % path)
        keys = np.empty(n.value, np.uint64)
        flags = np.empty(n.value, np.uint8)
        check(lib().ss_kmerfa_encode(os.fsencode(path), int(k), n.value, ptr(keys), ptr(flags), threads),
              "ss_kmerfa_encode(%s)" % path)
        return cls(keys, flags, k, upper_keys)

    @classmethod
    def from_text(cls, text, k=31, upper_keys=True):
        keys, flags = encode_kmer_fasta(text, k)
        return cls(keys, flags, k, upper_keys)

    def close(self):
        if getattr(self, "_h", None):
            lib().ss_db_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def info(self):
        a, b, c, k = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int()
        check(lib().ss_db_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(k)), "ss_db_info")
        ix = np.zeros(8, np.uint64)
        check(lib().ss_db_index_info(self._h, ptr(ix)), "ss_db_index_info")
        return dict(n_rows=a.value, n_distinct=b.value, capacity=c.value, k=k.value,
                    device_bytes=int(lib().ss_db_device_bytes(self._h)), layout=int(ix[0]), n_slots=int(ix[1]),
                    n_buckets=int(ix[2]), n_dir=int(ix[3]), filter_bits=int(ix[4]), n_mslots=int(ix[6]), n_inline=int(ix[7]))

    @property
    def row_valid(self):
        if self._row_valid is None:
            v = np.zeros(self.n_rows, np.uint8)
            check(lib().ss_db_row_valid(self._h, ptr(v)), "ss_db_row_valid")
            self._row_valid = v
        return self._row_valid

    @property
    def row_valid_dev(self):
        return lib().ss_db_row_valid_dev(self._h)

    def expect_hits(self, expect=True):
        """Hint: most read k-mers are in this table (a layer-2 cluster table)."""
        check(lib().ss_db_expect_hits(self._h, 1 if expect else 0), "ss_db_expect_hits")
        return self

    def probe_info(self):
        """What the last binned scan's probe found for this table: dict(set, comb, runs_per_tile)."""
        out = np.zeros(3, np.uint64)
        check(lib().ss_db_probe_info(self._h, ptr(out)), "ss_db_probe_info")
        return dict(set=int(out[0]), comb=bool(out[1]), runs_per_tile=float(out[2]) / 1000.0)

    def reset(self, stream=None):
        check(lib().ss_scan_reset(self._h, stream), "ss_scan_reset")

    def scan_flat_dev(self, dptr, n, stream=None):
        check(lib().ss_scan_flat_dev(self._h, dptr, int(n), stream), "ss_scan_flat_dev")

    def scan_flat(self, bases):
        b = bytes(bases) if not isinstance(bases, bytes) else bases
        check(lib().ss_scan_flat_host(self._h, b, len(b)), "ss_scan_flat_host")

    def scan_files(self, paths, shard_rank=0, shard_world=1):
        """Count in the reads of the files; with shard_world > 1 only this rank's share of them (parsed, copied, scanned)."""
        paths = [os.fsencode(p) for p in paths if p]
        arr = (C.c_char_p * len(paths))(*paths)
        nrec, nb = C.c_uint64(), C.c_uint64()
        check(lib().ss_scan_files_shard(self._h, arr, len(paths), int(shard_rank), int(shard_world), C.byref(nrec), C.byref(nb)),
              "ss_scan_files_shard")
        return nrec.value, nb.value

    def counts_rows(self):
        out = np.zeros(self.n_rows, np.uint32)
        check(lib().ss_counts_rows(self._h, ptr(out)), "ss_counts_rows")
        return out

    def load_counts_rows_dev(self, dptr, stream=None):
        check(lib().ss_counts_load_rows_dev(self._h, dptr, stream), "ss_counts_load_rows_dev")

    def counts_rows_dev(self, dptr, stream=None):
        check(lib().ss_counts_rows_dev(self._h, dptr, stream), "ss_counts_rows_dev")


class ReadSet:
    """FASTA/FASTQ(.gz) files parsed once and kept in HBM as flat base blocks."""

    def __init__(self, paths, shard_rank=0, shard_world=1):
        require_gpu()
        ps = [os.fsencode(p) for p in paths if p]
        arr = (C.c_char_p * len(ps))(*ps)
        h = C.c_void_p()
        check(lib().ss_reads_load(arr, len(ps), int(shard_rank), int(shard_world), C.byref(h)), "ss_reads_load")
        self._h = h

    @classmethod
    def from_flat_dev(cls, dptr, n, order=True):
        """A resident read set from a flat base block already on the device (copied; `order`: locality order)."""
        require_gpu()
        self = cls.__new__(cls)
        h = C.c_void_p()
        check(lib().ss_reads_from_flat_dev(dptr, int(n), int(bool(order)), C.byref(h)), "ss_reads_from_flat_dev")
        self._h = h
        return self

    def close(self):
        if getattr(self, "_h", None):
            lib().ss_reads_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def read_back(self):
        """The resident flat blocks as bytes (tests)."""
        n = C.c_uint64()
        check(lib().ss_reads_read_back(self._h, None, 0, C.byref(n)), "ss_reads_read_back")
        buf = np.zeros(max(1, n.value), np.uint8)
        check(lib().ss_reads_read_back(self._h, ptr(buf), buf.size, C.byref(n)), "ss_reads_read_back")
        return buf[:n.value].tobytes()

    def info(self):
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib().ss_reads_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "ss_reads_info")
        return dict(n_records=a.value, n_bases=b.value, n_blocks=c.value, device_bytes=d.value)

    def scan_into_many(self, kdbs, stream=None):
        """One pass over the resident reads for several tables (ss_scan_reads_multi); counts as scan_into on each."""
        arr = (C.c_void_p * len(kdbs))(*[k.handle for k in kdbs])
        check(lib().ss_scan_reads_multi(arr, len(kdbs), self._h, stream), "ss_scan_reads_multi")

    def scan_into(self, kdb, stream=None):
        check(lib().ss_scan_reads(kdb.handle, self._h, stream), "ss_scan_reads")


class NodeSet:
    """All tree-node k-mer row lists on the device (files <db>/kmers/<id>, identify.py:116-118)."""

    def __init__(self, row_lists):
        require_gpu()
        offs = np.zeros(len(row_lists) + 1, np.uint64)
        dedup = []
        for i, r in enumerate(row_lists):
            r = np.unique(np.asarray(r, np.int64))  # set(map(int, ...)) at identify.py:118
            dedup.append(r.astype(np.uint32))
            offs[i + 1] = offs[i] + r.size
        rows = np.concatenate(dedup) if dedup else np.zeros(0, np.uint32)
        rows = np.ascontiguousarray(rows, np.uint32)
        h = C.c_void_p()
        check(lib().ss_nodes_create(ptr(rows), ptr(offs), len(row_lists), C.byref(h)), "ss_nodes_create")
        self._h = h
        self.n_nodes = len(row_lists)
        self.n_rows_total = int(offs[-1])

    @classmethod
    def from_sorted(cls, rows, offsets):
        """Node lists that are already de-duplicated and sorted, back to back: rows u32[offsets[-1]],
        offsets[n_nodes + 1] (no per-list np.unique: 0.1 s for the 1645 lists of an E. coli database)."""
        require_gpu()
        rows = np.ascontiguousarray(rows, np.uint32)
        offs = np.ascontiguousarray(offsets, np.uint64)
        self = cls.__new__(cls)
        h = C.c_void_p()
        check(lib().ss_nodes_create(ptr(rows), ptr(offs), offs.size - 1, C.byref(h)), "ss_nodes_create")
        self._h = h
        self.n_nodes = offs.size - 1
        self.n_rows_total = int(offs[-1]) if offs.size else 0
        return self

    def close(self):
        if getattr(self, "_h", None):
            lib().ss_nodes_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reduce(self, db):
        st = np.zeros(self.n_nodes, NODE_STAT_DTYPE)
        check(lib().ss_nodes_reduce(self._h, db.handle, ptr(st)), "ss_nodes_reduce")
        return st

    # -- harvest path (ss_nodes.hip): bind once per database, then per scan harvest + reduce over the touched nodes
    def bind(self, db):
        check(lib().ss_nodes_bind(self._h, db.handle), "ss_nodes_bind")
        self._bound = db
        return self

    def harvest_dev(self, db, stream=None):
        check(lib().ss_nodes_harvest_dev(self._h, db.handle, stream), "ss_nodes_harvest_dev")

    def reduce_touched_dev(self, stats_dptr, stream=None):
        check(lib().ss_nodes_reduce_touched_dev(self._h, stats_dptr, stream), "ss_nodes_reduce_touched_dev")

    def harvest(self, db, between=None):
        """Statistics of all nodes from the counters of `db` (bound): harvest, [between(self): the multi-GPU
        exchange], reduce over the touched nodes.  `between` may return an object with .complete() (read after the
        statistics have arrived: dist.exchange_touched's capped packing never waits for the host; when the packed counts
        did not fit its buffer, the harvest is simply run again -- the counters are still in the table -- and the
        exchange repeats with the larger buffer it has sized meanwhile; all ranks see the same total and repeat together)."""
        if getattr(self, "_bound", None) is not db:
            self.bind(db)
        dst = C.c_void_p()
        nbytes = max(1, self.n_nodes) * NODE_STAT_DTYPE.itemsize
        check(lib().ss_dev_alloc(C.byref(dst), nbytes), "ss_dev_alloc")
        dirty = False
        try:
            for _ in range(3):
                dirty = True
                self.harvest_dev(db)
                pending = between(self) if between is not None else None
                self.reduce_touched_dev(dst)          # clears the count buffer and the flags
                dirty = False
                st = np.zeros(self.n_nodes, NODE_STAT_DTYPE)
                if self.n_nodes:
                    check(lib().ss_memcpy_d2h(ptr(st), dst, self.n_nodes * NODE_STAT_DTYPE.itemsize, None), "ss_memcpy_d2h")
                check(lib().ss_device_sync(), "ss_device_sync")
                if pending is None or not hasattr(pending, "complete") or pending.complete():
                    return st
            raise RuntimeError("exchange of the touched nodes did not fit its buffer after three rounds")
        finally:
            if dirty:
                # harvest or the exchange raised: the dense buffer still holds counts that the next harvest would add to
                lib().ss_nodes_clear_dev(self._h, None)
                lib().ss_device_sync()
            lib().ss_dev_free(dst)

    def reduce_dev(self, counts_rows_dptr, row_valid_dptr, stats_dptr, stream=None):
        check(lib().ss_nodes_reduce_dev(self._h, counts_rows_dptr, row_valid_dptr, stats_dptr, stream),
              "ss_nodes_reduce_dev")


def rows_reduce(db, rows):
    rows = np.ascontiguousarray(np.unique(np.asarray(rows, np.int64)), np.uint32)
    st = NodeStat()
    check(lib().ss_rows_reduce(db.handle, ptr(rows), rows.size, C.byref(st)), "ss_rows_reduce")
    return dict(length=st.length, n_pos=st.n_pos, n_kept=st.n_kept, sum_kept=st.sum_kept, median2=st.median2)


def gz_inflate(path, threads=0, mode=0):
    """Whole-file gunzip as the ingest does it (ss_gz_inflate): bytes, or None when the file is not inflated this
    way (see include/strainscan_hip.h).  mode 1 = threaded inflater only, 2 = libdeflate only."""
    text, n = C.c_void_p(), C.c_uint64()
    rc = lib().ss_gz_inflate(os.fsencode(path), int(threads), int(mode), C.byref(text), C.byref(n))
    if rc != SS_OK:
        return None
    try:
        return C.string_at(text, n.value)
    finally:
        lib().ss_gz_free(text)


def encode_kmer_fasta(text, k=31):
    nl = text.count(b"\n") + (1 if text and not text.endswith(b"\n") else 0)
    n = nl // 2
    keys = np.empty(n, np.uint64)
    flags = np.empty(n, np.uint8)
    check(lib().ss_kmerfa_encode_mem(text, len(text), int(k), n, ptr(keys), ptr(flags)), "ss_kmerfa_encode_mem")
    return keys, flags


def fastx_to_flat(text):
    out = np.empty(len(text) + 2, np.uint8)
    n, nrec = C.c_uint64(), C.c_uint64()
    check(lib().ss_fastx_to_flat(text, len(text), ptr(out), C.byref(n), C.byref(nrec)), "ss_fastx_to_flat")
    return out[:n.value].tobytes(), nrec.value


def read_flat_blocks(paths, cap=32 << 20, overlap=30):
    """Generator of (flat block bytes, n_records) from FASTA/FASTQ(.gz) files."""
    paths = [os.fsencode(p) for p in paths if p]
    arr = (C.c_char_p * len(paths))(*paths)
    h = C.c_void_p()
    check(lib().ss_reader_open(arr, len(paths), C.byref(h)), "ss_reader_open")
    try:
        check(lib().ss_reader_set_overlap(h, overlap), "ss_reader_set_overlap")
        buf = np.empty(cap, np.uint8)
        while True:
            n, nrec = C.c_uint64(), C.c_uint64()
            check(lib().ss_reader_next(h, ptr(buf), cap, C.byref(n), C.byref(nrec)), "ss_reader_next")
            if n.value == 0:
                break
            yield buf[:n.value].tobytes(), nrec.value
    finally:
        lib().ss_reader_close(h)


def revcomp(s):
    b = s.encode() if isinstance(s, str) else bytes(s)
    out = C.create_string_buffer(len(b))
    check(lib().ss_revcomp(b, out, len(b)), "ss_revcomp")
    return out.raw.decode() if isinstance(s, str) else out.raw

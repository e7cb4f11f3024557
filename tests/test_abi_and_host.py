"""CPU-side checks of the boundary and of host logic that needs no device:
  - the shared library loads and exports every symbol include/strainscan_hip.h declares
    (and the ctypes table binds exactly that set);
  - the FASTA/FASTQ -> flat-block reader (record grammar, cuts with k-1 overlap, gz);
  - k-mer FASTA row encoding;
  - layer-2 host helpers (Gram from pattern statistics, alpha grid, ShuffleSplit bits, 1-SE rule);
  - report writers against the reference's recorded reports.
No compute entry point is called here."""
import ctypes
import gzip
import json
import os
import re
import sys

import numpy as np
import pytest

from tests import scenarios as sc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from tests import synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(REPO, "include", "strainscan_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from strainscan_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(L, s), "declared in strainscan_hip.h but not exported: " + s
    assert sorted(_lib.SIGNATURES) == syms, (set(syms) ^ set(_lib.SIGNATURES))
    assert _lib.lib().ss_version() >= 100
    assert _lib.lib().ss_strerror(-2).decode().startswith("k-mer without")


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product must fail loudly, not fall back."""
    from strainscan_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_lib.SSError):
        _lib.KmerDB(np.zeros(1, np.uint64), np.ones(1, np.uint8))


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "strainscan_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(root, f)


def test_kmer_fasta_rows():
    from strainscan_amd import _lib
    c = sc.f1_case()
    keys, flags = _lib.encode_kmer_fasta(c["kmer_fa"], 31)
    rows = c["kmer_fa"].split(b"\n")[1::2]
    assert len(rows) == keys.size == c["n_rows"]
    code = {65: 0, 67: 1, 84: 2, 71: 3}
    for r, k, f in zip(rows, keys, flags):
        ok = len(r) == 31 and all(ch in b"ACGTacgt" for ch in r)
        assert bool(f & 1) == ok
        if ok:
            assert int(k) == sum(code[ch & 0xDF] << (2 * i) for i, ch in enumerate(r))
            assert bool(f & 2) == any(ch >= 97 for ch in r)


def test_kmer_fasta_file_parse_threads_and_gz(tmp_path):
    """ss_kmerfa_count_rows / ss_kmerfa_encode on a file -- mapped and cut into many chunks at arbitrary
    byte offsets (SS_HOST_THREADS), and through gzip -- equal the single-pass in-memory encoder: ragged
    rows, CR LF, trailing blanks, lower case, N, empty rows, no final newline."""
    import ctypes as C
    import gzip
    import subprocess
    import sys
    from strainscan_amd import _lib
    rs = np.random.RandomState(5)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rows = []
    for i in range(5000):
        r = lut[rs.randint(0, 4, size=31)].tobytes()
        x = rs.randint(0, 40)
        if x == 0: r = r[:17]
        elif x == 1: r = r[:10] + b"N" + r[11:]
        elif x == 2: r = r.lower()
        elif x == 3: r = r + b" \t"
        elif x == 4: r = b""
        elif x == 5: r = r + b"\r"
        rows.append(b">" + (b"h" * rs.randint(1, 9)) + b"\n" + r)
    text = b"\n".join(rows)                              # no newline at the end
    want_k, want_f = _lib.encode_kmer_fasta(text, 31)
    assert want_k.size == 5000
    plain = tmp_path / "kmer.fa"
    plain.write_bytes(text)
    gz = tmp_path / "kmer.fa.gz"
    with gzip.open(gz, "wb") as f:
        f.write(text)
    code = (
        "import sys, numpy as np, ctypes as C\n"
        "sys.path.insert(0, %r)\n"
        "from strainscan_amd import _lib\n"
        "n = C.c_uint64()\n"
        "_lib.check(_lib.lib().ss_kmerfa_count_rows(sys.argv[1].encode(), C.byref(n)), 'count')\n"
        "k = np.empty(n.value, np.uint64); f = np.empty(n.value, np.uint8)\n"
        "_lib.check(_lib.lib().ss_kmerfa_encode(sys.argv[1].encode(), 31, n.value, _lib.ptr(k), _lib.ptr(f), 0), 'encode')\n"
        "np.save(sys.argv[2] + '.k.npy', k); np.save(sys.argv[2] + '.f.npy', f)\n" % ROOT)
    for path, threads in ((plain, "1"), (plain, "7"), (plain, "64"), (gz, "3")):
        out = str(tmp_path / ("o_%s_%s" % (path.name, threads)))
        r = subprocess.run([sys.executable, "-c", code, str(path), out], env=dict(os.environ, SS_HOST_THREADS=threads),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert np.array_equal(np.load(out + ".k.npy"), want_k) and np.array_equal(np.load(out + ".f.npy"), want_f), (path, threads)


def _fastq_like(rs, n_reads, read_len=150):
    lut = np.frombuffer(b"ACGT", np.uint8)
    rec = np.empty((n_reads, 2 * read_len + 12), np.uint8)
    rec[:, :8] = np.frombuffer(b"@r/12345", np.uint8)
    rec[:, 3:8] = 48 + rs.randint(0, 10, size=(n_reads, 5))
    rec[:, 8] = 10
    rec[:, 9:9 + read_len] = lut[rs.randint(0, 4, size=(n_reads, read_len))]
    rec[:, 9 + read_len] = 10
    rec[:, 10 + read_len] = 43
    rec[:, 11 + read_len] = 10
    q = np.clip(38 - np.abs(rs.normal(0, 5, size=(n_reads, read_len))).astype(np.int64), 2, 40) + 33
    rec[:, 12 + read_len:12 + 2 * read_len - 1] = q[:, :read_len - 1].astype(np.uint8)
    rec[:, -1] = 10
    return rec.tobytes()


def test_threaded_gunzip_equals_gzip(tmp_path):
    """ss_gz_inflate (one gzip member inflated by several threads: entry points inside the deflate stream,
    unknown-window symbols, CRC-checked) returns exactly what Python's gzip returns: FASTQ-like text at levels
    1/6/9, text with long and overlapping matches at every distance, incompressible stretches (stored blocks)
    between text; files it must decline (several members, too small) are declined in mode 1 and still inflated
    in mode 0 (libdeflate) when that library is present."""
    import zlib
    from strainscan_amd import _lib
    rs = np.random.RandomState(11)
    fq = _fastq_like(rs, 90000)                                  # 28 MB
    unit = _fastq_like(rs, 300)
    rep = b"".join(unit[: rs.randint(1, len(unit))] + bytes(rs.randint(65, 70, size=rs.randint(0, 40)).astype(np.uint8))
                   for _ in range(1500)) + b"A" * 300000 + b"ACGT" * 100000 + unit * 40
    mixed = b"".join(_fastq_like(rs, 4000) + rs.bytes(300000) for _ in range(12))
    cases = [("fq1", fq, 1), ("fq6", fq, 6), ("fq9", fq[: len(fq) // 2], 9), ("rep6", rep * 3, 6), ("mixed6", mixed, 6)]
    have_libdeflate = None
    for name, data, level in cases:
        p = tmp_path / (name + ".gz")
        p.write_bytes(gzip.compress(data, level))
        size = p.stat().st_size
        for threads in (2, 3, 8):
            got = _lib.gz_inflate(str(p), threads, 1)
            if size >= threads * (2 << 20) and name != "mixed6":
                assert got is not None, (name, threads, size)
            if got is not None:
                assert got == data, (name, threads)
        auto = _lib.gz_inflate(str(p), 0, 0)
        if have_libdeflate is None:
            have_libdeflate = _lib.gz_inflate(str(p), 0, 2) is not None
        if auto is not None:
            assert auto == data, name
        else:
            assert not have_libdeflate
    # several large members (lanes concatenated with cat): member ends are discovered, every trailer is checked
    third = len(fq) // 3
    multi = tmp_path / "multi.gz"
    multi.write_bytes(gzip.compress(fq[:third], 6) + gzip.compress(fq[third:2 * third], 1) + gzip.compress(fq[2 * third:], 9))
    for threads in (2, 3, 5):
        assert _lib.gz_inflate(str(multi), threads, 1) == fq, threads
    # many small members (bgzip-like), a small trailing member, a file too small to split: declined by the threaded
    # inflater, inflated by libdeflate in mode 0
    tiny = tmp_path / "tiny_members.gz"
    tiny.write_bytes(b"".join(gzip.compress(fq[a:a + 60000], 6) for a in range(0, 6000000, 60000)))
    tail = tmp_path / "tail.gz"
    tail.write_bytes(gzip.compress(fq, 6) + gzip.compress(b"@r\nACGT\n+\nIIII\n", 6))
    small = tmp_path / "small.gz"
    small.write_bytes(gzip.compress(fq[:200000], 6))
    for p, data in ((tiny, fq[:6000000]), (tail, fq + b"@r\nACGT\n+\nIIII\n"), (small, fq[:200000])):
        assert _lib.gz_inflate(str(p), 4, 1) is None, p
        got = _lib.gz_inflate(str(p), 4, 0)
        assert (got == data) if have_libdeflate else (got is None)
    # bgzip (BGZF): members of <= 64 KB with their size in an extra field, an empty member at the end: inflated member
    # by member in parallel; a block with a wrong CRC makes the file go elsewhere
    def bgzf(data, level=6):
        import struct
        out = []
        for a in list(range(0, len(data), 65280)) + [len(data)]:
            blk = data[a:a + 65280] if a < len(data) else b""
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            raw = c.compress(blk) + c.flush()
            out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(raw) + 25) + raw +
                       struct.pack("<II", zlib.crc32(blk), len(blk)))
        return b"".join(out)
    bg = tmp_path / "reads.bgzf.gz"
    bg.write_bytes(bgzf(fq))
    assert gzip.decompress(bg.read_bytes()) == fq          # the writer above makes valid gzip
    for threads in (2, 3):
        assert _lib.gz_inflate(str(bg), threads, 1) == fq, threads
    raw = bytearray(bg.read_bytes())
    raw[len(raw) // 2] ^= 0x10
    (tmp_path / "bad.bgzf.gz").write_bytes(bytes(raw))
    assert _lib.gz_inflate(str(tmp_path / "bad.bgzf.gz"), 3, 1) is None
    # a damaged member is never accepted (CRC), a non-gzip file is declined
    bad = bytearray((tmp_path / "fq6.gz").read_bytes())
    bad[len(bad) // 2] ^= 0x5A
    (tmp_path / "bad.gz").write_bytes(bytes(bad))
    assert _lib.gz_inflate(str(tmp_path / "bad.gz"), 4, 0) is None
    (tmp_path / "plain.txt").write_bytes(fq[:100000])
    assert _lib.gz_inflate(str(tmp_path / "plain.txt"), 4, 0) is None


def test_threaded_gunzip_fuzz_sanitized(tmp_path):
    """The threaded inflater under AddressSanitizer + UBSan (host code, g++): a few hundred damaged gzip images
    (byte and bit flips, truncations, zeroed and random stretches) are refused or give the right text, without a
    single bad access."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = tmp_path / "pgz_fuzz"
    src = os.path.join(ROOT, "tests", "pgz_fuzz.cpp")
    r = subprocess.run(["g++", "-x", "c++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        src, "-o", str(exe), "-lz", "-lpthread"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rs = np.random.RandomState(21)
    data = _fastq_like(rs, 30000) + (b"ACGTTGCA" * 40000) + _fastq_like(rs, 8000)
    (tmp_path / "want.txt").write_bytes(data)
    (tmp_path / "in.gz").write_bytes(gzip.compress(data, 6))
    assert (tmp_path / "in.gz").stat().st_size > (5 << 20)
    # ... and a file of two members (the member loop: discovered end, cancelled look-ahead chunks, second pipeline)
    data2 = _fastq_like(rs, 50000)
    cut = len(data2) * 3 // 5
    (tmp_path / "want2.txt").write_bytes(data2)
    (tmp_path / "in2.gz").write_bytes(gzip.compress(data2[:cut], 6) + gzip.compress(data2[cut:], 1))
    for gz, want, iters in (("in.gz", "want.txt", "120"), ("in2.gz", "want2.txt", "80")):
        r = subprocess.run([str(exe), str(tmp_path / gz), str(tmp_path / want), iters], capture_output=True, text=True,
                           timeout=1500, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0 and "pgz_fuzz ok" in r.stdout, (gz, r.stdout[-500:], r.stderr[-3000:])


def test_shuffle_split_native_equals_numpy():
    """ss_shuffle_split_bits (MT19937 + masked-rejection Fisher-Yates, restated) draws exactly the test sets of
    numpy.random.RandomState(seed).permutation -- the specification sklearn's ShuffleSplit calls -- for sizes around
    every power of two it masks with, several seeds, split counts and test fractions (test_l2_host_helpers checks the
    same function against the ShuffleSplit indices recorded from scikit-learn itself)."""
    from strainscan_amd import l2
    for n in list(range(0, 20)) + [31, 32, 33, 255, 256, 257, 1000, 4095, 4096, 4097, 65535, 65536, 65537, 300001]:
        for seed in (0, 1, 12345, 2 ** 32 - 1):
            a, na = l2.shuffle_split_test_bits(n, 20, 0.5, seed)
            b, nb = l2.shuffle_split_test_bits_numpy(n, 20, 0.5, seed)
            assert na == nb and np.array_equal(a, b), (n, seed)
    for splits, frac in ((1, 0.5), (3, 0.1), (31, 0.9), (7, 1.0)):
        a, _ = l2.shuffle_split_test_bits(5003, splits, frac, 42)
        b, _ = l2.shuffle_split_test_bits_numpy(5003, splits, frac, 42)
        assert np.array_equal(a, b), (splits, frac)
    # the 64-words-per-step walk (AVX-512 hosts) takes over where a level's rows are >= 8192: sizes around its first levels
    # (a draw inside the band of 63 values it cannot decide sends a step to the scalar rule: a third of the steps at that level)
    for n in (16383, 16384, 16385, 16449, 20000, 32769, 131072, 131073, 131137, 262143, 262145, 1000003):
        for seed in (0, 77):
            a, na = l2.shuffle_split_test_bits(n, 20, 0.1, seed)
            b, nb = l2.shuffle_split_test_bits_numpy(n, 20, 0.1, seed)
            assert na == nb and np.array_equal(a, b), (n, seed)
    assert l2._lib.lib().ss_shuffle_split_bits(10, 32, 5, 0, None) != 0          # more than 31 splits: refused


def test_count_keep_equals_numpy():
    """ss_l2_count_keep = the row filter of identify_strains_L2_Enet_Pscan_new_sp.py:402-415 as numpy evaluates it on doubles
    (NaN bounds keep every row), from y alone on host threads; detect_core checks it against the device's count on every call."""
    from strainscan_amd import l2
    rng = np.random.default_rng(3)
    for n in (0, 1, 1000, 700001):
        y = rng.integers(0, 300, n).astype(np.int64)
        for b in ((10.0, 150.0, 120.0), (float("nan"), 150.0, 1e9), (0.0, 1e9, float("nan")), (5.0, 5.0, 5.0), (7.5, 7.4, 9.0)):
            d = y.astype(np.float64)
            with np.errstate(invalid="ignore"):
                want = int((~((d < b[0]) | (d > b[1]) | (d > b[2]))).sum())
            assert l2.count_keep(y, *b) == want, (n, b)


def test_shuffle_split_word_by_word_walk():
    """... and the same with SS_SPLIT_SIMD=0 (read once per process): the word-by-word walk a host without AVX-512 takes."""
    import subprocess
    code = ("import numpy as np\nfrom strainscan_amd import l2\n"
            "for n in (5003, 16385, 131073, 300001):\n"
            "    a, na = l2.shuffle_split_test_bits(n, 20, 0.1, 5)\n"
            "    b, nb = l2.shuffle_split_test_bits_numpy(n, 20, 0.1, 5)\n"
            "    assert na == nb and np.array_equal(a, b), n\n"
            "print('walk ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, SS_SPLIT_SIMD="0"), cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "walk ok" in r.stdout, r.stderr[-2000:]


def test_reader_grammar_cuts_and_gz(tmp_path):
    from oracle import oracle as orc
    from strainscan_amd import _lib
    rs = np.random.RandomState(3)
    g = synth.rand_seq(rs, 60000)
    # long multi-line FASTA records (forces cuts at cap=4096) + ragged FASTQ + blank lines + '\r'
    fa = b">c1 x\n" + b"\n".join(g[i:i + 70] for i in range(0, 30000, 70)) + b"\n>c2\n" + g[30000:41000] + b"\n"
    fq = b"".join(b"@q%d\n%s\n+\n%s\n" % (i, g[s:s + n], b"@" * n)
                  for i, (s, n) in enumerate(zip(rs.randint(0, 50000, 300), rs.randint(1, 200, 300))))
    fq += b"\n@last\r\n" + g[100:180] + b"\r\n+\r\n" + b"I" * 81 + b"\n"
    p1, p2 = tmp_path / "a.fa", tmp_path / "b.fq.gz"
    p1.write_bytes(fa)
    with gzip.open(p2, "wb") as f:
        f.write(fq)
    whole = synth.flat_bases_from_fastx(fa) + synth.flat_bases_from_fastx(fq)
    flat, nrec = _lib.fastx_to_flat(fa)
    assert flat == synth.flat_bases_from_fastx(fa) and nrec == 2
    big = list(_lib.read_flat_blocks([str(p1), str(p2)], cap=32 << 20, overlap=30))
    assert b"".join(b for b, _ in big) == whole and sum(n for _, n in big) == 2 + 301
    kms = [g[i:i + 31] for i in range(0, 45000, 5)]
    keys = np.array([orc.encode_kmer(k.decode()) for k in kms], np.uint64)
    want = orc.count_flat(keys, 31, whole)
    for k, ov in ((31, 30), (21, 20)):
        kk = np.array([orc.encode_kmer(km[:k].decode()) for km in kms], np.uint64)
        kk, first = np.unique(kk, return_index=True)
        want = orc.count_flat(kk, k, whole)
        small = list(_lib.read_flat_blocks([str(p1), str(p2)], cap=4096, overlap=ov))
        assert len(small) > 10 and all(len(b) <= 4096 for b, _ in small)
        got = sum(orc.count_flat(kk, k, b) for b, _ in small)
        assert np.array_equal(got, want), k          # cut records: every k-mer counted exactly once


def test_l2_host_helpers(golden_dir):
    from strainscan_amd import l2 as L2
    from strainscan_amd.identify_strains_L2_Enet_Pscan_new_sp import lasso_mpm
    rs = np.random.RandomState(0)
    p, n = 5, 4000
    X = (rs.random_sample((n, p)) < 0.4).astype(np.int64)
    y = rs.poisson(7, n).astype(np.int64)
    pat = (X << np.arange(p)).sum(axis=1)
    st = np.zeros((1 << p, 3), np.uint64)
    np.add.at(st[:, 0], pat, 1)
    np.add.at(st[:, 1], pat, y.astype(np.uint64))
    np.add.at(st[:, 2], pat, (y * y).astype(np.uint64))
    Q, q, yy, nn = L2.gram_from_stats(st, p)
    assert np.array_equal(Q, (X.T @ X).astype(float)) and np.array_equal(q, (X.T @ y).astype(float))
    assert yy == float(y @ y) and nn == n
    from oracle import oracle as orc
    assert np.array_equal(L2.alpha_grid(q, n), orc.alpha_grid(X, y))
    with open(os.path.join(golden_dir, "shuffle_split.json")) as f:
        g = json.load(f)
    for ns, e in g.items():
        bits, n_test = L2.shuffle_split_test_bits(int(ns))
        assert n_test == e["n_test"]
        assert sorted(np.nonzero(bits & 1)[0].tolist()) == sorted(e["test0"])
        assert set(np.nonzero(bits & (1 << 19))[0].tolist()) >= set(e["test19"])
    arrs = np.load(os.path.join(golden_dir, "l2_enet_arrays.npz"))
    with open(os.path.join(golden_dir, "l2_detect.json")) as f:
        gd = json.load(f)
    for name in ("two", "three", "many", "two_out"):
        a, _, _ = lasso_mpm(arrs[name + "_alphas"], arrs[name + "_mse_path"])
        assert a == gd[name]["alpha"]


def test_report_writers_match_reference(golden_dir, tmp_path):
    from strainscan_amd import Vote_Strain_L2_Lasso_new_sp as vote
    with open(os.path.join(golden_dir, "e2e_reports.json")) as f:
        g = json.load(f)
    with open(os.path.join(golden_dir, "report_headers.json")) as f:
        hdr = json.load(f)
    for ex in hdr.values():                       # committed Output_Example headers (format contract)
        # one identified cluster: final_report.txt is a copy of StrainVote.report (Vote_...:273)
        assert ex["final_report"] == vote.STRAINVOTE_HEADER
        assert ex["strain_vote"] == vote.STRAINVOTE_HEADER
    cls = {int(k): v for k, v in g["A_single"]["cls_dict"].items()}
    vote.generate_single_report(cls, str(tmp_path))
    assert (tmp_path / "final_report.txt").read_text() == g["A_single"]["final_report"]
    cls = {int(k): v for k, v in g["A_l2"]["cls_dict"].items()}
    (tmp_path / "C1").mkdir()
    (tmp_path / "C1" / "StrainVote.report").write_text(g["A_l2"]["strain_vote"])
    vote.merge_res(str(tmp_path), cls)
    assert (tmp_path / "final_report.txt").read_text() == g["A_l2"]["final_report"]


def test_tree_image_cache_roundtrip(l1_dbs, tmp_path, monkeypatch):
    """Text parse of kmer.fa + kmers/<id> and its binary cache give identical arrays."""
    from strainscan_amd import db as sdb
    monkeypatch.setenv("SS_IMAGE_CACHE", str(tmp_path / "cache"))
    tdb = os.path.join(l1_dbs["A"]["db_dir"], "Tree_database")
    k1, f1, ids1, l1 = sdb.load_tree_text(tdb)
    k2, f2, ids2, l2 = sdb.load_tree_text(tdb)           # second call: joins the first call's writer, reads the cache
    assert not sdb._CACHE_WRITERS
    assert [f[:5] for f in os.listdir(tmp_path / "cache")] == ["tree_"]      # one image, no temp file left
    assert isinstance(k2, np.memmap) or isinstance(getattr(k2, "base", None), (np.memmap, np.ndarray))
    assert np.array_equal(k1, k2) and np.array_equal(f1, f2) and ids1 == ids2
    assert all(np.array_equal(a, b) for a, b in zip(l1, l2))
    info = l1_dbs["A"]
    assert ids1 == info["tree"].ids
    for i, rows in zip(ids1, l1):
        assert rows.tolist() == info["row_of_node"][i]
    monkeypatch.setenv("SS_IMAGE_CACHE", "off")
    k3, _, _, _ = sdb.load_tree_text(tdb)
    assert np.array_equal(k1, k3)


def test_tree_cache_writers_do_not_collide(l1_dbs, tmp_path, monkeypatch):
    """`-b 1` on a new database loads the tree twice in a row (identify_low_depth.py:119 then identify.py:402).  The
    cache is written on a worker thread: the second load must wait for the first one's writer instead of parsing
    again and starting a second writer on the same file; every writer has a temp file of its own; what ends up in
    the cache maps back to exactly the arrays of a fresh parse."""
    import threading
    from strainscan_amd import db as sdb
    cache = tmp_path / "cache"
    monkeypatch.setenv("SS_IMAGE_CACHE", str(cache))
    tdb = os.path.join(l1_dbs["A"]["db_dir"], "Tree_database")
    gate = threading.Event()
    real_write = sdb._write_tree_cache
    n_writes = []

    def slow_write(path, t):
        n_writes.append(path)
        gate.wait(10)                     # the writer is still busy when the second load_tree starts
        real_write(path, t)

    monkeypatch.setattr(sdb, "_write_tree_cache", slow_write)
    t1 = sdb.load_tree(tdb)
    assert len(sdb._CACHE_WRITERS) == 1
    threading.Timer(0.3, gate.set).start()
    t2 = sdb.load_tree(tdb)               # joins the writer, then maps the finished file
    assert len(n_writes) == 1 and not sdb._CACHE_WRITERS
    files = os.listdir(cache)
    assert len(files) == 1 and files[0].startswith("tree_") and files[0].endswith(".bin")
    fresh = sdb._read_tree_cache(str(cache / files[0]))
    monkeypatch.setenv("SS_IMAGE_CACHE", "off")
    t3 = sdb.load_tree(tdb)
    for t in (t1, t2, fresh):
        assert list(t.ids) == list(t3.ids)
        for name in ("keys", "flags", "rows", "offs", "urows", "uoffs"):
            assert np.array_equal(getattr(t, name), getattr(t3, name)), name
    # two writers of the same image at once (two processes in real life): both finish, one complete file
    monkeypatch.setattr(sdb, "_write_tree_cache", real_write)
    ths = [threading.Thread(target=real_write, args=(str(cache / files[0]), t3)) for _ in range(4)]
    [th.start() for th in ths]
    [th.join() for th in ths]
    assert os.listdir(cache) == files
    again = sdb._read_tree_cache(str(cache / files[0]))
    assert np.array_equal(again.keys, t3.keys) and np.array_equal(again.rows, t3.rows)


def test_node_lists_native_parser_equals_numpy(tmp_path, monkeypatch):
    """kmers/<id> through ss_node_lists_parse (host threads) and through the numpy loop: same rows, offsets and de-duplicated
    forms -- ascending lists, a list with repeats and disorder, an empty file, tabs and trailing blanks, a second line that
    must be ignored (identify.py:116-118 reads one line); a row outside kmer.fa, a negative number or a word are left to
    the numpy loop, which raises."""
    from strainscan_amd import db as sdb
    kd = tmp_path / "kmers"
    kd.mkdir()
    rs = np.random.RandomState(3)
    lists = {1: np.sort(rs.choice(5000, 700, replace=False)), 2: rs.randint(0, 5000, 300), 3: np.zeros(0, np.int64), 7: np.arange(10, 20),
             12: np.array([4999, 0, 4999])}
    for i, r in lists.items():
        sep = "\t" if i == 7 else " "
        (kd / str(i)).write_text(sep.join(map(str, r.tolist())) + (" " if r.size else "") + ("\n999999 x\n" if i == 2 else ""))
    ids = sorted(lists)
    a = sdb._node_lists_native(str(kd), ids, 5000)
    assert a is not None
    monkeypatch.setattr(sdb, "_node_lists_native", lambda *args: None)
    b = sdb._node_lists(str(tmp_path), ids, 5000)
    for x, y in zip(a, b):
        assert np.array_equal(x, y) and x.dtype == y.dtype
    assert a[0][a[1][1]:a[1][2]].tolist() == lists[2].tolist()
    monkeypatch.undo()
    for bad in ("5000", "-3", "12 abc"):
        (kd / "7").write_text("1 2 " + bad + " ")
        assert sdb._node_lists_native(str(kd), ids, 5000) is None
        with pytest.raises(ValueError):
            sdb._node_lists(str(tmp_path), ids, 5000)


def test_plain_pickles_admit_no_globals(tmp_path):
    """id2strain_re.pkl is a list of names (Recls_withR_new.py:114-115): read without admitting any global."""
    import pickle
    from strainscan_amd.tree import load_plain_pkl
    p = tmp_path / "ok.pkl"
    with open(p, "wb") as f:
        pickle.dump(["GCF_1", "GCF_2", {"a": 1, "b": [2.5, None]}], f, pickle.HIGHEST_PROTOCOL)
    assert load_plain_pkl(str(p)) == ["GCF_1", "GCF_2", {"a": 1, "b": [2.5, None]}]
    q = tmp_path / "bad.pkl"
    with open(q, "wb") as f:
        pickle.dump([os.getcwd, "x"], f)
    with pytest.raises(pickle.UnpicklingError):
        load_plain_pkl(str(q))


def test_revcomp_host_vs_golden(golden_dir):
    """library/seqpy.c:5-36 (IUPAC table, case kept, other bytes unchanged) -- host forms of the product:
    ss_revcomp through ctypes and the seqpy drop-in module.  Golden pairs come from the reference's own
    seqpy.c compiled in the build container (tests/golden/make_golden.py)."""
    from strainscan_amd import _lib, seqpy
    with open(os.path.join(golden_dir, "revcomp.json")) as f:
        pairs = json.load(f)
    assert len(pairs) >= 8
    for s, want in pairs:
        assert seqpy.revcomp(s) == want
        assert _lib.revcomp(s.encode()) == want.encode()
        out = ctypes.create_string_buffer(max(1, len(s)))
        assert _lib.lib().ss_revcomp(s.encode(), out, len(s)) == 0
        assert out.raw[:len(s)] == want.encode()


def test_bench_gpus_argument_handling():
    """`python bench.py --gpus 2` without a launcher spawns two fresh rank processes (RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 set) before anything touches torch, and relays rank 0's line; under a launcher
    a --gpus that disagrees with WORLD_SIZE is an error.  SS_BENCH_WORKER_STUB stops the worker before
    the GPU work."""
    import subprocess
    import sys
    bench = os.path.join(REPO, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SS_BENCH_WORKER_STUB"] = "1"
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "3"], env=env, capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rank"] == 0 and d["master"] == "127.0.0.1" and d["argv"] == ["--gpus", "2", "--steps", "3"]
    # one process, no launcher: world 1
    r = subprocess.run([sys.executable, bench], env=env, capture_output=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.decode())["n_gpus"] == 1
    # launcher started 2 ranks but --gpus says 4: refuse
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, bench, "--gpus", "4"], env=env2, capture_output=True, timeout=120)
    assert r.returncode != 0 and b"--gpus 4" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env2, capture_output=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.decode())["n_gpus"] == 2


@pytest.mark.parametrize("shape", ["sampled", "contiguous"])
def test_bench_database_generator(shape):
    """bench.make_db on the CPU at a small size: both key conventions name the same k-mers (device: A0 C1 T2 G3, first
    base in the low bits; oracle: A0 C1 G2 T3, first base in the high bits), every row belongs to exactly one node list,
    sampled node sets are a sparse subset of their stretch with rows scattered over the file, and reads cut from the
    path of a leaf hit that path's rows (oracle counter)."""
    import torch
    import bench
    from oracle import oracle as orc
    dev = torch.device("cpu")
    spec = bench.make_db(torch, dev, 9, seed=3, lo_sites=300, hi_sites=900, shape=shape, hit_frac=0.05)
    keys, okeys = spec["keys"], spec["okeys"]
    assert keys.size == okeys.size == int(spec["row_off"][-1]) and spec["n_nodes"] == 17
    letters = "ACTG"
    for i in np.random.RandomState(1).choice(keys.size, size=200, replace=False):
        txt = "".join(letters[(int(keys[i]) >> (2 * j)) & 3] for j in range(31))
        assert orc.encode_kmer(txt) == int(okeys[i])
    assert sorted(spec["rows"].tolist()) == list(range(keys.size))
    off = spec["row_off"].astype(np.int64)
    per_node = np.diff(off)
    if shape == "sampled":
        stretch = np.diff(spec["seq_off"]) - 30
        assert np.all(per_node < 0.45 * 2 * stretch) and np.all(per_node > 0)          # a sparse sample of (site, orientation)
        assert not np.array_equal(spec["rows"], np.arange(keys.size))                # scattered over kmer.fa
    else:
        assert np.array_equal(per_node, 2 * spec["sites"])
    reads = bench.make_reads(torch, dev, spec, 3000, seed=5, hit_frac=0.05)
    counts = orc.count_flat(okeys, 31, reads.numpy(), 2)
    hit_nodes = {j for j in range(17) if counts[spec["rows"][off[j]:off[j + 1]]].sum() > 0}
    assert 0 in hit_nodes and 3 <= len(hit_nodes) <= 12                               # the root and the three leaves' paths
    frac = counts.sum() / (3000 * 120)
    assert 0.01 < frac < 0.08


def test_documents_name_only_what_the_header_declares():
    """Every `ss_*` entry point that INTEGRATION.md, DESIGN.md or README.md name is declared in include/strainscan_hip.h
    (file stems like `ss_mini.hip` and prefixes like `ss_nodes_*` aside)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "strainscan_hip.h")).read()
    known = set(re.findall(r"\b(ss_[a-z0-9_]+)\b", hdr))
    stems = {os.path.splitext(f)[0] for f in os.listdir(os.path.join(root, "strainscan_amd", "csrc"))}
    for doc in ("INTEGRATION.md", "DESIGN.md", "README.md"):
        text = open(os.path.join(root, doc)).read()
        names = set(re.findall(r"`(ss_[a-z0-9_]+)", text))
        unknown = sorted(n for n in names if n not in known and n not in stems and not n.endswith("_"))
        assert not unknown, (doc, unknown)


def test_documents_cite_only_tests_and_files_that_exist():
    """DESIGN.md's row-by-row map of SURVEY.md 8 (and every other `test_...` the documents name) points at tests of this suite;
    the `profiles/...` and `scripts/...` files DESIGN.md, README.md and INTEGRATION.md name are in the tree."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    have = set()
    for f in glob.glob(os.path.join(root, "tests", "test_*.py")):
        have |= set(re.findall(r"^def (test_[a-z0-9_]+)", open(f).read(), re.M))
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md")):
        text = open(os.path.join(root, doc)).read()
        cited = set(re.findall(r"`(?:tests/test_[a-z_]+\.py::)?(test_[a-z0-9_]+)`", text))
        missing = sorted(t for t in cited if t not in have and not os.path.exists(os.path.join(root, "tests", t + ".py")))
        assert not missing, (doc, missing)
        for path in set(re.findall(r"`((?:profiles|scripts|tests|oracle|include|docs)/[A-Za-z0-9_./-]+\.[a-z]{1,4})`", text)):
            if any(ch in path for ch in "*{<") or "_ref/" in path:
                continue
            assert os.path.exists(os.path.join(root, path)), (doc, path)


def test_cache_tags_and_npz_reader(tmp_path):
    """db.cache_tag: 128 bits, deterministic, different for inputs that differ in one character or only in length;
    _load_npz_csr: the arrays of scipy's save_npz file (compressed or not) without building the matrix, anything but CSR
    through scipy."""
    import scipy.sparse as sp
    from strainscan_amd.db import cache_tag
    from strainscan_amd.identify_strains_L2_Enet_Pscan_new_sp import _load_npz_csr
    a = cache_tag("/data/db/Tree_database|123456|1700000000000000000|31|1")
    assert a == cache_tag("/data/db/Tree_database|123456|1700000000000000000|31|1") and len(a) == 32 and int(a, 16) >= 0
    seen = {a}
    for other in ("/data/db/Tree_database|123456|1700000000000000000|31|2", "/data/db/Tree_databasf|123456|1700000000000000000|31|1",
                  "/data/db/Tree_database|123456|1700000000000000000|31|1 ", "", "a", "a\0", "ab"):
        t = cache_tag(other)
        assert t not in seen, other
        seen.add(t)
    rs = np.random.RandomState(3)
    m = sp.random(500, 37, density=0.1, format="csr", random_state=rs, dtype=np.float64)
    m.data[:] = 1
    m = m.astype(np.int8)
    for compressed in (True, False):
        p = str(tmp_path / ("m%d.npz" % compressed))
        sp.save_npz(p, m, compressed=compressed)
        c = _load_npz_csr(p)
        assert c.shape == (500, 37) and c.nnz == m.nnz
        assert np.array_equal(c.indptr, m.indptr) and np.array_equal(c.indices, m.indices) and np.array_equal(c.data, m.data)
    p = str(tmp_path / "csc.npz")
    sp.save_npz(p, m.tocsc())
    c = _load_npz_csr(p)
    assert np.array_equal(c.indptr, m.indptr) and np.array_equal(c.indices, m.indices)


def test_every_environment_variable_is_documented():
    """INTEGRATION.md section F lists every SS_* / STRAINSCAN_* variable the native sources and the package read, and nothing
    that is not read any more: a new knob without a line there -- default, reader, what it can change -- fails here."""
    import re
    src = {}
    for d, pat in ((os.path.join(REPO, "strainscan_amd", "csrc"), r'getenv\("((?:SS|STRAINSCAN)_[A-Z0-9_]+)"\)'),
                   (os.path.join(REPO, "strainscan_amd"), r'["\']((?:SS|STRAINSCAN)_[A-Z0-9_]+)["\']')):
        for fn in sorted(os.listdir(d)):
            if not fn.endswith((".hip", ".h", ".py")):
                continue
            text = open(os.path.join(d, fn)).read()
            for m in re.finditer(pat, text):
                src.setdefault(m.group(1), set()).add(fn)
    status_codes = {n for n in src if re.fullmatch(r"SS_(OK|E[A-Z]+)", n)}          # (names of the int status codes in _lib.py)
    read = set(src) - status_codes - {"SS_BENCH_SHARE_GPU"}
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    sec = doc[doc.index("## F. Environment variables"):]
    table = "\n".join(ln for ln in sec.splitlines() if ln.startswith("| `"))
    listed = set(re.findall(r"`((?:SS|STRAINSCAN)_[A-Z0-9_]+)`", table))
    assert read - listed == set(), "read but not documented: %s" % sorted((v, sorted(src[v])) for v in read - listed)
    assert listed - read == set(), "documented but no longer read: %s" % sorted(listed - read)
    assert len(read) <= 45

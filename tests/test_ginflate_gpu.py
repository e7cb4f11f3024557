"""The device gzip inflater (strainscan_amd/csrc/ss_ginflate.hip) against Python's gzip module.

The reference reads .gz samples through gzip.open (library/identify.py:100-108 via jellyfish's input pipe); here the
member is inflated on the GPU when SS_GZ_GPU=1.  Contract under test: whatever ss_gz_inflate_gpu returns with SS_OK
is byte-identical to gzip.decompress; anything it cannot do (several members, damage, too few sync points, data that
expands beyond its symbol budget) is DECLINED with SS_ERANGE, never answered wrongly; and the scan of .gz inputs
with SS_GZ_GPU=1 counts exactly like the plain text."""
import ctypes as C
import gzip
import time
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SS_OK, SS_ERANGE = 0, -34


@pytest.fixture(scope="module")
def L():
    from strainscan_amd import _lib
    _lib.require_gpu()
    return _lib


def _fastq(n, seed, read_len=150):
    rs = np.random.RandomState(seed)
    lut = np.frombuffer(b"ACGT", np.uint8)
    q = np.frombuffer(b"FFFFFFFF:FFF,FF#", np.uint8)
    bases = lut[rs.randint(0, 4, (n, read_len))]
    quals = q[rs.randint(0, 16, (n, read_len))]
    return b"".join(b"@SRR1234567.%d %d/1\n" % (i, i) + bases[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n" for i in range(n))


def _gpu_inflate(L, path):
    t, n = C.c_void_p(), C.c_uint64()
    rc = L.lib().ss_gz_inflate_gpu(os.fsencode(str(path)), C.byref(t), C.byref(n))
    if rc != SS_OK:
        return rc, None
    s = C.string_at(t, n.value)
    L.lib().ss_gz_free(t)
    return rc, s


def _counters(L):
    a, b = C.c_uint64(), C.c_uint64()
    L.check(L.lib().ss_gz_gpu_counters(C.byref(a), C.byref(b)), "ss_gz_gpu_counters")
    return int(a.value), int(b.value)


@pytest.fixture(scope="module")
def fastq_text():
    return _fastq(40000, 1)


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("chunk", [None, "16384", "65536"])
def test_fastq_levels_and_chunk_sizes(L, tmp_path, monkeypatch, fastq_text, level, chunk):
    """FASTQ text at gzip levels 1, 6, 9 (different block sizes and match statistics), at three chunk sizes (the
    sync search lands on different blocks): handled, and equal to the text."""
    if chunk:
        monkeypatch.setenv("SS_GZ_CHUNK", chunk)
    else:
        monkeypatch.delenv("SS_GZ_CHUNK", raising=False)
    p = tmp_path / "a.fq.gz"
    p.write_bytes(gzip.compress(fastq_text, level))
    h0, _ = _counters(L)
    rc, got = _gpu_inflate(L, p)
    assert rc == SS_OK
    assert got == fastq_text
    assert _counters(L)[0] == h0 + 1


def test_stored_fixed_and_tiny(L, tmp_path):
    """Random bytes (zlib emits stored blocks), a short text (one fixed-Huffman block, a single chunk), text written
    with Z_FIXED (fixed codes throughout) and with Z_HUFFMAN_ONLY (no matches): equal when handled."""
    rnd = np.random.RandomState(3).randint(0, 256, 3_000_000).astype(np.uint8).tobytes()
    tiny = b"hello world\n" * 10
    txt = _fastq(6000, 5)

    def deflate(data, strategy):
        c = zlib.compressobj(6, zlib.DEFLATED, 31, 8, strategy)
        return c.compress(data) + c.flush()

    cases = dict(random=(gzip.compress(rnd, 6), rnd), tiny=(gzip.compress(tiny, 6), tiny),
                 fixed=(deflate(txt, zlib.Z_FIXED), txt), huffman_only=(deflate(txt, zlib.Z_HUFFMAN_ONLY), txt),
                 rle=(deflate(txt, zlib.Z_RLE), txt), level0=(gzip.compress(txt, 0), txt))
    handled = 0
    for name, (gz, want) in cases.items():
        assert gzip.decompress(gz) == want
        p = tmp_path / (name + ".gz")
        p.write_bytes(gz)
        rc, got = _gpu_inflate(L, p)
        assert rc in (SS_OK, SS_ERANGE), name
        if rc == SS_OK:
            handled += 1
            assert got == want, name
    assert handled >= 4


def test_scratch_is_kept_and_can_be_released(L, tmp_path, fastq_text):
    """The scratch arena of a call is reused by the next one; ss_gz_gpu_release hands it back; the call after that builds
    a new one.  Same text every time."""
    p = tmp_path / "a.fq.gz"
    p.write_bytes(gzip.compress(fastq_text, 6))
    for step in range(4):
        rc, got = _gpu_inflate(L, p)
        assert rc == SS_OK and got == fastq_text, step
        if step == 1:
            assert L.lib().ss_gz_gpu_release() == 0
    assert L.lib().ss_gz_gpu_release() == 0


def test_header_fields(L, tmp_path, fastq_text):
    """FNAME / FCOMMENT / FEXTRA / FHCRC in the member header are skipped (RFC 1952 2.3)."""
    raw = gzip.compress(fastq_text[: 4 << 20], 6)
    body = raw[10:]
    name, comment, extra = b"reads_1.fq\0", b"made by a test\0", b"\x06\x00AB\x02\x00xy"
    flg = 4 | 8 | 16
    hdr = raw[:3] + bytes([flg]) + raw[4:10] + extra + name + comment
    p = tmp_path / "h.gz"
    p.write_bytes(hdr + body)
    assert gzip.decompress(hdr + body) == fastq_text[: 4 << 20]
    rc, got = _gpu_inflate(L, p)
    assert rc == SS_OK and got == fastq_text[: 4 << 20]
    hdr2 = raw[:3] + bytes([flg | 2]) + raw[4:10] + extra + name + comment
    hdr2 += (zlib.crc32(hdr2) & 0xFFFF).to_bytes(2, "little")
    p.write_bytes(hdr2 + body)
    assert gzip.decompress(hdr2 + body) == fastq_text[: 4 << 20]
    rc, got = _gpu_inflate(L, p)
    assert rc == SS_OK and got == fastq_text[: 4 << 20]


def test_declines_instead_of_answering_wrongly(L, tmp_path, fastq_text):
    """A bgzip-style file of many small members, a truncated file, a flipped byte in the deflate data, a wrong CRC-32 (of
    the last and of the first of two members), a wrong ISIZE, zero padding behind the member, text that expands 1000:1
    (beyond the symbol budget), not gzip at all: SS_ERANGE -- or, for a flipped byte that happens to survive, the exact
    text -- never different bytes."""
    txt = fastq_text[: 6 << 20]
    gz = gzip.compress(txt, 6)
    half = len(txt) // 2
    mid = len(gz) // 2
    bgz = b"".join(gzip.compress(txt[i:i + 65000], 6) for i in range(0, len(txt), 65000))
    m1 = gzip.compress(txt[:half], 6)
    cases = dict(
        many_members=bgz,
        first_member_bad_crc=m1[:-8] + bytes([m1[-8] ^ 1]) + m1[-7:] + gzip.compress(txt[half:], 6),
        zero_padding=gz + b"\0" * 512,
        truncated=gz[: len(gz) - 4000],
        flipped=gz[:mid] + bytes([gz[mid] ^ 0x10]) + gz[mid + 1:],
        bad_crc=gz[:-8] + bytes([gz[-8] ^ 1]) + gz[-7:],
        bad_isize=gz[:-4] + bytes([gz[-4] ^ 1]) + gz[-3:],
        expands=gzip.compress(b"A" * (64 << 20), 6),
        not_gzip=txt[: 1 << 20],
    )
    _, d0 = _counters(L)
    declined = 0
    for name, data in cases.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(data)
        rc, got = _gpu_inflate(L, p)
        if name in ("flipped", "many_members") and rc == SS_OK:      # (the members may just fit the chunk list)
            assert got == txt
            continue
        assert rc == SS_ERANGE, name
        declined += 1
    assert _counters(L)[1] == d0 + declined


def test_streams_without_dynamic_blocks_go_to_the_host(L, tmp_path, fastq_text):
    """The device inflater enters a stream at the starts of DYNAMIC blocks (and at places inside them).  A member made of stored
    blocks only (level 0) has none: one wave would copy all of it alone -- a stretch of more than 8 MB without an entry is declined
    (SS_ERANGE) at the entry list and the host inflaters take the file.  A member of fixed-Huffman blocks only (Z_FIXED) has no
    dynamic block either, but it is entered INSIDE its first block (all its blocks share the fixed tables, so the places found
    there are items' first bits): either the exact text or a decline, and quickly."""
    import zlib
    txt = fastq_text * ((48 << 20) // len(fastq_text) + 1)      # whole records, ~50 MB (the window is 32 KB: a repeat 12 MB back is no match)

    def member(data, level, strategy):
        c = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
        return c.compress(data) + c.flush()

    stored = member(txt, 0, zlib.Z_DEFAULT_STRATEGY)
    fixed = member(txt, 6, zlib.Z_FIXED)
    assert len(stored) > (40 << 20) and len(fixed) > (9 << 20)
    _, d0 = _counters(L)
    for name, data in (("stored", stored), ("fixed", fixed)):
        p = tmp_path / (name + ".gz")
        p.write_bytes(data)
        assert gzip.decompress(data) == txt
        t0 = time.perf_counter()
        rc, got = _gpu_inflate(L, p)
        took = time.perf_counter() - t0
        if name == "stored":
            assert rc == SS_ERANGE and took < 5.0, (name, rc, took)      # declined at the entry list, not after a wave has worked through it
        else:
            assert (rc == SS_OK and got == txt) or rc == SS_ERANGE, (name, rc)
            assert took < 20.0, (name, took)
    assert _counters(L)[1] >= d0 + 1
    small = member(txt[: 3 << 20], 6, zlib.Z_FIXED)
    p = tmp_path / "fixed_small.gz"
    p.write_bytes(small)
    rc, got = _gpu_inflate(L, p)
    assert rc == SS_ERANGE or got == txt[: 3 << 20]
    # the resident read set of such a file: the host inflaters' text, the same records
    rs_h = L.ReadSet([str(tmp_path / "fixed.gz")])
    assert rs_h.info()["n_records"] == txt.count(b"\n") // 4
    rs_h.close()


def _kmer_fa(seed, n_rows):
    rs = np.random.RandomState(seed)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rows = lut[rs.randint(0, 4, (n_rows, 31))]
    return rows, b"".join(b">1\n" + r.tobytes() + b"\n" for r in rows)


def test_scan_of_gz_inputs_through_the_device_inflater(L, tmp_path, monkeypatch):
    """ss_scan_files and ss_reads_load inflate .gz inputs on the device (the counters move) -- one member, a pair of
    files, a file of two members -- and the row counts equal those of the plain text; SS_GZ_GPU=0: the host inflaters."""
    rows, kfa = _kmer_fa(11, 50000)
    rs = np.random.RandomState(12)
    lut = np.frombuffer(b"ACGT", np.uint8)
    n = 60000
    reads = lut[rs.randint(0, 4, (n, 150))]
    for i in range(0, n, 3):                                  # every third read carries a database k-mer
        r = rows[rs.randint(0, rows.shape[0])]
        o = rs.randint(0, 150 - 31)
        reads[i, o:o + 31] = r
    fq = b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + b"F" * 150 + b"\n" for i in range(n))
    assert len(fq) > (16 << 20)
    half = fq.index(b"\n@r%d\n" % (n // 2)) + 1
    plain = tmp_path / "a.fq"
    plain.write_bytes(fq)
    one = tmp_path / "one.fq.gz"
    one.write_bytes(gzip.compress(fq, 6))
    p1, p2 = tmp_path / "p_1.fq.gz", tmp_path / "p_2.fq.gz"
    p1.write_bytes(gzip.compress(fq[:half], 1))
    p2.write_bytes(gzip.compress(fq[half:], 9))
    multi = tmp_path / "multi.fq.gz"
    multi.write_bytes(gzip.compress(fq[:half], 6) + gzip.compress(fq[half:], 6))
    db = L.KmerDB.from_text(kfa, 31, True)
    db.scan_files([str(plain)])
    want = db.counts_rows().copy()
    assert want.sum() >= n // 3
    monkeypatch.setenv("SS_GZ_GPU", "1")
    for paths, on_device in (([str(one)], 1), ([str(p1), str(p2)], 2), ([str(multi)], 1)):
        h0, d0 = _counters(L)
        db.reset()
        nrec, _ = db.scan_files(paths)
        h1, d1 = _counters(L)
        assert nrec == n
        assert np.array_equal(db.counts_rows(), want), paths
        assert h1 - h0 == on_device and d1 - d0 == len(paths) - on_device, paths
        rset = L.ReadSet(paths, 0, 1)
        assert rset.info()["n_records"] == n
        assert _counters(L)[0] - h1 == on_device
        db.reset()
        rset.scan_into(db)
        assert np.array_equal(db.counts_rows(), want), paths
        rset.close()
    monkeypatch.setenv("SS_GZ_GPU", "0")
    h0, d0 = _counters(L)
    db.reset()
    db.scan_files([str(one)])
    assert _counters(L) == (h0, d0)
    assert np.array_equal(db.counts_rows(), want)


def _records_of(read_set):
    """The resident flat blocks as a sorted list of records (block order and padding newlines do not matter)."""
    return sorted(r for r in read_set.read_back().split(b"\n") if r)


@pytest.mark.parametrize("case", ["strict", "no_final_newline", "crlf", "empty_sequence", "at_in_quality", "fasta", "two_line_sequence",
                                  "blank_lines", "short_quality", "plus_line_missing", "tiny"])
def test_device_fastq_extraction_equals_the_host_grammar(L, tmp_path, monkeypatch, case):
    """ss_fastq_dev.hip takes strict four-line FASTQ only and must then deliver what ss_fastx_to_flat's grammar (the
    reference's zcat | jellyfish --if reads the same sequence lines, identify.py:81-84) delivers; every other shape
    (FASTA, wrapped sequences, blank lines, a quality line of another length, ...) goes to that grammar on the host.
    Records and record count of the resident read set: SS_GZ_GPU=1 == SS_GZ_GPU=0 == ss_fastx_to_flat."""
    rs = np.random.RandomState(99)
    lut = np.frombuffer(b"ACGTN", np.uint8)
    n = 60 if case == "tiny" else 30000
    seqs = [lut[rs.randint(0, 5 if i % 50 == 0 else 4, rs.randint(40, 260))].tobytes() for i in range(n)]
    nl = b"\r\n" if case == "crlf" else b"\n"
    recs = []
    for i, sq in enumerate(seqs):
        q = bytes(33 + (i * 7 + j) % 41 for j in range(len(sq)))
        if case == "empty_sequence" and i % 97 == 5:
            sq, q = b"", b""
        if case == "at_in_quality":
            q = b"@" + q[1:] if q else q
        if case == "fasta":
            recs.append(b">r%d" % i + nl + sq + nl)
            continue
        if case == "two_line_sequence" and i % 3 == 0:
            h = len(sq) // 2
            recs.append(b"@r%d" % i + nl + sq[:h] + nl + sq[h:] + nl + b"+" + nl + q + nl)
            continue
        if case == "short_quality" and i == n // 2:
            q = q[: len(q) // 2] + nl + q[len(q) // 2:]                 # quality over two lines (the general grammar adds them up)
        plus = b"" if (case == "plus_line_missing" and i == n - 3) else b"+" + nl
        recs.append(b"@r%d some text" % i + nl + sq + nl + plus + q + nl + (nl if case == "blank_lines" and i % 11 == 0 else b""))
    text = b"".join(recs)
    if case == "no_final_newline":
        text = text[:-1]
    want_flat, want_n = L.fastx_to_flat(text)
    want = sorted(r for r in want_flat.split(b"\n") if r)
    p = tmp_path / "s.fq.gz"
    p.write_bytes(gzip.compress(text, 6))
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SS_GZ_GPU", mode)
        h0, d0 = _counters(L)
        rset = L.ReadSet([str(p)], 0, 1)
        got[mode] = (_records_of(rset), rset.info()["n_records"])
        rset.close()
        if mode == "0":
            assert _counters(L) == (h0, d0)
    assert got["1"][0] == got["0"][0] == want, case
    assert got["1"][1] == got["0"][1] == want_n, case


def test_a_wrong_entry_point_is_dropped_and_the_chunks_inflated_again(L, tmp_path, monkeypatch, fastq_text):
    """The sync search accepts a position where a valid dynamic header parses and a few hundred symbols decode; about
    one candidate in a million that passes lies INSIDE a block (seen on a 264 MB file).  The chunk in front of such an
    entry ends its block behind it: it goes on to the entry after that and the wrong one's chunk is dropped (or, when the
    symbol region has no room, the two chunks are merged and inflated again).  ss_test_hook(1, n) plants a wrong entry:
    same text, still verified by CRC-32 and ISIZE."""
    p = tmp_path / "a.fq.gz"
    p.write_bytes(gzip.compress(fastq_text, 6))
    for host_only in (False, True):          # the chunk runs over the wrong entry itself / the host merges the two chunks
        if host_only:
            monkeypatch.setenv("SS_GZ_NO_RUNOVER", "1")
        try:
            for chunk in (3, 17, 40):
                L.check(L.lib().ss_test_hook(1, chunk), "ss_test_hook")
                rc, got = _gpu_inflate(L, p)
                assert rc == SS_OK, (chunk, host_only)
                assert got == fastq_text, (chunk, host_only)
        finally:
            L.lib().ss_test_hook(1, 0)


@pytest.mark.parametrize("layout", ["two", "lanes", "small_middle", "small_last", "levels"])
def test_files_of_several_members(L, tmp_path, layout, fastq_text):
    """Lanes joined with `cat a.gz b.gz` are one file of several gzip members.  The chunk that meets a member's final
    block finds trailer and next header behind it; the next member's first block becomes a chunk of its own (nothing in
    front of it), every member is checked against ITS CRC-32 and ISIZE.  Equal to gzip.decompress."""
    t = fastq_text
    n = len(t)
    cuts = dict(two=[n // 2], lanes=[n // 4, n // 2, 3 * n // 4], small_middle=[n // 3, n // 3 + 5000], small_last=[n - 3000],
                levels=[n // 5, 2 * n // 5, 3 * n // 5, 4 * n // 5])[layout]
    parts = [t[a:b] for a, b in zip([0] + cuts, cuts + [n])]
    levels = [6, 1, 9, 4, 6] if layout == "levels" else [6] * len(parts)
    gz = b"".join(gzip.compress(part, lv) for part, lv in zip(parts, levels))
    assert gzip.decompress(gz) == t
    p = tmp_path / "m.fq.gz"
    p.write_bytes(gz)
    rc, got = _gpu_inflate(L, p)
    assert rc == SS_OK, layout
    assert got == t, layout


def _bgzf(data, block=60000, level=6):
    """A bgzip file: members of at most 64 KB, each with the "BC" extra field (its size - 1), and the empty EOF member."""
    import struct
    out = []
    for a in list(range(0, len(data), block)) + [None]:
        piece = b"" if a is None else data[a:a + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(piece) + co.flush()
        bsize = 12 + 6 + len(body) + 8
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + body +
                   struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece) & 0xFFFFFFFF))
    return b"".join(out)


def test_bgzip_file(L, tmp_path, fastq_text):
    """bgzip (BGZF) writes thousands of small members with their size in the header: the members are the chunks (no sync
    search, nothing unknown in front of any), each checked against its own CRC-32 and ISIZE.  A damaged member: declined."""
    gz = _bgzf(fastq_text)
    assert gzip.decompress(gz) == fastq_text
    p = tmp_path / "b.fq.gz"
    p.write_bytes(gz)
    rc, got = _gpu_inflate(L, p)
    assert rc == SS_OK and got == fastq_text
    mid = len(gz) // 2
    p.write_bytes(gz[:mid] + bytes([gz[mid] ^ 0x21]) + gz[mid + 1:])
    rc, got = _gpu_inflate(L, p)
    assert rc == SS_ERANGE or got == fastq_text


@pytest.mark.parametrize("seg_kb", ["256", "1024", "3000"])
def test_segments(L, tmp_path, monkeypatch, fastq_text, seg_kb):
    """The chunks are inflated segment by segment (128 MB of deflate data each; here SS_GZ_SEG_KB forces tens of them):
    the window in front of a segment's first chunk is the end of the text so far; wrong entries at segment edges, members
    across segments and the scratch kept between calls must not show.  One member, four members, bgzip."""
    monkeypatch.setenv("SS_GZ_SEG_KB", seg_kb)
    n = len(fastq_text)
    one = gzip.compress(fastq_text, 6)
    four = b"".join(gzip.compress(fastq_text[a:b], lv) for a, b, lv in ((0, n // 3, 6), (n // 3, n // 3 + 70000, 1), (n // 3 + 70000, n - 10, 9), (n - 10, n, 6)))
    for name, gz in (("one", one), ("four", four), ("bgzf", _bgzf(fastq_text))):
        p = tmp_path / (name + ".gz")
        p.write_bytes(gz)
        try:
            for inject in (0, 5, 33):
                L.check(L.lib().ss_test_hook(1, inject), "ss_test_hook")
                rc, got = _gpu_inflate(L, p)
                assert rc == SS_OK, (name, inject)
                assert got == fastq_text, (name, inject)
        finally:
            L.lib().ss_test_hook(1, 0)


def test_warm_up_then_pinned_upload(L, tmp_path, monkeypatch):
    """ss_gz_warm_up makes the pinned upload buffers ahead of time (the CLI's warm-up thread); a file of 32 MB or more then
    travels through them (eight pread threads with a stream each, blocks of n / 16, the files of a call one after the other on
    the link) instead of being copied out of the mapping: the same records
    (SS_READS_ORDER=file: as they stand in the file)."""
    monkeypatch.setenv("SS_READS_ORDER", "file")
    assert L.lib().ss_gz_warm_up(2) == SS_OK
    assert L.lib().ss_gz_warm_up(0) == SS_OK
    t = _fastq(330000, 9)                      # ~100 MB of text, ~40 MB of .gz
    p = tmp_path / "w.fq.gz"
    p.write_bytes(gzip.compress(t, 1))
    assert p.stat().st_size >= (32 << 20)
    h0, _ = _counters(L)
    rs = L.ReadSet([str(p)])
    got = rs.read_back()
    rs.close()
    assert _counters(L)[0] == h0 + 1
    want, n_rec = L.fastx_to_flat(t)
    assert n_rec == 330000
    assert [r for r in got.split(b"\n") if r] == [r for r in want.split(b"\n") if r]
    # ... and as the two mates of a pair: both files in flight, one after the other on the link, a set of buffers each
    p2 = tmp_path / "w2.fq.gz"
    p2.write_bytes(gzip.compress(t, 1))
    assert p2.stat().st_size >= (32 << 20)
    rs = L.ReadSet([str(p), str(p2)])
    assert rs.info()["n_records"] == 2 * 330000
    got2 = rs.read_back()
    rs.close()
    assert _counters(L)[0] == h0 + 3
    assert sorted(r for r in got2.split(b"\n") if r) == sorted([r for r in want.split(b"\n") if r] * 2)

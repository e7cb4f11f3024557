"""Differential fuzz of the device inflater against zlib: texts of many statistics x levels x strategies x chunk sizes.
Whatever ss_gz_inflate_gpu returns with SS_OK must equal the text; anything else must be SS_ERANGE."""
import ctypes as C, os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from strainscan_amd import _lib
L = _lib.lib(); _lib.require_gpu()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def text_of(rs, kind, n):
    if kind == 0:      # FASTQ-like, variable read length
        out = []
        tot = 0
        lut = np.frombuffer(b"ACGTN", np.uint8)
        while tot < n:
            ln = int(rs.randint(30, 300))
            s = lut[rs.randint(0, 5 if rs.rand() < 0.05 else 4, ln)].tobytes()
            q = (np.clip(38 - np.abs(rs.normal(0, rs.randint(1, 8), ln)).astype(np.int64), 2, 40) + 33).astype(np.uint8).tobytes()
            r = b"@read%d/%d\n" % (rs.randint(0, 10 ** 9), rs.randint(1, 3)) + s + b"\n+\n" + q + b"\n"
            out.append(r); tot += len(r)
        return b"".join(out)
    if kind == 1:      # random bytes (stored blocks)
        return rs.randint(0, 256, n).astype(np.uint8).tobytes()
    if kind == 2:      # long runs and short periods (overlapping matches, dist 1..8)
        out = []
        tot = 0
        while tot < n:
            per = rs.randint(1, 9)
            unit = rs.randint(0, 256, per).astype(np.uint8).tobytes()
            rep = int(rs.randint(1, 400))
            out.append(unit * rep); tot += per * rep
            if rs.rand() < 0.3:
                junk = rs.randint(0, 256, rs.randint(1, 200)).astype(np.uint8).tobytes()
                out.append(junk); tot += len(junk)
        return b"".join(out)
    if kind == 3:      # skewed alphabet (long Huffman codes for the rare symbols)
        p = 1.0 / np.arange(1, 257) ** rs.uniform(0.8, 2.5)
        p /= p.sum()
        return rs.choice(256, size=n, p=p).astype(np.uint8).tobytes()
    if kind == 4:      # text with far repeats (32 KB window edge)
        base = rs.randint(97, 123, 40000).astype(np.uint8).tobytes()
        out = []
        tot = 0
        while tot < n:
            a = rs.randint(0, len(base) - 600)
            out.append(base[a:a + rs.randint(3, 600)]); tot += len(out[-1])
            if rs.rand() < 0.5:
                out.append(rs.randint(97, 123, rs.randint(1, 3000)).astype(np.uint8).tobytes()); tot += len(out[-1])
        return b"".join(out)
    # mixture
    parts = [text_of(rs, k, n // 4) for k in (0, 2, 3, 4)]
    return b"".join(parts)


t_end = time.time() + budget
it = ok = declined = 0
seed = seed0
while time.time() < t_end:
    rs = np.random.RandomState(seed)
    kind = seed % 6
    n = int(rs.choice([300_000, 2_000_000, 6_000_000, 20_000_000]))
    txt = text_of(rs, kind, n)
    level = int(rs.choice([1, 2, 4, 6, 9]))
    strat = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rs.randint(0, 6))]
    mem = int(rs.choice([8, 9, 4, 1]))
    n_members = int(rs.choice([1, 1, 1, 2, 3, 5]))                       # lanes joined with cat
    cuts = sorted(int(x) for x in rs.randint(0, len(txt) + 1, n_members - 1))
    gz = b""
    for a, b in zip([0] + cuts, cuts + [len(txt)]):
        co = zlib.compressobj(level, zlib.DEFLATED, 31, mem, strat)
        gz += co.compress(txt[a:b]) + co.flush()
    p = "/tmp/fuzz_gi_%d.gz" % os.getpid()
    open(p, "wb").write(gz)
    os.environ["SS_GZ_CHUNK"] = str(int(rs.choice([4096, 8192, 16384, 32768, 65536])))
    t = C.c_void_p(); m = C.c_uint64()
    rc = L.ss_gz_inflate_gpu(os.fsencode(p), C.byref(t), C.byref(m))
    if rc == 0:
        got = C.string_at(t, m.value); L.ss_gz_free(t)
        if got != txt:
            print("MISMATCH seed", seed, "kind", kind, "n", len(txt), "level", level, "strategy", strat, "memLevel", mem, "chunk", os.environ["SS_GZ_CHUNK"], flush=True)
            sys.exit(1)
        ok += 1
    elif rc == -34:
        declined += 1
    else:
        print("UNEXPECTED rc", rc, "seed", seed, flush=True)
        sys.exit(1)
    it += 1
    seed += 1
os.remove(p)
print("fuzz: %d files, %d inflated and equal, %d declined, seeds %d..%d" % (it, ok, declined, seed0, seed - 1))

"""The sharded identification path on the device, several ranks on ONE GPU: every rank is a separate process with its
own torch.distributed rank (backend gloo -- RCCL refuses two ranks on one device; gloo stages the GPU tensors itself),
parses and scans only ITS share of the reads (ss_reads_load / ss_scan_files_shard with shard_rank, shard_world) and
the product's collectives -- dist.exchange_touched for the tree's node statistics, dist.allreduce_table where single
rows are needed -- must reproduce the reference's golden results (tests/golden/l1_search.json) bit for bit, as the
single-process run does.  This is BASELINE.json configs[2]'s code path at world sizes 2, 3 and 8 (eight ranks on one
device: the rehearsal of an 8-GPU node that was never available), plus the configs[0]-shaped 105-node database."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import scenarios as sc

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import contextlib, io, json, os, sys
import numpy as np
sys.path.insert(0, %(repo)r)
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="file://" + os.environ["SS_TEST_STORE"], rank=rank, world_size=world)   # (no fixed port: nothing to collide with)
from strainscan_amd import identify, identify_low_mem, identify_low_depth, db as ssdb, dist as sdist, _lib
assert sdist.is_distributed() and sdist.rank_world() == (rank, world)
jobs = json.load(open(%(jobs)r))
out = []
for job in jobs:
    mod = {"identify": identify, "identify_low_mem": identify_low_mem}[job["module"]]
    np.random.seed(job["seed"])
    if job.get("stream"):
        ssdb.RESIDENT_LIMIT_BYTES = 0            # no resident read set: ss_scan_files_shard streams this rank's chunks
    else:
        ssdb.RESIDENT_LIMIT_BYTES = 1 << 40
    ssdb.clear_cache()
    buf = io.StringIO()
    err = res = None
    with contextlib.redirect_stdout(buf):
        try:
            res = mod.identify_cluster(tuple(job["fq"]), job["tdb"], list(job["cutoff"]))
        except BaseException as e:
            err = type(e).__name__
    img = ssdb.tree_image(job["tdb"], job["module"] == "identify")
    st = img.node_stats()
    rec = dict(error=err, result=None if res is None else {str(k): {a: (b if isinstance(b, (int, str)) else float(b)) for a, b in v.items()} for k, v in res.items()},
               stats=[[int(x) for x in (s["length"], s["n_pos"], s["n_kept"], s["sum_kept"], s["median2"])] for s in st],
               rows_global=bool(img._rows_global), text=buf.getvalue())
    if job.get("ranks"):
        r, e2, _ = None, None, None
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                r = identify_low_depth.identify_ranks(tuple(job["fq"]), job["tdb"])
        except BaseException as e:
            e2 = type(e).__name__
        rec["ranks"] = dict(error=e2, result=None if r is None else [[int(a), float(b)] for a, b in r])
    out.append(rec)
out.append(dict(index_events=dict(ssdb.INDEX_EVENTS)))
json.dump(out, open(os.path.join(%(out)r, "rank%%d.json" %% rank), "w"))
dist.barrier()
dist.destroy_process_group()
'''


def _spawn(world, jobs, tmp_path):
    import socket
    jp = tmp_path / "jobs.json"
    jp.write_text(json.dumps(jobs))
    code = WORKER % dict(repo=REPO, jobs=str(jp), out=str(tmp_path))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SS_IMAGE_CACHE=str(tmp_path / "cache"),      # one cache for the node
                   SS_TEST_STORE=str(tmp_path / ("store_%d" % port)))
        env.pop("STRAINSCAN_QUIET", None)
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stderr=subprocess.PIPE))
    errs = [p.communicate(timeout=900)[1].decode()[-3000:] for p in procs]
    assert all(p.returncode == 0 for p in procs), errs
    return [json.loads((tmp_path / ("rank%d.json" % r)).read_text()) for r in range(world)]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_identify_equals_golden(world, golden_dir, l1_dbs, l1_reads, tmp_path):
    from tests import hostlogic as hl
    from oracle import oracle as orc
    with open(os.path.join(golden_dir, "l1_search.json")) as f:
        golden = json.load(f)
    jobs, wants = [], []
    for sname in ("A_mix3", "B_mix", "A_leaf6_single", "A_none", "D_one"):
        dbn = sc.L1_SAMPLES[sname][0]
        tdb = os.path.join(l1_dbs[dbn]["db_dir"], "Tree_database")
        for run in golden[sname]["runs"]:
            if run["cutoff"] != [0.1, 0.4, 1] and sname != "B_mix":
                continue
            jobs.append(dict(module=run["module"], seed=sc.POISSON_SEED, fq=[l1_reads[sname][0], ""], tdb=tdb, cutoff=run["cutoff"],
                             stream=(len(jobs) % 3 == 1), ranks=(len(jobs) % 4 == 0)))
            wants.append((sname, run))
    # a paired / gz input: this rank's share of two files, one of them inflated once per node
    import gzip
    fq, data = l1_reads["A_mix3"]
    recs = data.split(b"@r")[1:]
    half = len(recs) // 2
    p1, p2 = tmp_path / "a_1.fq", tmp_path / "a_2.fq.gz"
    p1.write_bytes(b"".join(b"@r" + r for r in recs[:half]))
    with gzip.open(p2, "wb") as f:
        f.write(b"".join(b"@r" + r for r in recs[half:]))
    run0 = [r for r in golden["A_mix3"]["runs"] if r["module"] == "identify" and r["cutoff"] == [0.1, 0.4, 1]][0]
    tdbA = os.path.join(l1_dbs["A"]["db_dir"], "Tree_database")
    jobs.append(dict(module="identify", seed=sc.POISSON_SEED, fq=[str(p1), str(p2)], tdb=tdbA, cutoff=[0.1, 0.4, 1]))
    wants.append(("A_mix3", run0))
    outs = _spawn(world, jobs, tmp_path)
    # one image cache for all ranks (db.rank0_first): rank 0 parsed and built every index and exported it, the others
    # waited at the barrier and imported -- no rank but 0 built anything, and no temp file is left
    ev = [o.pop()["index_events"] for o in outs]
    assert ev[0]["built"] >= 1 and all(e["built"] == 0 and e["imported"] >= 1 for e in ev[1:]), ev
    assert all(f.endswith(".bin") for f in os.listdir(tmp_path / "cache"))
    any_rows = False
    for ji, (sname, run) in enumerate(wants):
        recs_ = [o[ji] for o in outs]
        for r in recs_[1:]:                                  # every rank holds the same global answer
            assert r["result"] == recs_[0]["result"] and r["stats"] == recs_[0]["stats"] and r["error"] == recs_[0]["error"]
        got = recs_[0]
        tag = (world, sname, run["module"], run["cutoff"])
        assert got["error"] == run["error"], (tag, got["text"][-300:])
        if got["error"] is None:
            hl.assert_result_equal({int(k): v for k, v in got["result"].items()}, run["result"], tag)
        assert [g[0] for g in hl.parse_trace(got["text"])] == [w[0] for w in run["trace"]], tag
        any_rows = any_rows or got["rows_global"]
        # node statistics = the oracle's match_node on the golden counts of the WHOLE sample
        info = l1_dbs[sc.L1_SAMPLES[sname][0]]
        kfa = open(os.path.join(info["db_dir"], "Tree_database", "kmer.fa"), "rb").read()
        want_c, want_v = orc.jellyfish_count(kfa, [l1_reads[sname][1]], k=31, upper=(run["module"] == "identify"))
        from strainscan_amd import db as ssdb
        ids = ssdb.load_tree(os.path.join(info["db_dir"], "Tree_database"), 31).ids
        for j, nid in enumerate(ids):
            o = orc.match_node(want_c, want_v, np.array(info["row_of_node"][nid]))
            assert got["stats"][j] == [o["length"], o["n_pos"], o["n_kept"], o["sum_kept"], int(round(2 * o["median"])) if o["n_pos"] else 0], (tag, nid)
        if "ranks" in got:
            want_r = golden[sname]["ranks"]
            assert got["ranks"]["error"] == want_r["error"]
            if want_r["result"] is not None:
                assert [a for a, _ in got["ranks"]["result"]] == [a for a, _ in want_r["result"]]
                for (_, b), (_, wb) in zip(got["ranks"]["result"], want_r["result"]):
                    assert abs(b - wb) <= 1e-12 * max(1.0, abs(wb))
    assert any_rows            # B_mix's Poisson branch asked for single rows: the full row vector was all-reduced there


@pytest.mark.parametrize("world", [3, 8])
def test_sharded_mid_identify_equals_golden(world, golden_dir, mid_dbs, tmp_path):
    """The configs[0]-shaped database (53 clusters / 157 strains / 105 nodes, tests/scenarios_mid.py) with the reads sharded
    over 3 and 8 ranks: every rank ends with the reference's own result dict, visit order and identify_ranks scores
    (tests/golden/mid_l1.json, recorded from the real reference)."""
    from tests import hostlogic as hl
    with open(os.path.join(golden_dir, "mid_l1.json")) as f:
        golden = json.load(f)
    tdb = os.path.join(mid_dbs["DB_M"]["db_dir"], "Tree_database")
    jobs, wants = [], []
    for sname in ("M_mix", "M_recon", "M_low2"):
        for run in golden[sname]["runs"]:
            if run["cutoff"] not in ([0.1, 0.4, 1], [0.05, 0.05, 1]):
                continue
            jobs.append(dict(module=run["module"], seed=sc.POISSON_SEED, fq=[mid_dbs["reads"][sname][0], ""], tdb=tdb, cutoff=run["cutoff"],
                             stream=(len(jobs) % 3 == 1), ranks=(len(jobs) % 4 == 0)))
            wants.append((sname, run))
    outs = _spawn(world, jobs, tmp_path)
    ev = [o.pop()["index_events"] for o in outs]
    assert ev[0]["built"] >= 1 and all(e["built"] == 0 and e["imported"] >= 1 for e in ev[1:]), ev
    for ji, (sname, run) in enumerate(wants):
        recs_ = [o[ji] for o in outs]
        for r in recs_[1:]:
            assert r["result"] == recs_[0]["result"] and r["stats"] == recs_[0]["stats"] and r["error"] == recs_[0]["error"]
        got = recs_[0]
        tag = (world, sname, run["module"], run["cutoff"])
        assert got["error"] == run["error"], (tag, got["text"][-300:])
        hl.assert_result_equal({int(k): v for k, v in got["result"].items()}, run["result"], tag)
        got_tr = hl.parse_trace(got["text"])
        assert [g[0] for g in got_tr] == [w[0] for w in run["trace"]], tag
        for a, w in zip(got_tr, run["trace"]):
            if len(w) == 4:
                assert abs(a[1] - w[1]) < 2e-6 and abs(a[2] - w[2]) < 2e-6 and a[3] == w[3], (tag, a, w)
        if "ranks" in got:
            want_r = golden[sname]["ranks"]
            assert got["ranks"]["error"] == want_r["error"]
            assert [a for a, _ in got["ranks"]["result"]] == [a for a, _ in want_r["result"]]
            for (_, b), (_, wb) in zip(got["ranks"]["result"], want_r["result"]):
                assert abs(b - wb) <= 1e-12 * max(1.0, abs(wb))


GZ_WORKER = r'''
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, %(repo)r)
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="file://" + os.environ["SS_TEST_STORE"], rank=rank, world_size=world)   # (no fixed port: nothing to collide with)
from strainscan_amd import dist as sdist, _lib
job = json.load(open(%(jobs)r))
for which, name in ((1, "T_INJECT_ENTRY"), (2, "T_INJECT_DECLINE"), (3, "T_SKIP_CHAIN")):      # the test's hooks: an explicit call, not the environment
    if os.environ.get(name):
        _lib.check(_lib.lib().ss_test_hook(which, int(os.environ[name])), "ss_test_hook")
kdb = _lib.KmerDB.from_text(open(job["kfa"], "rb").read(), 31, True)
nrec, nb = sdist.scan_files_sharded(kdb, job["paths"], allreduce=True)
counts = kdb.counts_rows()
rset = sdist.load_agreed(job["paths"], lambda use: _lib.ReadSet(use, rank, world), discard=lambda r: r.close())
own = rset.info()["n_records"]
rset.close()
a, b, rf, rp = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
_lib.lib().ss_gz_gpu_counters(C.byref(a), C.byref(b))
_lib.lib().ss_gz_range_counters(C.byref(rf), C.byref(rp))
np.save(os.path.join(%(out)r, "counts%%d.npy" %% rank), counts)
json.dump(dict(nrec=int(nrec), own=int(own), handled=int(a.value), declined=int(b.value), range_files=int(rf.value), range_pieces=int(rp.value),
               chain_failures=len(sdist.CHAIN_FAILURES)),
          open(os.path.join(%(out)r, "rank%%d.json" %% rank), "w"))
dist.barrier()
if sdist.CHAIN_FAILURES:
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(0)           # (a helper thread still sits in its abandoned receive: no orderly shutdown of the group)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,decline,mode", [(2, False, "range"), (3, False, "range"), (2, False, "whole"), (3, False, "members"), (2, False, "members"),
                                                  (2, True, "range"), (3, True, "range"), (2, False, "entry100"), (2, False, "entry128"), (3, False, "entry129"),
                                                  (2, False, "bgzip"), (3, False, "bgzip"), (2, False, "gone"), (3, False, "gone"),
                                                  (8, False, "range"), (8, False, "bgzip"), (8, True, "range")])
def test_sharded_scan_of_gz_pair_on_the_device(world, decline, mode, tmp_path):
    """A pair of .fastq.gz files under torch.distributed; the summed row counts equal the single-process scan of the plain
    text, the ranks' record counts add up, and the device inflater (not the host's) did the work.
    `range`: the ranks SHARE every file's inflation (ss_gz_set_range: slices of the deflate data -- SS_GZ_SLICE_KB makes
    them small here -- a chain of messages for the windows, newline counts and straddling records, CRC-32 down the chain).
    `whole` (SS_GZ_RANGE=0): every rank inflates both files on its GPU and keeps its blocks of 4096 records.
    `members`: one file is two members joined with cat -- the member that ends inside a slice is checked against its
    trailer there, the next starts with nothing in front of it, CRC-32 and length of the open member travel down the chain.
    `bgzip`: both files are BGZF -- a slice's chunks are the members that begin in its byte range, the chain carries the newline
    count and the straddling record.  `decline`: the device path of rank 1 alone declines (test hook: it still serves the chain) --
    dist.load_agreed must move ALL ranks on, in the end to the host inflaters (one inflate into /dev/shm, parse chunks
    shared out), or reads would be counted twice or not at all.  `entryN`: a wrong entry point (a position inside a
    block, test hook) in search chunk N -- inside a slice the chunk in front of it runs over it as in the whole-file path;
    at a slice's edge (128 = the first chunk of slice 1) the slices no longer meet and the shared inflation is declined.
    `gone`: rank 1 leaves the shared inflation WITHOUT serving the chain (test hook: what a crashed peer looks like to the
    others) -- their receives end after SS_GZ_CHAIN_TIMEOUT (3 s here), status -1 travels on, all ranks fall back to the
    whole-file path, range mode is switched off for the process group; counts as ever."""
    import gzip
    import socket
    from strainscan_amd import _lib as L
    rs = np.random.RandomState(5)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rows = lut[rs.randint(0, 4, (30000, 31))]
    kfa = b"".join(b">1\n" + r.tobytes() + b"\n" for r in rows)
    n = 50000 if world < 8 else 280000      # (eight ranks: files of ~8.5 MB = 17 slices of the smallest size there is, 128 search chunks)
    lvl = 6 if world < 8 else 1
    reads = lut[rs.randint(0, 4, (n, 150))]
    for i in range(0, n, 2):
        o = rs.randint(0, 119)
        reads[i, o:o + 31] = rows[rs.randint(0, rows.shape[0])]
    fq = [b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + bytes(35 + (i + j) % 30 for j in range(150)) + b"\n" for i in range(n)]
    half = n // 2 + 777
    kp = tmp_path / "k.fa"
    kp.write_bytes(kfa)
    plain = tmp_path / "all.fq"
    plain.write_bytes(b"".join(fq))
    p1, p2 = tmp_path / "s_1.fq.gz", tmp_path / "s_2.fq.gz"
    p1.write_bytes(gzip.compress(b"".join(fq[:half]), lvl))
    if mode == "bgzip":
        from tests.test_ginflate_gpu import _bgzf
        p1.write_bytes(_bgzf(b"".join(fq[:half])))
        p2.write_bytes(_bgzf(b"".join(fq[half:]), block=33333, level=4))
    elif mode == "members":
        q = half + (n - half) // 3
        p2.write_bytes(gzip.compress(b"".join(fq[half:q]), 6) + gzip.compress(b"".join(fq[q:]), 1))
    else:
        p2.write_bytes(gzip.compress(b"".join(fq[half:]), lvl))
    assert min(p1.stat().st_size, p2.stat().st_size) > ((1 << 20) if world < 8 else (7 << 20))
    db = L.KmerDB.from_text(kfa, 31, True)
    db.scan_files([str(plain)])
    want = db.counts_rows().copy()
    db.close()
    assert want.sum() >= n // 2
    jp = tmp_path / "job.json"
    jp.write_text(json.dumps(dict(kfa=str(kp), paths=[str(p1), str(p2)])))
    code = GZ_WORKER % dict(repo=REPO, jobs=str(jp), out=str(tmp_path))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SS_TEST_STORE=str(tmp_path / ("store_%d" % port)))
        env.pop("SS_GZ_GPU", None)
        env["SS_GZ_SLICE_KB"] = "256" if world < 8 else "64"      # (a slice is 128 search chunks at least: 512 KB; eight ranks: ~16 slices per
                                                                  #  file, two per rank, 8 x 2 hops of the chain)
        env["SS_GZ_CHUNK"] = "4096"          # (search chunks of 4 KB: slices of 128 of them, three or so per file)
        if mode == "whole":
            env["SS_GZ_RANGE"] = "0"
        if mode.startswith("entry"):
            env["T_INJECT_ENTRY"] = mode[5:]
        if decline and r == 1:
            env["T_INJECT_DECLINE"] = "1"
        if mode == "gone":                   # rank 1 leaves range mode without serving the chain: the others' bounded wait
            env["SS_GZ_CHAIN_TIMEOUT"] = "3"
            if r == 1:
                env["T_SKIP_CHAIN"] = "1"
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stderr=subprocess.PIPE))
    errs = [p.communicate(timeout=600)[1].decode()[-6000:] for p in procs]
    if not all(p.returncode == 0 for p in procs):
        (tmp_path / "errs.txt").write_text("\n=====\n".join(errs))
        print("\n=====\n".join(e[-1500:] for e in errs))
    assert all(p.returncode == 0 for p in procs), [p.returncode for p in procs]
    if os.environ.get("SS_INGEST_TRACE"):
        print("\n".join("rank %d:\n%s" % (r, e) for r, e in enumerate(errs)))
    infos = [json.loads((tmp_path / ("rank%d.json" % r)).read_text()) for r in range(world)]
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("counts%d.npy" % r)), want), r          # the global counts, on every rank
        if not decline and mode in ("range", "bgzip"):      # two files, scanned and loaded, both times shared between the ranks
            assert infos[r]["handled"] == 4 and infos[r]["declined"] == 0 and infos[r]["range_files"] == 4 and infos[r]["range_pieces"] >= 4, (infos, errs)
        elif not decline and mode == "whole":
            assert infos[r]["handled"] == 4 and infos[r]["declined"] == 0 and infos[r]["range_files"] == 0
        elif mode == "members":                  # the second file is two members: followed inside the slices like one
            assert infos[r]["handled"] == 4 and infos[r]["declined"] == 0 and infos[r]["range_files"] == 4, (infos, errs)
        elif decline:   # the ranks that were not declined did inflate on the device (a rank whose slices all lie in front of
            #             rank 1's even finished its share of the shared inflation), and gave that up
            assert infos[r]["handled"] == 0 if r == 1 else infos[r]["handled"] >= 4
    assert sum(i["nrec"] for i in infos) == n and sum(i["own"] for i in infos) == n
    assert decline or all(i["nrec"] > 0 for i in infos)      # (parse chunks are 24 MB: these small files are one chunk each)
    if mode == "gone":
        assert sum(i["chain_failures"] for i in infos) >= 1 and all(i["range_files"] == 0 for i in infos), infos
        assert all(i["handled"] >= 2 for i in infos)         # the whole-file device path took over (not the host inflaters)
    if mode == "entry100":                                   # run over inside a slice: still shared
        assert all(i["range_files"] == 4 for i in infos)
    if mode in ("range", "bgzip") and not decline:           # every rank inflated about its share, not everything
        assert max(i["own"] for i in infos) < 0.75 * n

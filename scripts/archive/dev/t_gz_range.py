#!/usr/bin/env python3
"""Range mode of the device inflater at a realistic size: N ranks on ONE GPU (gloo) load a pair of .fastq.gz files of
`reads` reads each through dist.load_agreed; the summed counts must equal the single-process scan of the same files, and
the time each rank spends in the load is printed (SS_INGEST_TRACE=1 shows the kernels' share).
    t_gz_range.py [reads per file = 1000000] [world = 2] [gzip level = 6]"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

WORKER = r'''
import os, sys, time, json
import numpy as np
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="file://" + os.path.join(%(out)r, "store_" + os.environ.get("SS_GZ_RANGE", "1")), rank=rank, world_size=world)
from strainscan_amd import dist as sdist, _lib
import ctypes as C
job = json.load(open(%(job)r))
kdb = _lib.KmerDB(np.load(job["keys"]), np.ones(job["n_keys"], np.uint8), 31, True)
for it in range(2):
    dist.barrier()
    t0 = time.perf_counter()
    rs = sdist.load_agreed(job["paths"], lambda use: _lib.ReadSet(use, rank, world), discard=lambda r: r.close())
    _lib.check(_lib.lib().ss_device_sync(), "sync")
    dt = time.perf_counter() - t0
    own = rs.info()["n_records"]
    if it == 0:
        rs.close()
kdb.reset()
rs.scan_into(kdb)
_lib.check(_lib.lib().ss_device_sync(), "sync")
t = torch.from_numpy(kdb.counts_rows().astype(np.int64))
dist.all_reduce(t)
rf, rp = C.c_uint64(), C.c_uint64()
_lib.lib().ss_gz_range_counters(C.byref(rf), C.byref(rp))
json.dump(dict(own=int(own), load_s=dt, range_files=int(rf.value), pieces=int(rp.value)), open(os.path.join(%(out)r, "r%%d.json" %% rank), "w"))
if rank == 0:
    np.save(os.path.join(%(out)r, "sum.npy"), t.numpy())
dist.barrier()
dist.destroy_process_group()
'''


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    level = sys.argv[3] if len(sys.argv) > 3 else "6"
    import torch
    import bench
    from strainscan_amd import _lib
    dev = torch.device("cuda", 0)
    spec = bench.make_db(torch, dev, 53, seed=7, shape="sampled", hit_frac=0.05)
    reads = bench.make_reads(torch, dev, spec, 2 * n, seed=3, hit_frac=0.05)
    base = tempfile.mkdtemp(prefix="ss_gzrange_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    paths = []
    for f in range(2):
        p = os.path.join(base, "s_%d.fq" % (f + 1))
        bench.write_fastq(reads[f * n * 151:(f + 1) * n * 151], n, p, noisy_quality_seed=5 + f)
        paths.append(p)
    db = _lib.KmerDB(spec["keys"], np.ones(spec["keys"].size, np.uint8), 31, True)
    db.scan_files(paths)
    want = db.counts_rows().astype(np.int64)
    db.close()
    procs = [subprocess.Popen(["gzip", "-" + level, "-f", p]) for p in paths]
    assert all(pr.wait() == 0 for pr in procs)
    gz = [p + ".gz" for p in paths]
    np.save(os.path.join(base, "keys.npy"), spec["keys"])
    job = os.path.join(base, "job.json")
    json.dump(dict(paths=gz, keys=os.path.join(base, "keys.npy"), n_keys=int(spec["keys"].size)), open(job, "w"))
    del reads, spec
    torch.cuda.empty_cache()
    out = {}
    for mode in ("range", "whole"):
        code = WORKER % dict(root=ROOT, job=job, out=base)
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
            if mode == "whole":
                env["SS_GZ_RANGE"] = "0"
            trace = os.environ.get("SS_INGEST_TRACE") and r == int(os.environ.get("SS_TRACE_RANK", "0"))
            if not trace:
                env.pop("SS_INGEST_TRACE", None)
            procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stderr=(open(os.path.join(ROOT, "gpurun_out", "gzrange_%s.err" % mode), "w") if trace else None)))
        assert all(p.wait() == 0 for p in procs), mode
        infos = [json.load(open(os.path.join(base, "r%d.json" % r))) for r in range(world)]
        got = np.load(os.path.join(base, "sum.npy"))
        out[mode] = dict(counts_equal=bool(np.array_equal(got, want)), records=sum(i["own"] for i in infos), per_rank=infos)
    print(json.dumps(dict(reads=2 * n, world=world, gz_mb=round(sum(os.path.getsize(p) for p in gz) / 1e6, 1), **out)))
    import shutil
    shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    main()

#!/bin/bash
# does it matter WHEN the inflater's scratch arenas are allocated?  early (a first .gz load in a fresh process) vs late (after the
# database, 20 M resident reads and some allocation churn, as in bench.py)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for mode in late early early_then_release; do
MODE=$mode timeout 400 python - <<'PY' 2>&1 | grep -v amdgpu
import os, sys, time, subprocess
sys.path.insert(0, '.')
import numpy as np, torch, bench
from strainscan_amd import _lib
dev = torch.device("cuda", 0)
mode = os.environ["MODE"]
base = "/dev/shm/ss_gzf_%d" % os.getpid(); os.makedirs(base)
small = torch.randint(0, 4, (1_000_000 * 151,), dtype=torch.uint8, device=dev)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
flat = lut[small.long()]; flat[150::151] = 10
paths = []
for f in range(2):
    p = os.path.join(base, "gz_%d.fq" % (f + 1))
    bench.write_fastq(flat[f * 500_000 * 151:(f + 1) * 500_000 * 151], 500_000, p, noisy_quality_seed=77 + f)
    paths.append(p)
for pr in [subprocess.Popen(["gzip", "-6", "-f", p]) for p in paths]: pr.wait()
gz = [p + ".gz" for p in paths]
del small, flat
def load():
    t0 = time.perf_counter(); rs = _lib.ReadSet(gz); _lib.check(_lib.lib().ss_device_sync(), "sync"); dt = time.perf_counter() - t0
    rs.close(); return dt * 1e3
if mode.startswith("early"):
    _lib.warm_up(gz=2); load(); load()
if mode == "early_then_release":     # the arenas, pinned buffers and pooled streams go; what the stream-ordered allocator keeps stays
    _lib.check(_lib.lib().ss_gz_gpu_release(), "release")
spec = bench.make_db(torch, dev, 103, seed=20231013)
reads = bench.make_reads(torch, dev, spec, 20_000_000, seed=2, hit_frac=0.05)
churn = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(6)]
del churn[::2]
rs20 = _lib.ReadSet.from_flat_dev(reads.data_ptr(), reads.numel(), order=True); rs20.close()
if mode == "late":
    _lib.warm_up(gz=2); load(); load()
ts = sorted(load() for _ in range(25))
print(mode, "min %.1f q1 %.1f median %.1f q3 %.1f ms" % (ts[0], ts[len(ts) // 4], ts[len(ts) // 2], ts[3 * len(ts) // 4]))
import shutil; shutil.rmtree(base)
PY
done

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_gz; mkdir -p $O; cd $R
for T in 1; do
SS_GZ_PIECES=$T timeout 600 python - <<'PY' > $O/trace_$T.txt 2>&1
import os, sys, time, subprocess
sys.path.insert(0, '.')
import numpy as np, torch, bench
from strainscan_amd import _lib
dev = torch.device("cuda", 0)
spec = bench.make_db(torch, dev, 103, seed=20231013)
reads = bench.make_reads(torch, dev, spec, 1_000_000, seed=2, hit_frac=0.05)
base = "/dev/shm/ss_gzt_%d" % os.getpid(); os.makedirs(base)
paths = []
for f in range(2):
    p = os.path.join(base, "gz_%d.fq" % (f + 1))
    bench.write_fastq(reads[f * 500_000 * 151:(f + 1) * 500_000 * 151], 500_000, p, noisy_quality_seed=77 + f)
    paths.append(p)
for pr in [subprocess.Popen(["gzip", "-6", "-f", p]) for p in paths]: pr.wait()
gz = [p + ".gz" for p in paths]
os.environ["SS_INGEST_TRACE"] = "1"
_lib.warm_up(gz=2)
for it in range(8):
    pass
    t0 = time.perf_counter(); rs = _lib.ReadSet(gz); _lib.check(_lib.lib().ss_device_sync(), "sync"); dt = time.perf_counter() - t0
    print("load %d: %.1f ms" % (it, dt * 1e3), flush=True); rs.close()
import shutil; shutil.rmtree(base)
PY
echo pieces $T; grep -v "amdgpu.ids" $O/trace_$T.txt | awk "/^load 5/{f=1;next} /^load 6/{f=0} f" | grep -v "candidate\|entries\|segments\|amdgpu\|reorder" | cut -c1-150
done

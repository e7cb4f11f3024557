#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_gz2; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ginflate_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/pair.txt
import os, sys, time, subprocess
sys.path.insert(0, '.')
import numpy as np, torch, bench
from strainscan_amd import _lib
dev = torch.device("cuda", 0)
spec = bench.make_db(torch, dev, 103, seed=20231013)
reads = bench.make_reads(torch, dev, spec, 2_000_000, seed=2, hit_frac=0.05)
base = "/dev/shm/ss_gzt_%d" % os.getpid(); os.makedirs(base)
paths = []
for f in range(2):
    p = os.path.join(base, "gz_%d.fq" % (f + 1))
    bench.write_fastq(reads[f * 1_000_000 * 151:(f + 1) * 1_000_000 * 151], 1_000_000, p, noisy_quality_seed=77 + f)
    paths.append(p)
for pr in [subprocess.Popen(["gzip", "-6", "-f", p]) for p in paths]: pr.wait()
gz = [p + ".gz" for p in paths]
_lib.warm_up(gz=2)
for pipe in ("1", "0", "1", "0"):
    os.environ["SS_GZ_PIPELINE"] = pipe
    ts = []
    for it in range(6):
        t0 = time.perf_counter(); rs = _lib.ReadSet(gz); _lib.check(_lib.lib().ss_device_sync(), "sync"); ts.append((time.perf_counter() - t0) * 1e3)
        n = rs.info()["n_records"]; rs.close()
    print("pipeline", pipe, "records", n, "ms", [round(t, 1) for t in ts], flush=True)
import shutil; shutil.rmtree(base)
PY

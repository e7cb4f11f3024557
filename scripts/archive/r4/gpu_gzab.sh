#!/bin/bash
# .gz pair load time under switches that the library reads per call, interleaved in ONE process (the boxes' host side is noisy)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_gz; mkdir -p $O; cd $R
timeout 800 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
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
_lib.warm_up(gz=2)
variants = [dict(SS_GZ_UPLOAD_TURNS="0"), dict(SS_GZ_UPLOAD_TURNS="1"), dict(SS_GZ_UPLOAD_TURNS="1", SS_READS_ORDER="file")]
times = [[] for _ in variants]
for it in range(41):
    for vi, v in enumerate(variants):
        for k in ("SS_GZ_PIECES", "SS_GZ_UPLOAD_DIV", "SS_GZ_UPLOAD_THREADS", "SS_GZ_UPLOAD_TURNS", "SS_READS_ORDER"): os.environ.pop(k, None)
        os.environ.update(v)
        t0 = time.perf_counter(); rs = _lib.ReadSet(gz); _lib.check(_lib.lib().ss_device_sync(), "sync"); dt = time.perf_counter() - t0
        rs.close()
        if it: times[vi].append(dt * 1e3)
for v, t in zip(variants, times):
    t = sorted(t)
    print("%-40s min %.1f  q1 %.1f  median %.1f  q3 %.1f  p90 %.1f  max %.1f ms" % (v, t[0], t[len(t) // 4], t[len(t) // 2], t[3 * len(t) // 4], t[9 * len(t) // 10], t[-1]))
import shutil; shutil.rmtree(base)
PY

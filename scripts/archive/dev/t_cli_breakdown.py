#!/usr/bin/env python3
"""Where a fresh `strainscan` process spends its time (database image cached): writes the synthetic database + sample of
scripts/bench_cli.py, warms the cache, then runs the CLI with -X importtime and SS_INGEST_TRACE and prints the totals."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from scripts import bench_cli
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
dev = torch.device("cuda", 0)
base = "/dev/shm/ss_clib_%d" % os.getpid()
os.makedirs(base)
os.environ["SS_IMAGE_CACHE"] = os.path.join(base, "cache")
spec = bench.make_db(torch, dev, 823, seed=20231013, shape="sampled")
tdir = bench_cli.write_db(torch, dev, spec, 823, base)
r = bench.make_reads(torch, dev, spec, n_reads, seed=2, hit_frac=0.05)
half = n_reads // 2
fq = [os.path.join(base, "s_%d.fq" % (i + 1)) for i in range(2)]
bench.write_fastq(r[: half * 151], half, fq[0]); bench.write_fastq(r[half * 151:], n_reads - half, fq[1])
del r, spec
torch.cuda.empty_cache()
cmd = [sys.executable, "-m", "strainscan_amd.StrainScan", "-i", fq[0], "-j", fq[1], "-d", base, "-o", os.path.join(base, "o")]
subprocess.run(cmd, cwd=ROOT, capture_output=True)                       # warms the image cache
for extra_env, label in (({}, "plain"), ({"SS_INGEST_TRACE": "1", "SS_CLI_TRACE": "1"}, "traced")):
    t0 = time.perf_counter()
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, **extra_env))
    print(label, "wall %.3f s" % (time.perf_counter() - t0))
    if label == "traced":
        print("\n".join(ln for ln in (p.stderr + p.stdout).split("\n") if "[cli]" in ln or "ss_reads_load" in ln or "running time" in ln or "worker" in ln or "ginflate" in ln or "reorder" in ln)[:6000])
t0 = time.perf_counter()
p = subprocess.run([sys.executable, "-X", "importtime"] + cmd[1:], cwd=ROOT, capture_output=True, text=True)
print("importtime wall %.3f s" % (time.perf_counter() - t0))
rows = []
for ln in p.stderr.split("\n"):
    if ln.startswith("import time:") and "|" in ln:
        a = ln.split("|")
        try:
            rows.append((int(a[1]), a[2].strip()))
        except ValueError:
            pass
rows.sort(reverse=True)
print("top imports (cumulative us):", rows[:12])
import shutil; shutil.rmtree(base, ignore_errors=True)

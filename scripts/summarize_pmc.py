#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs of gpurun_out/pmc_* (per kernel, per launch) as text + JSON."""
import collections
import csv
import glob
import json
import os
import sys

O = sys.argv[1]
out = {}
lines = []
for d in sorted(glob.glob(os.path.join(O, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d)[4:]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            kn = r.get("Kernel_Name", "")
            if "anonymous namespace" not in kn or "at::native" in kn:
                continue
            short = kn.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1][:48]
            k = (short, r["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
        for (kn, cn), (n, v) in sorted(acc.items()):
            lines.append("%-14s %-28s %-24s launches=%d per_launch=%.6g" % (tag, kn, cn, n, v / n))
            out.setdefault(tag, {}).setdefault(kn, {})[cn] = v / n
print("\n".join(lines))
with open(os.path.join(O, "pmc_summary.txt"), "w") as f:
    f.write("# rocprofv3 --pmc <counters> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline [--calib-stream]\n")
    f.write("# one pass per counter group; FETCH_SIZE / WRITE_SIZE in KiB\n")
    f.write("\n".join(lines) + "\n")
with open(os.path.join(O, "pmc_summary.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)

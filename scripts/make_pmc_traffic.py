#!/usr/bin/env python3
"""gpurun_out/r2/<tag>/ (scripts/gpu_round2.sh) -> profiles/<prefix>_<tag>_{bench.json,kernel_stats.csv,pmc_summary.txt}
and one entry of profiles/pmc_traffic.json per configuration: corrected HBM bytes per launch of the scan kernel, read
requests, VALU occupancy -- what bench.py quotes in `roofline` (it cannot collect counters inside its timed region).

usage: make_pmc_traffic.py <prefix> <tag> [<tag> ...]     (calibration: the tag that has pmc_calib_fetch, else the stored one)"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix, tags = sys.argv[1], sys.argv[2:]
commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
table = json.load(open(path)) if os.path.exists(path) else {}


def scan_entry(d):
    for k, v in d.items():
        if k.startswith("scan_mini_kernel") or k.startswith("scan_kernel"):
            return v
    return {}


calib = table.get("_calibration", {})
for tag in tags:
    src = os.path.join(ROOT, "gpurun_out", "r2", tag)
    pm = json.load(open(os.path.join(src, "pmc_summary.json")))
    bench = json.load(open(os.path.join(src, "bench.json")))
    cfg = bench["config"]
    n_tiles = -(-cfg["reads_per_gpu"] * 151 // 992)
    streamed = n_tiles * 1024                       # 64 lanes x 16 B per tile of 992 positions
    if "calib_fetch" in pm:
        seen = scan_entry(pm["calib_fetch"])["FETCH_SIZE"] * 1024
        calib = dict(streamed_bytes=streamed, fetch_size_bytes=seen, seen_fraction=round(seen / streamed, 4),
                     note="all-'N' reads: the kernel only streams the base block with 16-byte loads; FETCH_SIZE reports this "
                          "fraction of it (MI355X_MICROARCH.md: wide streams are tallied at half); the unseen part is added to "
                          "the traffic of every configuration; random 16-byte page/bucket loads are single 64-byte requests "
                          "counted 1:1 (TCC_EA0_RDREQ x 64 B = FETCH_SIZE, RDREQ_32B = 0)",
                     source="profiles/%s_%s_pmc_summary.txt" % (prefix, tag), commit=commit)
    fetch = scan_entry(pm["fetch"])["FETCH_SIZE"] * 1024
    write = scan_entry(pm["write"])["WRITE_SIZE"] * 1024
    l2 = scan_entry(pm["l2"])
    va = scan_entry(pm.get("valu", {}))
    unseen = streamed * (1.0 - calib.get("seen_fraction", 0.547))
    key = "%s:%s:%g" % (cfg["table_layout"], cfg["db_shape"], cfg["hit_frac"])
    table[key] = dict(
        traffic_gb_per_launch=round((fetch + unseen + write) / 1e9, 2), fetch_size_gb=round(fetch / 1e9, 2),
        unseen_stream_gb=round(unseen / 1e9, 2), write_size_gb=round(write / 1e9, 2),
        rdreq_per_launch=l2.get("TCC_EA0_RDREQ_sum"), l2_hit_rate=round(l2["TCC_HIT_sum"] / (l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"]), 3),
        valu_insts_per_tile=(round(va["SQ_INSTS_VALU"] / n_tiles, 1) if va else None),
        # SQ_ACTIVE_INST_VALU counts per SIMD in quad-cycles (a wave64 VALU instruction keeps its SIMD busy four cycles);
        # SQ_BUSY_CYCLES is summed over the 32 shader engines; 1024 SIMDs
        valu_busy=(round(va["SQ_ACTIVE_INST_VALU"] * 4 / (va["SQ_BUSY_CYCLES"] / 32 * 1024), 3) if va else None),
        kernel_ms_at_collection=bench["roofline"]["kernel_ms"],
        source="profiles/%s_%s_pmc_summary.txt" % (prefix, tag), commit=commit)
    for name in ("bench.json", "kernel_stats.csv", "pmc_summary.txt"):
        shutil.copy(os.path.join(src, name), os.path.join(ROOT, "profiles", "%s_%s_%s" % (prefix, tag, name)))
    # keep this package's kernels only in the stats file
    ks = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (prefix, tag))
    lines = open(ks).read().split("\n")
    open(ks, "w").write("\n".join([lines[0]] + [ln for ln in lines[1:] if "anonymous namespace" in ln and "at::native" not in ln]) + "\n")
    print(key, table[key])
table["_calibration"] = calib
json.dump(table, open(path, "w"), indent=1, sort_keys=True)

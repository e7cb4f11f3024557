#!/usr/bin/env python3
"""gpurun_out/r4_binned_<shape>_pmc.txt (scripts/r4/gpu_pmc_round.sh) -> profiles/pmc_traffic.json keys
"mini:<shape>:0.05:binned" (the tree scan over the binned resident read set, what the product runs) and the committed copies
profiles/r04_binned_<shape>_pmc_summary.txt.   usage: add_binned_traffic.py <binned kernel ms sampled> <... contiguous>"""
import json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
table = json.load(open(path))
seen = table["_calibration"]["seen_fraction"]
commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()
n_tiles = -(-20_000_000 * 152 // 992)            # binned records are padded to 8 bytes: 152 per read
streamed = n_tiles * 1024
for shape, ms in zip(("sampled", "contiguous"), sys.argv[1:3]):
    src = os.path.join(ROOT, "gpurun_out", "r4_binned_%s_pmc.txt" % shape)
    v = {}
    for ln in open(src):
        m = re.search(r"(\S+)\s+launches=\d+ mean=(\S+)", ln)
        if m and "scan_mini" in ln:
            v[m.group(1)] = float(m.group(2))
    fetch, write = v["FETCH_SIZE"] * 1024, v["WRITE_SIZE"] * 1024
    unseen = streamed * (1.0 - seen)
    table["mini:%s:0.05:binned" % shape] = dict(
        traffic_gb_per_launch=round((fetch + unseen + write) / 1e9, 2), fetch_size_gb=round(fetch / 1e9, 2), unseen_stream_gb=round(unseen / 1e9, 2),
        write_size_gb=round(write / 1e9, 2), rdreq_per_launch=v["TCC_EA0_RDREQ_sum"],
        l2_hit_rate=round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 3),
        valu_insts_per_tile=round(v["SQ_INSTS_VALU"] / n_tiles, 1),
        valu_busy=round(v["SQ_ACTIVE_INST_VALU"] * 4 / (v["SQ_BUSY_CYCLES"] / 32 * 1024), 3),
        kernel_ms_at_collection=float(ms), source="profiles/r04_binned_%s_pmc_summary.txt" % shape, commit=commit)
    shutil.copy(src, os.path.join(ROOT, "profiles", "r04_binned_%s_pmc_summary.txt" % shape))
    print(shape, table["mini:%s:0.05:binned" % shape])
json.dump(table, open(path, "w"), indent=1, sort_keys=True)

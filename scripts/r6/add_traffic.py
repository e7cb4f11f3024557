#!/usr/bin/env python3
"""gpurun_out/r6/<shape>/{bench.json,pmc_summary.txt} + gpurun_out/r5_<shape>_kernel_stats.csv (scripts/r6/gpu_round6.sh)
-> profiles/pmc_traffic.json keys "mini:<shape>:0.05:binned" (the scan kernel over a binned resident read set: the dominant kernel of the headline step) and the committed copies
profiles/r06_<shape>_{pmc_summary.txt,kernel_stats.csv}.  The first scan of a read set against a table is TWO launches (the probe
of 8192 tiles, then the rest), and since round 6 every headline step bins the records anew -- a new read set, hence a probe
per step: 7 headline scans (probe + rest each) + 7 scans of the resident binned set (one probe) = 14 scans in 22 launches.  The
per-launch means of the counter passes mix probes and full launches, so the figures of a FULL launch are taken from the `max`
column (the full launches agree within 0.1 %)."""
import json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
table = json.load(open(path))
seen = table["_calibration"]["seen_fraction"]
commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()
n_tiles = -(-20_000_000 * 152 // 992)            # binned records are padded to 8 bytes: 152 per read
streamed = n_tiles * 1024
for shape in ("sampled", "contiguous"):
    src = os.path.join(ROOT, "gpurun_out", "r6", shape)
    ms = json.load(open(os.path.join(src, "bench.json")))["roofline"]["kernel_ms"]
    v = {}
    for ln in open(os.path.join(src, "pmc_summary.txt")):
        m = re.search(r"(\S+)\s+launches=\d+ mean=\S+ max=(\S+)", ln)
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
        kernel_ms_at_collection=float(ms), source="profiles/r06_%s_pmc_summary.txt" % shape, commit=commit,
        note="figures of a FULL launch (the `max` column: the first of the 7 scans is a probe launch + the rest)")
    shutil.copy(os.path.join(src, "pmc_summary.txt"), os.path.join(ROOT, "profiles", "r06_%s_pmc_summary.txt" % shape))
    ks = open(os.path.join(ROOT, "gpurun_out", "r6_%s_kernel_stats.csv" % shape)).read().splitlines()
    first = ks[1].split(",")
    total, mn = float(first[-4]), float(first[-2])
    with open(os.path.join(ROOT, "profiles", "r06_%s_kernel_stats.csv" % shape), "w") as f:
        f.write("\n".join(ks[:14]) + "\n")
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --db-shape %s --no-cpu-baseline --no-phases --no-config3 --no-cli-e2e --no-file-order --steps 5 --warmup 2\n" % shape)
        calls = int(first[-5])
        scans = 14
        f.write("# scan_mini_kernel: %d calls = %d scans: 7 headline steps (each bins the records anew: a probe of 8192 tiles = min_ms, then the rest) + 7 steps over the resident binned set (one probe).\n" % (calls, scans))
        f.write("# per scan: total_ms / %d = %.3f ms under the profiler; bench.py's HIP events around a headline step's scan (probe + host answer + rest): %.3f ms (roofline.kernel_ms)\n" % (scans, total / scans, ms))
        f.write("# count_fixed_kernel + place_fixed_kernel (+ the prefix kernels): the binning inside every headline step (ss_reorder.hip, records of one length)\n")
    print(shape, table["mini:%s:0.05:binned" % shape], "per scan under profiler %.3f" % (total / 14))
json.dump(table, open(path, "w"), indent=1, sort_keys=True)

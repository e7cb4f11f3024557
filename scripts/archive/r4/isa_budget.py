#!/usr/bin/env python3
"""Static instruction counts of scan_mini_kernel per phase, from the SSMARK labels of ss_mini.hip, per instantiation; with a
trip-count file (scripts/r4/scan_trips.py on the GPU: tiles, runs, lookups, found runs per launch) the loop bodies are weighted
and the sum is set beside the measured SQ_INSTS_VALU per tile.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -S --cuda-device-only strainscan_amd/csrc/ss_mini.hip -o /tmp/ss_mini.s
    python scripts/r4/isa_budget.py /tmp/ss_mini.s <instantiation substring> [lookup_rounds_per_tile candidate_rounds_per_tile bloom_rounds]
An instruction belongs to the label that precedes it in the file (the compiler keeps the order of the volatile asm labels;
blocks it moves elsewhere -- cold paths -- are counted where they land)."""
import collections
import re
import sys

NAMES = {-1: "kernel prologue", 20: "0 load + encode (+ next tile's addresses)", 0: "1a m-mer keys", 1: "1b minimizers, run starts, index bytes",
         6: "1b merge across lanes, prefix sum, walk -> q1", 2: "lambda setup", 7: "2 before the loops", 11: "2a Bloom round (body)",
         12: "2b lookup round: run -> hash -> page load (body)", 13: "2b lookup round: tags, inline k-mers, found runs (body)",
         14: "2b round end", 3: "3 before the loop (combiner: claim)", 10: "3 combiner begin", 9: "3 combiner after claim", 8: "3 combiner after flush",
         15: "3 candidate round (body)", 16: "3 round end + overflow runs", 4: "end of tile barrier", 5: "loop back / epilogue"}
LOOP = {11: "bloom", 12: "lookup", 13: "lookup", 14: "lookup", 15: "cand", 16: None}


def main():
    src = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2]
    trips = dict(lookup=float(sys.argv[3]), cand=float(sys.argv[4]), bloom=float(sys.argv[5])) if len(sys.argv) > 5 else None
    cur, region = None, -1
    acc, order = collections.defaultdict(collections.Counter), []
    for ln in src:
        m = re.match(r"^(_ZN\S*scan_mini_kernel\S*):", ln)
        if m:
            cur = m.group(1) if want in m.group(1) else None
            region = -1
            continue
        if cur is None:
            continue
        if ln.strip().startswith(".Lfunc_end"):
            break
        mm = re.search(r"; SSMARK (\d+)", ln)
        if mm:
            region = int(mm.group(1))
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        cls = ("VALU" if op.startswith("v_") else "ctl" if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch", "s_endpgm")) else
               "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        acc[region][cls] += 1
        if region not in order:
            order.append(region)
    print("| phase (label) | VALU | SALU | LDS | VMEM | x per tile | VALU per tile |")
    print("|---|---|---|---|---|---|---|")
    tot = 0.0
    for r in order:
        c = acc[r]
        mult = 1.0
        if trips and LOOP.get(r):
            mult = trips[LOOP[r]]
        if r == -1:
            mult = 0.0
        tot += c["VALU"] * mult
        print("| %s | %d | %d | %d | %d | %s | %.0f |" % (NAMES.get(r, str(r)), c["VALU"], c["SALU"], c["LDS"], c["VMEM"], ("%.2f" % mult) if trips else "", c["VALU"] * mult))
    print("| **sum** | | | | | | **%.0f** |" % tot)


main()

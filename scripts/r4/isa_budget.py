#!/usr/bin/env python3
"""Static instruction counts of scan_mini_kernel between the SSMARK phase markers of ss_mini.hip, per instantiation.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -S --cuda-device-only strainscan_amd/csrc/ss_mini.hip -o /tmp/ss_mini.s
    python scripts/r4/isa_budget.py /tmp/ss_mini.s [instantiation substring ...]
Regions are named by the marker that ENDS them (SS_T(i) closes phase i).  Loop bodies are counted once: the dynamic totals
(PMC SQ_INSTS_*) divided by the tiles of a launch give the trip-weighted sums to set beside them."""
import collections
import re
import sys

names = {0: "0 load+encode", 1: "1a m-mer keys", 6: "1b minimizers+run starts", 2: "1b merge+walk->q1", 7: "(setup)", 3: "2 bloom+pages",
         10: "3 comb: begin", 8: "3 comb: flush", 9: "3 comb: claim", 4: "3 candidates+overflow", 5: "end barrier", -1: "prologue", 99: "epilogue"}
src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2:]
cur_k = None
region = -1
acc = {}
order = {}
for ln in src:
    m = re.match(r"^(_ZN\S*scan_mini_kernel\S*):", ln)
    if m:
        cur_k = m.group(1)
        region = -1
        acc[cur_k] = collections.defaultdict(lambda: collections.Counter())
        order[cur_k] = []
        continue
    if cur_k is None:
        continue
    if ln.startswith("\t.end_amdhsa_kernel") or ln.strip().startswith(".Lfunc_end"):
        cur_k = None
        continue
    mm = re.search(r"; SSMARK (\d+)", ln)
    if mm:
        region = int(mm.group(1)) + 1000      # instructions after marker i belong to the NEXT phase: resolved below
        continue
    t = ln.strip()
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        cls = "VALU"
    elif op.startswith("s_"):
        cls = "SALU" if not op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch", "s_endpgm")) else "ctl"
    elif op.startswith("ds_"):
        cls = "LDS"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cls = "VMEM"
    else:
        cls = "other"
    acc[cur_k][region][cls] += 1
    if region not in order[cur_k]:
        order[cur_k].append(region)
# marker i ENDS phase i: instructions after marker a and before marker b belong to b.  Code is laid out mostly in program
# order, so region "after marker a" is labelled by the next marker in the source order of SS_T calls.
nxt = {-1: 0, 1000: 1, 1001: 6, 1006: 2, 1002: 7, 1007: 3, 1003: 10, 1010: 8, 1008: 9, 1009: 4, 1004: 5, 1005: 99}
for k, regs in acc.items():
    short = re.sub(r"^.*scan_mini_kernelI", "", k)[:16]
    if want and not any(w in k for w in want):
        continue
    print("== %s  (ALIGNED, BLOOM, COMB, waves/SIMD = %s)" % (short, short))
    tot = collections.Counter()
    print("%-28s %6s %6s %5s %5s %5s" % ("phase (static, loops once)", "VALU", "SALU", "LDS", "VMEM", "ctl"))
    for r in order[k]:
        c = regs[r]
        ph = nxt.get(r, r)
        print("%-28s %6d %6d %5d %5d %5d" % (names.get(ph, str(ph)), c["VALU"], c["SALU"], c["LDS"], c["VMEM"], c["ctl"]))
        tot.update(c)
    print("%-28s %6d %6d %5d %5d %5d" % ("total", tot["VALU"], tot["SALU"], tot["LDS"], tot["VMEM"], tot["ctl"]))

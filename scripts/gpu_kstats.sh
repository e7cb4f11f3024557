#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command; this package's kernels (and rocprim's) -> gpurun_out/<name>_kernel_stats.csv
# usage: gpu_kstats.sh NAME python3-script [args...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
name=$1; shift
export TMPDIR=/tmp; cd /tmp
rm -rf $O/kst_$name
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kst_$name -o k -- python3 "$@" > /dev/null 2> $O/kst_$name.err
f=$(find $O/kst_$name -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" > $O/${name}_kernel_stats.csv <<'PY'
import csv, sys
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "total_ms", "average_ms", "min_ms", "max_ms"])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "at::native" in n or "elementwise" in n or "rocclr" in n:
        continue
    short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
    if "rocprim" in short:
        short = "rocprim::" + short.split("detail::")[-1].split("<")[0] + ("<...radix_sort...>" if "radix_sort" in n else "<...scan...>" if "scan" in n else "")
    w.writerow([short[:70], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 4), round(float(r["AverageNs"]) / 1e6, 4), round(float(r["MinNs"]) / 1e6, 4), round(float(r["MaxNs"]) / 1e6, 4)])
PY
rm -rf $O/kst_$name
head -30 $O/${name}_kernel_stats.csv

#!/bin/bash
# kernel durations of the device inflater (rocprofv3 --kernel-trace --stats) on a 1 M-read FASTQ .gz
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/gi_stats; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o gi -- python3 $R/scripts/dev/t_ginf_prof.py ${1:-1000000} > $O/run.out 2> $O/run.err
f=$(find $O -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %4s  total %10.3f ms  avg %9.3f ms" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
tail -3 $O/run.out

#!/bin/bash
# kernel durations of the device inflater for the library given as $1 (a path) on a 1 M-read FASTQ .gz; $2 = tag
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/gi_stats_$2; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
export SS_LIB=$1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o gi -- python3 $R/scripts/dev/t_ginf_prof.py 1000000 > $O/run.out 2> $O/run.err
python3 - "$O" "$2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "inflate_kernel" in r["Name"] or "sync_kernel" in r["Name"]:
        print(sys.argv[2], r["Name"].split("::")[-1][:14], "avg %.3f ms" % (float(r["AverageNs"]) / 1e6))
PY
tail -1 $O/run.out

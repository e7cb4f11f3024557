#!/bin/bash
# kernel durations for a given library (SS_LIB): rocprofv3 --kernel-trace --stats on a 1 M-read FASTQ .gz
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/gi_stats_$1; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o gi -- python3 $R/scripts/dev/t_ginf_prof.py 1000000 > $O/run.out 2> $O/run.err
f=$(find $O -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -E "inflate_kernel|sync_kernel" $f | awk -F, '{print $1, "avg ms", $4/1e6}' | sed 's/(anonymous namespace):://' | cut -c1-90

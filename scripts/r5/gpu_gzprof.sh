#!/bin/bash
# kernel statistics of a large .fastq.gz pair's load (rocprofv3 --kernel-trace --stats)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5_gzprof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export SS_GZ_INPROCESS=1
rocprofv3 --kernel-trace --stats -d $O/prof -o gz -- python3 $R/scripts/r5/gz_scale.py ${1:-10000000} 1 3 > $O/run.log 2>&1
tail -4 $O/run.log
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo $f; head -25 $f | cut -c1-200
cp $f $O/kernel_stats.csv

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_comb2; mkdir -p $O; cd $R
SS_LIB=$R/build_tmp/lib_cstats.so timeout 600 python scripts/dev/t_hit_heavy.py 5000000 20000000 2>&1 | tail -1 | tee $O/hit_heavy_stats.txt

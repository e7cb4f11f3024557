#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_l2files; mkdir -p $O; cd $R
timeout 900 python scripts/bench_l2_files.py 1000000 200 2>&1 | tail -3 | tee $O/l2_files.txt

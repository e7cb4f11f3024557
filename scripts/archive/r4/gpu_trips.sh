#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_trips; mkdir -p $O; cd $R
for shape in sampled contiguous; do
  SS_LIB=$R/build_tmp/lib_cstats.so timeout 600 python scripts/r4/scan_trips.py $shape 2>/dev/null | tail -1 | tee $O/trips_$shape.json
done

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_atom; mkdir -p $O; cd $R
timeout 300 ./build_tmp/atomics 40 2>&1 | tee $O/atomics_40MB.txt
timeout 300 ./build_tmp/atomics 2 2>&1 | tee $O/atomics_2MB.txt
for p in 0 1; do
  SS_PRIVATE_COUNTS=$p timeout 600 python scripts/dev/t_hit_heavy.py 5000000 20000000 2>&1 | tail -1 | tee $O/hit_heavy_priv$p.txt
done

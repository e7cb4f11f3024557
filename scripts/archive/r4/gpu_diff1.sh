#!/bin/bash
# round 4, step 1: difference counters -- parity, then the hit-heavy cluster scan with and without them
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_diff1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_scan_gpu.py -x -q -k "difference or random_vs or low_complexity or queue_overflow or edge or shard_lin or harvest or sampled_database" 2>&1 | tail -8 | tee $O/pytest.log
for d in 0 1; do
  SS_DIFF_COUNTS=$d timeout 600 python scripts/dev/t_hit_heavy.py 5000000 20000000 2>&1 | tail -1 | tee $O/hit_heavy_diff$d.txt
done

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_comb1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_scan_gpu.py -x -q -k "combined or random_vs or low_complexity or queue_overflow or edge or shard_lin or harvest or sampled_database or locality" 2>&1 | tail -8 | tee $O/pytest.log
SS_EXPECT_HITS=0 timeout 600 python scripts/dev/t_hit_heavy.py 5000000 20000000 2>&1 | tail -1 | tee $O/hit_heavy_base.txt
for lb in 6 8 4; do
  SS_MINI_LB=$lb timeout 600 python scripts/dev/t_hit_heavy.py 5000000 20000000 2>&1 | tail -1 | tee $O/hit_heavy_comb_lb$lb.txt
done
SS_COMBINE=1 timeout 600 python scripts/dev/t_hit_heavy.py 5000000 20000000 2>&1 | tail -1 | tee $O/hit_heavy_comb_always.txt

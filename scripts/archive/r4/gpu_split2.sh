#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_split; mkdir -p $O; cd $R
for rnd in 1 2 3; do for pool in 1 0; do
SS_SPLIT_POOL=$pool SS_SPLIT_TRACE=1 timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/split2.txt
import sys, time, os
sys.path.insert(0, '.')
from strainscan_amd import l2
l2.shuffle_split_test_bits(10)
for it in range(3):
    t = time.perf_counter(); l2.shuffle_split_test_bits(5_000_000); print("pool", os.environ["SS_SPLIT_POOL"], "call", it, round((time.perf_counter() - t) * 1e3, 1), "ms")
PY
done; done

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_split; mkdir -p $O; cd $R
SS_SPLIT_TRACE=1 timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/split.txt
import sys, json, time, os
sys.path.insert(0, '.')
import numpy as np
from strainscan_amd import l2
l2.shuffle_split_test_bits(10)
a, _ = l2.shuffle_split_test_bits(400_003); b, _ = l2.shuffle_split_test_bits_numpy(400_003); print("equal to numpy:", bool(np.array_equal(a, b)))
for rnd in range(3):
    for blk in ("1", "0"):
        os.environ["SS_SPLIT_BLOCKED"] = blk
        t = time.perf_counter(); l2.shuffle_split_test_bits(5_000_000); print("blocked", blk, round((time.perf_counter() - t) * 1e3, 1), "ms")
os.environ.pop("SS_SPLIT_BLOCKED")

PY

#!/bin/bash
# ShuffleSplit on the box's host: alone, and in a process that has initialised torch + HIP (what the bench's l2_solve block sees)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_split; mkdir -p $O; cd $R
echo "--- alone"; SS_SPLIT_TRACE=1 timeout 100 python3 scripts/dev/t_split_host.py 2>&1 | grep shuffle | tail -4 | cut -c1-330
echo "--- with torch + HIP initialised"
SS_SPLIT_TRACE=1 timeout 300 python3 - <<'PY' 2>&1 | grep shuffle | tail -6 | cut -c1-330
import sys, time
sys.path.insert(0, '.')
import torch
torch.zeros(10, device="cuda").sum().item()
from strainscan_amd import l2, _lib
_lib.lib()
for it in range(6):
    l2.shuffle_split_test_bits(5_000_000)
PY
echo "--- numactl"; numactl -H 2>/dev/null | head -5; lscpu | grep -i numa

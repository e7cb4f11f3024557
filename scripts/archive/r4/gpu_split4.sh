#!/bin/bash
# ShuffleSplit 5 M x 20 in a process with torch + HIP initialised: wall time by the number of swap workers in flight
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for fl in 8 10 12 14; do
SS_SPLIT_IN_FLIGHT=$fl timeout 300 python3 - <<'PY' 2>&1 | grep -v amdgpu
import sys, time, os
sys.path.insert(0, '.')
import torch
torch.zeros(10, device="cuda").sum().item()
from strainscan_amd import l2, _lib
ts = []
for it in range(24):
    t = time.perf_counter(); l2.shuffle_split_test_bits(5_000_000); ts.append((time.perf_counter() - t) * 1e3)
ts = sorted(ts[2:])
print("in flight", os.environ["SS_SPLIT_IN_FLIGHT"], "min %.1f q1 %.1f median %.1f q3 %.1f max %.1f ms" % (ts[0], ts[len(ts) // 4], ts[len(ts) // 2], ts[3 * len(ts) // 4], ts[-1]))
PY
done

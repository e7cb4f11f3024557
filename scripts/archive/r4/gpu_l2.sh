#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_l2; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests/test_l2_gpu.py tests/test_abi_and_host.py -x -q 2>&1 | tail -5 | tee $O/pytest.log
SS_SPLIT_TRACE=1 timeout 600 python - <<'PY' 2>&1 | tee $O/l2_block.txt
import sys, json, time
sys.path.insert(0, '.')
import bench, torch, numpy as np
from strainscan_amd import l2
l2.shuffle_split_test_bits(10)
for n in (1_000_000, 5_000_000):
    for fl in (None, 8, 14, 20):
        import os
        if fl: os.environ["SS_SPLIT_IN_FLIGHT"] = str(fl)
        t = time.perf_counter(); l2.shuffle_split_test_bits(n); print(n, fl, round((time.perf_counter() - t) * 1e3, 1), "ms")
os.environ.pop("SS_SPLIT_IN_FLIGHT", None)
args = bench.parse_args(["--reads", "200000", "--cluster-genome", "200000"])
dev = torch.device("cuda", 0)
out = bench.measure_config3(torch, dev, args, torch.cuda.current_stream().cuda_stream)
print(json.dumps(out["l2_solve"], indent=1))
PY

#!/bin/bash
# the bench's l2_solve block alone (5 M x 300)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_l2; mkdir -p $O; cd $R
SS_SPLIT_TRACE=1 timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu | cut -c1-250 | tee $O/l2_block2.txt
import sys, json
sys.path.insert(0, '.')
import bench, torch
args = bench.parse_args(["--reads", "200000", "--cluster-genome", "200000"])
dev = torch.device("cuda", 0)
out = bench.measure_config3(torch, dev, args, torch.cuda.current_stream().cuda_stream)
print(json.dumps(out["l2_solve"], indent=1))
PY

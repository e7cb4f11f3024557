#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_full; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests/test_scan_gpu.py tests/test_l2_gpu.py -x -q -k "full_size or config3_size" --durations=8 2>&1 | tail -16 | tee $O/pytest.log

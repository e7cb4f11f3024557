#!/bin/bash
# gpu_t.sh "<pytest -k expression>" [files...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_t; mkdir -p $O; cd $R
K="$1"; shift
timeout 2400 python -m pytest ${@:-tests} -m gpu -x -q -k "$K" 2>&1 | tail -12 | tee $O/pytest.log

#!/bin/bash
# round-4 evidence: smoke, then bench line + rocprof kernel stats + PMC passes for both database shapes (scripts/gpu_round2.sh)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $R/gpurun_out/smoke.log
TAG=sampled CALIB=1 BENCH_ARGS="--db-shape sampled" bash scripts/gpu_round2.sh 2>&1 | tail -30
TAG=contiguous BENCH_ARGS="--db-shape contiguous" bash scripts/gpu_round2.sh 2>&1 | tail -30

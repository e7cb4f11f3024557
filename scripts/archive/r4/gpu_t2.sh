#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_t; mkdir -p $O; cd $R
K="$1"; shift
SS_INGEST_TRACE=1 timeout 900 python -m pytest ${@:-tests} -m gpu -x -q -k "$K" > $O/pytest_full.log 2>&1
tail -5 $O/pytest_full.log

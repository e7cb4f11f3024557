#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_inl; mkdir -p $O; cd $R
for im in 2 0 1; do
  echo "== SS_INLINE_MAX=$im" | tee -a $O/inl.txt
  SS_INLINE_MAX=$im timeout 600 python scripts/dev/t_hit_heavy.py 5000000 20000000 2>&1 | tail -1 | tee -a $O/inl.txt
done

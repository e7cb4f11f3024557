#!/bin/bash
# per-phase cycle shares of scan_mini_kernel from the -DSS_TIMING variant (scripts/build_variant.sh timing "-DSS_TIMING")
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for hf in ${HF:-0.05}; do
  echo "== hit_frac $hf"
  SS_LIB=$R/build_tmp/lib_${V:-timing}.so timeout 600 python scripts/phase_timing.py $hf 2>&1 | grep -v amdgpu.ids
done

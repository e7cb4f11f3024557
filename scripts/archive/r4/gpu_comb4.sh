#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_comb4; mkdir -p $O; cd $R
for v in noflush nolds noclaim nolds_noflush none3; do
  echo "== $v" | tee -a $O/variants.txt
  SS_LIB=$R/build_tmp/lib_$v.so timeout 600 python scripts/dev/t_hit_heavy.py 5000000 20000000 2>&1 | tail -1 | tee -a $O/variants.txt
done

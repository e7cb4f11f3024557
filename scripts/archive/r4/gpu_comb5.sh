#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_comb5; mkdir -p $O; cd $R
for v in c_noinl noatom noinl; do
  echo "== $v (expect hits)" | tee -a $O/variants.txt
  SS_LIB=$R/build_tmp/lib_$v.so timeout 600 python scripts/dev/t_hit_heavy.py 5000000 20000000 2>&1 | tail -1 | tee -a $O/variants.txt
done
echo "== noatom, no hint" | tee -a $O/variants.txt
SS_EXPECT_HITS=0 SS_LIB=$R/build_tmp/lib_noatom.so timeout 600 python scripts/dev/t_hit_heavy.py 5000000 20000000 2>&1 | tail -1 | tee -a $O/variants.txt
echo "== noatom, no hint, no bloom" | tee -a $O/variants.txt
SS_BLOOM_BITS=0 SS_EXPECT_HITS=0 SS_LIB=$R/build_tmp/lib_noatom.so timeout 600 python scripts/dev/t_hit_heavy.py 5000000 20000000 2>&1 | tail -1 | tee -a $O/variants.txt
python - <<'PY' | tee -a $O/variants.txt
import numpy as np, torch, sys
sys.path.insert(0, '.')
from strainscan_amd import _lib
PY

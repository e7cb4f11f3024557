#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_inl; mkdir -p $O; cd $R
for round in 1 2; do for im in 2 0; do
  SS_INLINE_MAX=$im timeout 600 python bench.py --no-cpu-baseline --no-phases --no-config3 --db-shape contiguous 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inline_max $im', d['roofline']['kernel_ms'], d['resident_read_set']['kernel_ms'], d['config']['index'])" | tee -a $O/inl_contig.txt
done; done

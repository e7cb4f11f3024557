#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_prep; mkdir -p $O; cd $R
for shape in sampled contiguous sampled; do
timeout 600 python bench.py --no-cpu-baseline --no-phases --no-config3 --db-shape $shape 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['resident_read_set']; print('$shape', r['prepare_ms'], r['prepare_ms_all'], r['prepare_breakdown_ms'])" | tee -a $O/prep.txt
done

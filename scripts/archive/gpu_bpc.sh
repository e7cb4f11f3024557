#!/bin/bash
# sweep the persistent grid size of scan_mini_kernel: BPCS="4 5 6" blocks per CU, LBS launch bounds
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for lb in ${LBS:-4}; do for bpc in ${BPCS:-5 8}; do
  SS_MINI_LB=$lb SS_MINI_BLOCKS_PER_CU=$bpc timeout 600 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --hit-frac ${HF:-0.05} 2>$O/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LB=$lb BPC=$bpc', d['value'], 'M reads/s kernel_ms', d['roofline']['kernel_ms'], d['check']['total_hits'])" || tail -3 $O/ab.err
done; done

#!/bin/bash
# A/B the mini kernel tuning knobs: CFGS="LB:BPC:hitfrac ..." (space separated)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
[ "${SKIP_TESTS:-0}" = 1 ] || timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
CFGS=${CFGS:-"4:0:0.05 3:0:0.05 4:0:0.002 3:0:0.002"}
for cfg in $CFGS; do
  lb=$(echo $cfg | cut -d: -f1); bpc=$(echo $cfg | cut -d: -f2); hf=$(echo $cfg | cut -d: -f3)
  SS_MINI_LB=$lb SS_MINI_BLOCKS_PER_CU=$bpc timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --hit-frac $hf 2>$O/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LB=$lb BPC=$bpc hit=$hf', d['value'], 'M reads/s kernel_ms', d['roofline']['kernel_ms'], d['check']['total_hits'])" || tail -3 $O/ab.err
done

#!/bin/bash
# A/B library variants built by scripts/build_variant.sh: LIBS="base pf2 ..." (base = the in-tree library),
# HF="0.05 0.5" hit fractions, LB = launch bound to use
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for hf in ${HF:-0.05}; do
for v in ${LIBS:-base}; do
  lib=$R/build_tmp/lib_$v.so; [ "$v" = base ] && lib=$R/strainscan_amd/lib/libstrainscan_hip.so
  for lb in ${LBS:-5}; do
  SS_LIB=$lib SS_MINI_LB=$lb timeout 600 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --hit-frac $hf 2>$O/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$v LB=$lb hit=$hf', d['value'], 'M reads/s kernel_ms', d['roofline']['kernel_ms'], d['check']['total_hits'])" || tail -3 $O/ab.err
  done
done
done

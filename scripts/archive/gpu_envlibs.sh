#!/bin/bash
# A/B of (library variant, environment) pairs in one box: RUNS="lib|ENV=..;lib|ENV=.." (lib: base or a build_tmp variant)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
IFS=';' read -ra RUNS_ <<< "${RUNS}"
for hf in ${HF:-0.05}; do
for r in "${RUNS_[@]}"; do
  v=${r%%|*}; e=${r#*|}
  lib=$R/build_tmp/lib_$v.so; [ "$v" = base ] && lib=$R/strainscan_amd/lib/libstrainscan_hip.so
  env SS_LIB=$lib $e timeout 600 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --hit-frac $hf 2>$O/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$r] hit=$hf', d['value'], 'M reads/s kernel_ms', d['roofline']['kernel_ms'], d['check']['total_hits'])" || tail -3 $O/ab.err
done; done

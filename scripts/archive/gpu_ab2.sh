#!/bin/bash
# A/B of scan-kernel variants: RUNS="lib:ENV=val,ENV=val:bench args;..." (lib = base or a build_tmp/lib_<name>.so)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
IFS=';' read -ra RR <<< "${RUNS}"
for r in "${RR[@]}"; do
  lib=$(echo "$r" | cut -d: -f1); envs=$(echo "$r" | cut -d: -f2 | tr ',' ' '); args=$(echo "$r" | cut -d: -f3)
  so=$R/build_tmp/lib_$lib.so; [ "$lib" = base ] && so=$R/strainscan_amd/lib/libstrainscan_hip.so
  ( export SS_LIB=$so; for e in $envs; do export $e; done
    timeout 600 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-phases $args 2>$O/ab2.err | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('%-10s %-40s %-40s kernel_ms %.3f step_ms %.3f hits %d' % ('$lib', '$envs', '$args', d['roofline']['kernel_ms'], d['ms_per_step'], d['check']['total_hits']))
except Exception as e:
    print('$lib $envs $args FAILED', e)
" || tail -3 $O/ab2.err )
done

#!/bin/bash
# bench.py under a list of environment settings: ENVS="A=1 B=2;A=3" (';' separates runs)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
IFS=';' read -ra RUNS <<< "${ENVS}"
for hf in ${HF:-0.05}; do
for e in "${RUNS[@]}"; do
  env $e timeout 600 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --hit-frac $hf 2>$O/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$e] hit=$hf', d['value'], 'M reads/s kernel_ms', d['roofline']['kernel_ms'], d['check']['total_hits'])" || tail -3 $O/ab.err
done; done

#!/bin/bash
# quick GPU check: parity tests (-x), then bench with each table layout
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
for lay in ${LAYOUTS:-mini flat}; do
  SS_LAYOUT=$lay timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> $O/bench_$lay.err | tee $O/bench_$lay.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lay', d['value'], 'M reads/s', d['roofline'], d['check'])"
  tail -2 $O/bench_$lay.err
done

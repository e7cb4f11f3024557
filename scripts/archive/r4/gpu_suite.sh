#!/bin/bash
# the whole -m gpu suite, then the bench line for both database shapes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${NAME:-r4_suite}; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
for shape in sampled contiguous; do
  timeout 900 python bench.py --db-shape $shape ${BENCH_ARGS:-} 2> $O/bench_$shape.err | tee $O/bench_$shape.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$shape', d['value'], d['ms_per_step'], d['roofline'], d.get('resident_read_set'))"
  tail -2 $O/bench_$shape.err
done

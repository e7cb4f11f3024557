#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r2b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_scan_gpu.py -x -q -k "harvest or sampled or tinyq or queue" 2>&1 | tail -5
for sh in sampled contiguous; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --db-shape $sh > $O/$sh.json 2> $O/$sh.err; tail -2 $O/$sh.err | cut -c1-200
  python -c "
import json; d=json.load(open('$O/$sh.json')); print('$sh', d['value'], d['ms_per_step'], d['step_breakdown_ms'], d['check'])"
done
SS_BENCH_FORCE_EXCHANGE=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/xchg.json 2> $O/xchg.err; tail -2 $O/xchg.err | cut -c1-300
python -c "
import json; d=json.load(open('$O/xchg.json')); print('exchange(self group)', d['value'], d['ms_per_step'], d['step_breakdown_ms'], d['check'])"

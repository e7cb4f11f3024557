#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_bench1; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_scan_gpu.py -x -q -k "bench_line" 2>&1 | tail -5 | tee $O/pytest.log
( time timeout 900 python bench.py 2> $O/bench.err | tee $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['cluster_scan'], indent=1)); print(json.dumps(d['l2_solve'], indent=1)); print(d['value'], d['resident_read_set']['kernel_ms'])" ) 2>&1 | tee $O/summary.txt
tail -3 $O/bench.err

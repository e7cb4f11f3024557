#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_multi; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests/test_scan_gpu.py tests/test_l2_gpu.py -x -q -k "multi or three_clusters or end_to_end or combined or bench_line or full_size" 2>&1 | tail -8 | tee $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-phases 2> $O/bench.err | tee $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['cluster_scan'], indent=1)); print(json.dumps(d['l2_solve']['phases_ms'])); print(d['l2_solve']['wall_ms_all']); print(d['value'], d['roofline']['kernel_ms'], d['resident_read_set']['kernel_ms'])" 2>&1 | tee $O/summary.txt
timeout 600 python bench.py --no-cpu-baseline --no-phases --no-config3 --db-shape contiguous 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('contiguous', d['value'], d['roofline']['kernel_ms'], d['resident_read_set']['kernel_ms'])" 2>&1 | tee -a $O/summary.txt

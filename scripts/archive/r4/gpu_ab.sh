#!/bin/bash
# quick A/B of the scan: the scan tests, then the bench lines (kernel times of both shapes, both orders, cluster scan)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${NAME:-r4_ab}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_scan_gpu.py -m gpu -x -q -k "${TESTS:-random_vs or low_complexity or queue_overflow or combined or multi or sampled_database or harvest or locality}" 2>&1 | tail -3 | tee $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-phases 2> $O/bench.err | tee $O/bench_sampled.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['cluster_scan']; print('sampled', d['roofline']['kernel_ms'], d['resident_read_set']['kernel_ms'], 'cluster', c['file_order']['kernel_ms'], c['binned']['kernel_ms'], c['three_tables']['one_pass_ms'], 'l2', d['l2_solve']['wall_ms'])" 2>&1 | tee $O/summary.txt
timeout 600 python bench.py --no-cpu-baseline --no-phases --no-config3 --db-shape contiguous 2>> $O/bench.err | tee $O/bench_contiguous.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('contiguous', d['roofline']['kernel_ms'], d['resident_read_set']['kernel_ms'])" 2>&1 | tee -a $O/summary.txt

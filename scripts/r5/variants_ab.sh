#!/bin/bash
# headline (binned) kernel time of the default build and of compile-flag variants of ss_mini.hip, both database shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for lib in "" rpl2 u3_2 u3_8; do for shape in sampled contiguous; do
  L=""; [ -n "$lib" ] && L=$R/build_tmp/lib_$lib.so
  SS_LIB=$L python bench.py --db-shape $shape --no-cpu-baseline --no-phases --no-config3 --no-cli-e2e --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib=[$lib] $shape binned', d['roofline']['kernel_ms'], 'file', d['file_order']['roofline']['kernel_ms'], d['file_order']['node_stats_equal'], d['check']['harvest_equals_gather'])"
done; done

#!/bin/bash
# same box, two libraries: bench kernel times (file order / binned) of one shape, alternating, three rounds
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_ablibs; mkdir -p $O; cd $R
: > $O/ab.txt
for round in 1 2 3; do
  for lib in ${LIBS:-presolid head}; do
    L=$R/build_tmp/lib_$lib.so; [ "$lib" = head ] && L=$R/strainscan_amd/lib/libstrainscan_hip.so
    SS_LIB=$L timeout 600 python bench.py --no-cpu-baseline --no-phases --no-config3 --db-shape ${SHAPE:-contiguous} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['roofline']['kernel_ms'], d['resident_read_set']['kernel_ms'])" | tee -a $O/ab.txt
  done
done

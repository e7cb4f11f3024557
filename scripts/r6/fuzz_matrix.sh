#!/bin/bash
# The whole-flow campaign (scripts/r6/fuzz_product.py DIR flow [WORLD]) once per alternative code path: every environment variable of
# INTEGRATION.md F that selects a PATH (not a resource) and is reachable from the command line, alone and under 3 ranks.
#   fuzz_matrix.sh DIR [second]   -> gpurun_out/fuzz_matrix.log (one summary line per setting; `second`: the settings added after the first run)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; D=${1:-tests/golden/_fuzz_tmp}; O=$R/gpurun_out; mkdir -p $O; : > $O/fuzz_matrix.log
run() {   # name, world, env...
  name=$1; world=$2; shift 2
  env "$@" SS_FUZZ_VERBOSE=90 timeout 900 python scripts/r6/fuzz_product.py $D flow $world > $O/fuzz_matrix_$name.log 2>&1
  echo "$name (world $world; $*): rc=$? $(grep -c DISAGREES $O/fuzz_matrix_$name.log) disagreements; $(grep '^fuzz_product' $O/fuzz_matrix_$name.log)" | tee -a $O/fuzz_matrix.log
  grep "DISAGREES\|Timeout" $O/fuzz_matrix_$name.log | head -5 | cut -c1-400 | tee -a $O/fuzz_matrix.log
}
if [ "${2:-all}" = "second" ]; then
run no_cache       1 SS_IMAGE_CACHE=off
run no_cache_w3    3 SS_IMAGE_CACHE=off
run serial_l2_w3   3 SS_L2_ONE_PASS=0 SS_L2_THREADS=1
run file_order_w3  3 SS_READS_ORDER=file
run flat_table_w3  3 SS_LAYOUT=flat
run host_build_w2  2 SS_BUILD=host
run cached_second_run    1 SS_FUZZ_TWICE=1
run cached_second_run_w3 3 SS_FUZZ_TWICE=1
run gz_small_units       1 SS_GZ_SLICE_KB=64 SS_GZ_SEG_KB=64 SS_GZ_CHUNK=4096 SS_GZ_SPLIT_KB=4
run gz_small_units_w3    3 SS_GZ_SLICE_KB=64 SS_GZ_SEG_KB=64 SS_GZ_CHUNK=4096 SS_GZ_SPLIT_KB=4
exit 0
fi
run streaming      1 SS_READS_RESIDENT_GB=0
run streaming_w3   3 SS_READS_RESIDENT_GB=0
run host_gz        1 SS_GZ_GPU=0
run host_gz_zlib   1 SS_GZ_GPU=0 SS_NO_PGZ=1 SS_NO_LIBDEFLATE=1
run host_gz_w3     3 SS_GZ_GPU=0
run file_order     1 SS_READS_ORDER=file
run serial_l2      1 SS_L2_ONE_PASS=0 SS_L2_THREADS=1
run flat_table     1 SS_LAYOUT=flat
run host_build     1 SS_BUILD=host
run combine0       1 SS_COMBINE=0
run combine1       1 SS_COMBINE=1
run seq_ingest     1 SS_INGEST=seq
run no_cache       1 SS_IMAGE_CACHE=off
run no_range_w3    3 SS_GZ_RANGE=0
run no_share_w3    3 SS_GZ_SHARE=0

#!/bin/bash
# build_reorder_variant.sh NAME "-DFLAGS": the library with other ss_reorder.hip compile flags, into build_tmp/ (A/B runs: SS_LIB=...)
set -e
R=$(cd $(dirname $0)/../.. && pwd); cd $R/strainscan_amd/csrc; mkdir -p $R/build_tmp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -Wno-unused-result -Wno-unused-value"
/opt/rocm/bin/hipcc $F $2 -c ss_reorder.hip -o $R/build_tmp/ss_reorder_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_tmp/lib_$1.so ss_scan.o ss_build_dev.o ss_ingest.o ss_mini.o ss_ginflate.o ss_fastq_dev.o ss_pgz.o ss_host.o ss_nodes.o ss_l2.o ss_enet.o $R/build_tmp/ss_reorder_$1.o -lz -lpthread -ldl
echo built build_tmp/lib_$1.so

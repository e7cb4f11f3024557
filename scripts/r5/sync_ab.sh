#!/bin/bash
# sync_kernel before / after (build_tmp/lib_old.so = the commit before): the same entries and candidates, kernel time of both
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
export SS_GZ_KEEP_DIR=/dev/shm/ss_sync_ab SS_GZ_INPROCESS=1
timeout 300 python scripts/r5/gz_scale.py 6000000 1 0 > gpurun_out/sync_ab.log 2>&1
export SS_GZ_REUSE=1
for v in old new old new; do
  L=""; [ $v = old ] && L=$R/build_tmp/lib_old.so
  echo "== $v" >> gpurun_out/sync_ab.log
  SS_LIB=$L SS_INGEST_TRACE=1 timeout 300 python scripts/r5/gz_scale.py 6000000 1 3 2>&1 | grep "candidate blocks\|entries inside\|^load\|sync  " >> gpurun_out/sync_ab.log
done
for v in old new; do
  L=""; [ $v = old ] && L=$R/build_tmp/lib_old.so
  SS_LIB=$L bash scripts/gpu_kstats.sh sync_$v $R/scripts/r5/gz_scale.py 6000000 1 3 2>&1 | grep "sync_kernel" | sed "s/^/$v /" >> gpurun_out/sync_ab.log
done
rm -rf /dev/shm/ss_sync_ab
cat gpurun_out/sync_ab.log | grep -v "^\[ginflate\] seg"

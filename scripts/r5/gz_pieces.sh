#!/bin/bash
# fresh-process loads of a 2 x 25 M-read .gz pair with the search of a large image in 1 / 4 / 8 / 16 pieces (ss_ginflate.hip)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
export SS_GZ_KEEP_DIR=/dev/shm/ss_gz_pieces
SS_GZ_FRESH=1 timeout 700 python scripts/r5/gz_scale.py 50000000 1 0 > gpurun_out/gz_pieces.log 2>&1
for p in 1 8 4 16 1 8; do
  echo "pieces $p" >> gpurun_out/gz_pieces.log
  SS_GZ_SEARCH_PIECES=$p SS_GZ_FRESH=1 SS_GZ_REUSE=1 timeout 300 python scripts/r5/gz_scale.py 50000000 1 3 2>&1 | grep "fresh process\|^load" >> gpurun_out/gz_pieces.log
done
rm -rf /dev/shm/ss_gz_pieces
grep -v "^\[" gpurun_out/gz_pieces.log

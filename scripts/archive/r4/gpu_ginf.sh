#!/bin/bash
# the device inflater after a change: its tests, 90 s of differential fuzz, kernel durations on a 1 M-read .gz, the dist gz tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_ginf; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ginflate_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
timeout 300 python scripts/dev/fuzz_ginflate.py ${FUZZ_S:-90} ${FUZZ_SEED:-7000} 2>&1 | tail -3 | tee $O/fuzz.txt
bash scripts/dev/gpu_ginf_stats.sh 1000000 2>&1 | grep -i "inflate_kernel\|sync_kernel\|subsync\|bytes_kernel\|windows\|tails\|crc" | tee $O/kernels.txt
timeout 900 python -m pytest tests/test_dist_gpu.py -m gpu -x -q -k "gz" 2>&1 | tail -3 | tee -a $O/pytest.log

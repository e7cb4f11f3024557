#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_fuzz; mkdir -p $O; cd $R
timeout 420 python scripts/dev/fuzz_ginflate.py 300 ${SEED:-9000} 2>&1 | tail -2 | tee $O/fuzz_ginflate.txt
timeout 300 python scripts/dev/fuzz_fastq_dev.py 120 ${SEED:-9000} 2>&1 | tail -2 | tee $O/fuzz_fastq.txt

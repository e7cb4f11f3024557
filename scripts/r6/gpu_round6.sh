#!/bin/bash
# Round 6 evidence for the headline step (since this round: binning of the file-order records + scan of the binned set, a sample scanned once; resident_binned beside it), both database
# shapes: bench line, rocprofv3 --kernel-trace --stats of the same command (only headline launches of scan_mini_kernel:
# --no-file-order; scripts/gpu_kstats.sh), separate --pmc passes (FETCH_SIZE; WRITE_SIZE; L2; VALU) -> gpurun_out/r6/<shape>/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp
A="--no-cpu-baseline --no-phases --no-config3 --no-cli-e2e --no-file-order --steps 5 --warmup 2"
for shape in sampled contiguous; do
  O=$R/gpurun_out/r6/$shape; mkdir -p $O; cd $R
  timeout 300 python bench.py --db-shape $shape $A > $O/bench.json 2> $O/bench.err
  bash scripts/gpu_kstats.sh r6_$shape $R/bench.py --db-shape $shape $A > /dev/null 2>&1
  cp $R/gpurun_out/r6_${shape}_kernel_stats.csv $O/kernel_stats.csv
  cd $R
  KERNELS=scan_mini PASS_TIMEOUT=240 bash scripts/gpu_pmc.sh r6_$shape "FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum;SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" $R/bench.py --db-shape $shape $A > /dev/null 2>&1
  cp $R/gpurun_out/r6_${shape}_pmc.txt $O/pmc_summary.txt
  echo "== $shape"; python3 -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['roofline']['kernel_ms'])"; head -5 $O/kernel_stats.csv; cat $O/pmc_summary.txt
done

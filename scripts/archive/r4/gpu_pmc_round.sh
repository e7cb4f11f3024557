#!/bin/bash
# round 4 counters: the BINNED tree scan of both shapes (what the product runs), the cluster scan's memory side, kernel stats
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
export KERNELS=scan_mini PASS_TIMEOUT=240
G="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum;SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
for shape in sampled contiguous; do
  bash scripts/gpu_pmc.sh r4_binned_$shape "$G" $R/scripts/pmc_locality.py bin $shape > /dev/null 2>&1
  cat $O/r4_binned_${shape}_pmc.txt
done
bash scripts/gpu_pmc.sh r4_cluster_mem "TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum;FETCH_SIZE;WRITE_SIZE" $R/scripts/dev/t_hit_heavy.py 5000000 20000000 > /dev/null 2>&1
cat $O/r4_cluster_mem_pmc.txt
export TMPDIR=/tmp; cd /tmp
rm -rf $O/r4_cluster_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_cluster_prof -o scan -- python3 $R/scripts/dev/t_hit_heavy.py 5000000 20000000 > $O/r4_cluster_prof.txt 2>&1
f=$(find $O/r4_cluster_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { head -1 $f; grep "anonymous namespace" $f | grep -v "at::native" | head -12; } | cut -c1-220 | tee $O/r4_cluster_kernel_stats.csv
rm -rf $O/r4_cluster_prof

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export KERNELS=scan_mini PASS_TIMEOUT=240
bash scripts/gpu_pmc.sh r4_cluster2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES;SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY;TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum;TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum;SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE" $R/scripts/dev/t_hit_heavy.py 5000000 20000000 > /dev/null 2>&1
grep "true, 6" $R/gpurun_out/r4_cluster2_pmc.txt
export TMPDIR=/tmp; cd /tmp; O=$R/gpurun_out
rm -rf $O/r4_cluster_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_cluster_prof -o scan -- python3 $R/scripts/dev/t_hit_heavy.py 5000000 20000000 > $O/r4_cluster_prof.txt 2>&1
f=$(find $O/r4_cluster_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" > $O/r4_cluster_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs")
for r in rows:
    n = r["Name"]
    if "anonymous namespace" not in n or "at::native" in n: continue
    short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    print('"%s",%s,%s,%s,%s,%s' % (short, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
PY
head -5 $O/r4_cluster_kernel_stats.csv; tail -1 $O/r4_cluster_prof.txt
rm -rf $O/r4_cluster_prof

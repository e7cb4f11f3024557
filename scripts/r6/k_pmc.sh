#!/bin/bash
# instruction / wait counters of scan_minik_kernel (bench_k.py at one k): rocprofv3 --pmc passes, per launch
# usage: k_pmc.sh [k] [lib]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp BENCH_K_LIST=${1:-25}; cd /tmp
[ -n "${2:-}" ] && export SS_LIB=$2
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  rm -rf $O/pmc_k
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_k -o pmc -- python3 $R/scripts/bench_k.py > /dev/null 2> $O/pmc_k.err
  f=$(find $O/pmc_k -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    kn = r.get("Kernel_Name", "")
    if "scan_mini" not in kn: continue
    short = kn.split("::")[-1].split("(")[0][:40]
    k = (short, r.get("Counter_Name"))
    acc[k][0] += 1; acc[k][1] += float(r.get("Counter_Value", 0))
for (kn, cn), (n, v) in sorted(acc.items()):
    print("%-40s %-22s launches=%d per_launch=%.6g" % (kn, cn, n, v / n))
PY
done
rm -rf $O/pmc_k

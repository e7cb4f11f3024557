#!/bin/bash
# rocprofv3 --pmc passes over the layer-2 kernels at BASELINE configs[3] scale (K = 5 M x S = 300); one pass per ';' group of SETS
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
K=${K:-5000000}; S=${S:-300}
export TMPDIR=/tmp; cd /tmp
IFS=';' read -ra GROUPS_ <<< "${SETS:-SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS;SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE}"
i=0
: > $O/r03_l2_pmc.txt
echo "# rocprofv3 --pmc <group> -- python3 scripts/bench_l2.py $K $S   (per kernel: launches, counter per launch; FETCH_SIZE / WRITE_SIZE in KiB)" >> $O/r03_l2_pmc.txt
for c in "${GROUPS_[@]}"; do
  i=$((i+1))
  rm -rf $O/pmcq_$i
  timeout ${PASS_TIMEOUT:-240} rocprofv3 --pmc $c --output-format csv -d $O/pmcq_$i -o pmc -- python3 $R/scripts/bench_l2.py $K $S > /dev/null 2> $O/pmcq_$i.err
  f=$(find $O/pmcq_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" >> $O/r03_l2_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    kn = r.get("Kernel_Name", "")
    if "anonymous namespace" not in kn or "at::native" in kn: continue
    short = kn.split("::")[-1].split("(")[0][:36]
    k = (short, r.get("Counter_Name"))
    v = float(r.get("Counter_Value", 0))
    acc[k][0] += 1; acc[k][1] += v; acc[k][2] = max(acc[k][2], v)
for (kn, cn), (n, v, mx) in sorted(acc.items()):
    print("%-38s %-24s launches=%d mean=%.6g max=%.6g" % (kn, cn, n, v / n, mx))
PY
  rm -rf $O/pmcq_$i
done
cat $O/r03_l2_pmc.txt

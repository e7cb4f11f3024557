#!/bin/bash
# rocprofv3 --pmc passes over one python script: gpu_pmc.sh NAME "CTR CTR;CTR ..." script.py [args]   (one pass per ';' group)
# -> gpurun_out/NAME_pmc.txt: per kernel of this package and counter: launches, mean and max per launch
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
NAME=$1; SETS=$2; shift 2
export TMPDIR=/tmp; cd /tmp
IFS=';' read -ra GROUPS_ <<< "$SETS"
i=0
: > $O/${NAME}_pmc.txt
echo "# rocprofv3 --pmc <group> -- python3 $(echo "$@" | sed "s#$R/##g")   (per kernel: launches, counter per launch)" >> $O/${NAME}_pmc.txt
for c in "${GROUPS_[@]}"; do
  i=$((i+1))
  rm -rf $O/pmcg_$i
  timeout ${PASS_TIMEOUT:-300} rocprofv3 --pmc $c --output-format csv -d $O/pmcg_$i -o pmc -- python3 "$@" > /dev/null 2> $O/pmcg_$i.err
  f=$(find $O/pmcg_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" "${KERNELS:-}" >> $O/${NAME}_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
only = [k for k in sys.argv[2].split(",") if k]
for r in csv.DictReader(open(sys.argv[1])):
    kn = r.get("Kernel_Name", "")
    if "anonymous namespace" not in kn or "at::native" in kn: continue
    short = kn.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("::")[-1][:40]
    if only and not any(o in short for o in only): continue
    k = (short, r.get("Counter_Name"))
    v = float(r.get("Counter_Value", 0))
    acc[k][0] += 1; acc[k][1] += v; acc[k][2] = max(acc[k][2], v)
for (kn, cn), (n, v, mx) in sorted(acc.items()):
    print("%-40s %-24s launches=%d mean=%.6g max=%.6g" % (kn, cn, n, v / n, mx))
PY
  else echo "# group '$c': no counters (see below)" >> $O/${NAME}_pmc.txt; tail -3 $O/pmcg_$i.err >> $O/${NAME}_pmc.txt; fi
  rm -rf $O/pmcg_$i
done
cat $O/${NAME}_pmc.txt

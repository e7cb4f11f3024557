#!/bin/bash
# rocprofv3 --pmc passes over scripts/pmc_locality.py; MODES="file bin" SETS="A B;C D" (one pass per ';' group), env passes through
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
IFS=';' read -ra GROUPS_ <<< "${SETS}"
for mode in ${MODES:-file bin}; do
i=0
for c in "${GROUPS_[@]}"; do
  i=$((i+1))
  rm -rf $O/pmcl_${mode}_$i
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmcl_${mode}_$i -o pmc -- python3 $R/scripts/pmc_locality.py $mode ${SHAPE:-sampled} > /dev/null 2> $O/pmcl_${mode}_$i.err
  f=$(find $O/pmcl_${mode}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$mode" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    kn = r.get("Kernel_Name", "")
    if not any(x in kn for x in ('scan_mini', 'count_kernel', 'place_kernel')): continue
    short = kn.split("::")[-1].split("(")[0][:40]
    k = (short, r.get("Counter_Name"))
    acc[k][0] += 1
    acc[k][1] += float(r.get("Counter_Value", 0))
for (kn, cn), (n, v) in sorted(acc.items()):
    print("PMC %-5s xcd=%s bits=%s %-40s %-28s n=%d per_launch=%.6g" % (sys.argv[2], __import__("os").environ.get("SS_MINI_XCD", "1"), __import__("os").environ.get("SS_ORDER_BITS", "12"), kn, cn, n, v / n))
PY
  rm -rf $O/pmcl_${mode}_$i
done
done

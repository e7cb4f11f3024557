#!/bin/bash
# rocprofv3 --pmc passes over bench.py; SETS="A B;C D" (one pass per ';'-separated group); ENVV passed through
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
IFS=';' read -ra GROUPS_ <<< "${SETS}"
i=0
for c in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmcx_$i -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > /dev/null 2> $O/pmcx_$i.err
  f=$(find $O/pmcx_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'scan' not in r.get("Kernel_Name", ""): continue
    k = (r.get("Kernel_Name", "")[:44], r.get("Counter_Name"))
    acc[k][0] += 1
    acc[k][1] += float(r.get("Counter_Value", 0))
for (kn, cn), (n, v) in sorted(acc.items()):
    print("PMC %-44s %-34s n=%d per_launch=%.5g" % (kn, cn, n, v / n))
PY
done

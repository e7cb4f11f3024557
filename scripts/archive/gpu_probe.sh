#!/bin/bash
# perf probes for the scan kernel: hit fraction sweep + FETCH_SIZE
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for hf in 0.002 0.05 0.5; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --hit-frac $hf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hit_frac $hf', d['value'], 'M reads/s kernel_ms', d['roofline']['kernel_ms'], d['check'])"
done
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" ; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc2_$n -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc2_$n.err
  f=$(find $O/pmc2_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'scan' not in r.get("Kernel_Name", ""): continue
    k = (r.get("Kernel_Name", "")[:50], r.get("Counter_Name"))
    acc[k][0] += 1
    acc[k][1] += float(r.get("Counter_Value", 0))
for (kn, cn), (n, v) in sorted(acc.items()):
    print("PMC %-50s %-22s launches=%d per_launch=%.5g" % (kn, cn, n, v / n))
PY
done

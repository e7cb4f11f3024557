#!/bin/bash
# time and dynamic VALU/SALU/LDS/VMEM instruction counts of truncated kernels (build_variant.sh stopN "-DSS_STOP_AFTER=N")
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for v in ${LIBS:-stop1 stop2 stop3 base}; do
  lib=$R/build_tmp/lib_$v.so; [ "$v" = base ] && lib=$R/strainscan_amd/lib/libstrainscan_hip.so
  export SS_LIB=$lib
  python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} 2>$O/stop.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$v kernel_ms', d['roofline']['kernel_ms'])" || tail -3 $O/stop.err
  rm -rf $O/pmcs_$v
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/pmcs_$v -o pmc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > /dev/null 2> $O/pmcs_$v.err
  f=$(find $O/pmcs_$v -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'scan_mini' not in r.get("Kernel_Name", ""): continue
    acc[r["Counter_Name"]][0] += 1; acc[r["Counter_Name"]][1] += float(r["Counter_Value"])
tiles = 20e6 * 151 / 992
print("   per tile: " + "  ".join("%s=%.1f" % (k[9:], v / n / tiles) for k, (n, v) in sorted(acc.items())))
PY
done

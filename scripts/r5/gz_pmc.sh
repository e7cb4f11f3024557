#!/bin/bash
# instruction counts of the device inflater's kernels over one load of a 2 x 3 M-read .gz pair (one rocprofv3 --pmc pass)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp SS_GZ_INPROCESS=1; cd /tmp
rm -rf $O/pmc_gz
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $O/pmc_gz -o pmc -- python3 $R/scripts/r5/gz_scale.py 6000000 1 1 > /dev/null 2> $O/pmc_gz.err
f=$(find $O/pmc_gz -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" > $O/r05_gz_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    kn = r.get("Kernel_Name", "")
    if "anonymous namespace" not in kn or "at::native" in kn: continue
    short = kn.split("::")[-1].split("(")[0][:36]
    k = (short, r.get("Counter_Name"))
    acc[k][0] += 1; acc[k][1] += float(r.get("Counter_Value", 0))
print("# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -- python3 scripts/r5/gz_scale.py 6000000 1 1  (one load of a 2 x 3 M-read .gz pair; totals over the launches)")
for (kn, cn), (n, v) in sorted(acc.items()):
    print("%-28s %-18s launches=%d total=%.6g" % (kn, cn, n, v))
PY
rm -rf $O/pmc_gz
cat $O/r05_gz_pmc.txt | grep "sync_kernel\|inflate_kernel\|subsync\|^#"

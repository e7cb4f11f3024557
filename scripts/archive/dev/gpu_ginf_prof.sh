#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/gi; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
pmc() { timeout 600 rocprofv3 --pmc $2 --output-format csv -d $O/pmc_$1 -o pmc -- python3 $R/scripts/dev/t_ginf_prof.py 400000 > $O/pmc_$1.out 2> $O/pmc_$1.err
  f=$(find $O/pmc_$1 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    kn = r.get("Kernel_Name", "")
    if "inflate_kernel" not in kn and "sync_kernel" not in kn: continue
    k = (kn.split("::")[-1].split("(")[0], r["Counter_Name"]); acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for (kn, cn), (n, v) in sorted(acc.items()): print("%-16s %-28s per_launch=%.5g" % (kn, cn, v / n))
PY
}
SS_GZ_CHUNK=32768 pmc a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"
SS_GZ_CHUNK=32768 pmc b "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
SS_GZ_CHUNK=32768 pmc c "SQ_IFETCH SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"
grep "gpu inflate" $O/pmc_a.out | tail -2

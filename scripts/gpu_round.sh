#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel stats.  Run through gpurun.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== nproc $(nproc); $(rocm-smi --showproductname 2>/dev/null | grep -m1 -i 'card series' || true)"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $O/smoke.log
timeout 900 python bench.py ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err; tail -5 $O/bench.err; cat $O/bench.json
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o scan -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof_bench.err
tail -3 $O/prof_bench.err; cat $O/prof_bench.json
find $O/prof -name "*stats*" | head -5
for f in $(find $O/prof -name "*kernel_stats.csv" | head -1); do head -12 $f; done
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_$n.err
  f=$(find $O/pmc_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r.get("Kernel_Name", "")[:60], r.get("Counter_Name"))
    acc[k][0] += 1
    acc[k][1] += float(r.get("Counter_Value", 0))
for (kn, cn), (n, v) in sorted(acc.items(), key=lambda x: -x[1][1])[:8]:
    print("PMC %-60s %-24s launches=%d total=%.4g per_launch=%.4g" % (kn, cn, n, v, v / n))
PY
done

#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel stats and PMC traffic.  Run through gpurun.
# Output: gpurun_out/{pytest_gpu.log,smoke.log,bench.json,prof/*,pmc_*}
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== nproc $(nproc)"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.log
timeout 900 python bench.py ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cat $O/bench.json
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o scan -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof_bench.err
cat $O/prof_bench.json
for f in $(find $O/prof -name "*kernel_stats.csv" | head -1); do head -6 $f | cut -c1-200; done
pmc() {  # name, counters, extra bench args
  timeout 600 rocprofv3 --pmc $2 --output-format csv -d $O/pmc_$1 -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $3 > /dev/null 2> $O/pmc_$1.err
}
pmc fetch FETCH_SIZE ""
pmc write WRITE_SIZE ""
pmc l2 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" ""
pmc calib_fetch FETCH_SIZE "--calib-stream"
python3 $R/scripts/summarize_pmc.py $O

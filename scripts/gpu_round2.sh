#!/bin/bash
# Round-2 evidence run for ONE bench configuration: bench line, rocprofv3 kernel stats, PMC passes (traffic, L2, VALU,
# streaming calibration).  TAG=name  BENCH_ARGS="--db-shape sampled --hit-frac 0.05"  -> gpurun_out/r2/<TAG>/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${TAG:-sampled}; O=$R/gpurun_out/r2/$TAG; mkdir -p $O; cd $R
A="${BENCH_ARGS:-}"
timeout 900 python bench.py $A ${BENCH_EXTRA:-} > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err | cut -c1-200; cut -c1-400 $O/bench.json
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o scan -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-phases --no-readset --no-config3 $A > $O/prof_bench.json 2> $O/prof_bench.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp $f $O/kernel_stats.csv; head -8 $f | cut -c1-160; }
pmc() {  # name, counters, extra bench args
  timeout 600 rocprofv3 --pmc $2 --output-format csv -d $O/pmc_$1 -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-phases --no-readset --no-config3 $A $3 > /dev/null 2> $O/pmc_$1.err
}
pmc fetch FETCH_SIZE ""
pmc write WRITE_SIZE ""
pmc l2 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" ""
pmc valu "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" ""
pmc wr "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum" ""
[ "${CALIB:-0}" = 1 ] && pmc calib_fetch FETCH_SIZE "--calib-stream"
rm -rf $O/prof/*/*.db 2>/dev/null
python3 $R/scripts/summarize_pmc.py $O | tail -40

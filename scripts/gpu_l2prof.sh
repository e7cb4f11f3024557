#!/bin/bash
# rocprofv3 kernel stats of the layer-2 kernels at BASELINE configs[3] scale (K = 5 M k-mers x S = 300 strains)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r2/l2; mkdir -p $O; cd $R
K=${K:-5000000}; S=${S:-300}
timeout 1200 python scripts/bench_l2.py $K $S > $O/bench_l2.json 2> $O/bench_l2.err; cat $O/bench_l2.json; tail -2 $O/bench_l2.err
export TMPDIR=/tmp; cd /tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o l2 -- python3 $R/scripts/bench_l2.py $K $S > $O/prof.json 2> $O/prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { grep -v "at::native\|rocprim\|hipcub" $f > $O/kernel_stats.csv; cut -c1-150 $O/kernel_stats.csv | head -20; }
rm -rf $O/prof

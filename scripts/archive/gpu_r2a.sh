#!/bin/bash
# round 2, first measurement: the round-1 kernel on both database shapes, filter-size sweep on the sampled one
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r2a; mkdir -p $O; cd $R
run() { # tag, env..., -- args
  tag=$1; shift
  ( timeout 600 env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${ARGS:-} > $O/$tag.json 2> $O/$tag.err ) ; tail -2 $O/$tag.err | cut -c1-300; cat $O/$tag.json | cut -c1-900
}
ARGS="--db-shape contiguous" run contig SS_X=0
ARGS="--db-shape sampled" run sampled SS_X=0
for b in 0 25 27 28 29 30; do ARGS="--db-shape sampled" run sampled_bloom$b SS_BLOOM_BITS=$b; done
ARGS="--db-shape sampled --hit-frac 0.5" run sampled_h50 SS_X=0
ARGS="--db-shape sampled --hit-frac 0.002" run sampled_h002 SS_X=0

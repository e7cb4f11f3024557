R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
A="--no-cpu-baseline --no-phases --no-config3 --no-cli-e2e --no-file-order --steps 5 --warmup 2"
for shape in sampled contiguous; do bash scripts/gpu_kstats.sh r5_$shape $R/bench.py --db-shape $shape $A 2>&1 | head -6; done

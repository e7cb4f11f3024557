#!/bin/bash
# The round 6: final evidence in one gpurun call: the default bench line of both shapes, the CLI with layer 2 on the path
# (20 M reads text, 50 M text, 50 M .gz).  -> gpurun_out/r6_final/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6_final; mkdir -p $O; cd $R
timeout 600 python bench.py > $O/sampled_bench.json 2> $O/sampled_bench.err
timeout 600 python bench.py --db-shape contiguous > $O/contiguous_bench.json 2> $O/contiguous_bench.err
timeout 900 python scripts/bench_cli_l2.py 20000000 > $O/cli_l2_text.json 2> $O/cli_l2_text.err
if [ "${1:-all}" = "all" ]; then
  timeout 1200 python scripts/bench_cli_l2.py 50000000 > $O/cli_l2_text50.json 2> $O/cli_l2_text50.err
  timeout 1500 python scripts/bench_cli_l2.py 50000000 5000000x300,2000000x120,1000000x60 gz > $O/cli_l2_gz50.json 2> $O/cli_l2_gz50.err
fi
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*_bench.json")):
    d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["bound"], "resident", d["resident_binned"]["value"], "file", d["file_order"]["value"], "prep", d["prepare"]["ms"], d["step_breakdown_ms"], d["cpu_baseline"]["value"], d["cpu_baseline"]["parity_on_sample"])
    print("  cluster", d["cluster_scan"]["binned"]["kernel_ms"], d["cluster_scan"]["three_tables"]["one_pass_ms"], "l2", d["l2_solve"]["wall_ms"], d["l2_solve"]["four_clusters"]["wall_ms"], "gz", d["phases"]["gz_ingest"]["device_ms"], d["phases"]["gz_ingest"]["device_ms_all"], "e2e", d["e2e_reads_per_s"])
    print("  cli_e2e", [r["wall_s"] for r in d["cli_e2e"]["fresh_process"]])
for f in sorted(glob.glob("$O/cli_l2_*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], [(r["label"], r["wall_s"]) for r in d["cli_fresh_process"]], d["in_process"]["warm"]["total_s"], d["all_expected_strains_reported"])
    print("  ", d["cli_fresh_process"][-1]["phases_s"])
PY

# cluster scan + headline, short: bench blocks cluster_scan and the binned step
python bench.py --no-cpu-baseline --no-phases --steps 5 --l2-rows 300000 --l2-strains 40 --l2-check-rows 100000 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); cs=d['cluster_scan']
print('headline', d['value'], d['roofline']['kernel_ms'], 'file', d['file_order']['roofline']['kernel_ms'], 'prep', d['prepare']['ms'], d['prepare']['ms_kernels'])
print('cluster file', cs['file_order']['kernel_ms'], 'binned', cs['binned']['kernel_ms'], 'nocomb', cs['binned_without_lds_combining']['kernel_ms'], 'three', cs['three_tables']['one_pass_ms'], cs['three_tables']['scan_per_table_ms'], 'eq', cs['counts_equal_across_orders'], cs['parity_on_sample'], cs['three_tables']['counts_equal'])"

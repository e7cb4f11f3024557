for hf in ${HFS:-0.5 0.2}; do for shape in sampled contiguous; do for c in ${MODES:-"" 2}; do
  SS_COMBINE=$c python bench.py --hit-frac $hf --db-shape $shape --no-cpu-baseline --no-phases --no-config3 --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('hf $hf $shape comb=[$c]', 'binned', d['roofline']['kernel_ms'], 'file', d['file_order']['roofline']['kernel_ms'], 'hits', d['check']['total_hits'], 'inline', d['config']['index']['inline_kmers'], 'eq', d['file_order']['node_stats_equal'])"
done; done; done

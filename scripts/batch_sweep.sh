# batch-size sweep of bench.py (diagnostic)
for b in 250000 500000 1000000 2000000; do
  python bench.py --steps 2 --warmup 1 --no-cpu --no-ceilings --batch-reads $b 2>/dev/null > /tmp/o.json
  python -c "
import json;d=json.load(open('/tmp/o.json'));k=d['kernel_ms_per_step_rank0'];print('batch $b', '%.3e'%d['value'], round(d['ms_per_step'],1), d['outputs']['flag_positions_rank0'], k['walk_stage'], k['scan_flags'], k['load_mark'], k['load_resolve'])"
done

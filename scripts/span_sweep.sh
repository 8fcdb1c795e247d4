# walk-window span sweep of bench.py (diagnostic)
for l in 21 22 23 24 25; do
  FGPU_MAX_SPAN_LOG2=25 FAUCET_WALK_SPAN=$((1<<l)) python bench.py --steps 2 --warmup 1 --no-cpu --no-ceilings --profile-walk 2>/dev/null > /tmp/o.json
  python -c "
import json;d=json.load(open('/tmp/o.json'));k=d['kernel_ms_per_step_rank0'];o=d['outputs'];print('span 2^$l', '%.3e'%d['value'], round(d['ms_per_step'],1), o['walk_windows_rank0'], o['walk_followers_rank0'], o['walk_max_cluster_rank0'], o['flag_positions_rank0'], {n:k[n] for n in ('walk_stage','walk','walk_lookup','walk_link','walk_clean','walk_cluster','scan_flags')})"
done

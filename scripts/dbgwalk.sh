for d in 0 1 2 3 4 7; do
  FGPU_DEBUG_WALK=$d FGPU_NO_OVERLAP=1 python bench.py --steps 1 --warmup 1 --no-cpu --profile-walk 2>/dev/null > /tmp/o.json
  python -c "
import json;d=json.load(open('/tmp/o.json'));k=d['kernel_ms_per_step_rank0'];print('dbg $d walk',k['walk'],'stage',k['walk_stage'],d['outputs']['walk_windows_rank0'])"
done

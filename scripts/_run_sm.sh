timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for b in 0 1024 2048 4096 8192; do
  echo "== FGPU_FLAGS_SM_BLOCKS=$b"
  FGPU_FLAGS_SM_BLOCKS=$b timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-ceilings 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step_rank0']
print('%.3e'%d['value'], round(d['ms_per_step'],1), 'flags', k['scan_flags'], 'walk_stage', k['walk_stage'], 'junctions', d['outputs']['junctions'])"
done
for b in 0 2048 4096; do
  echo "== no overlap FGPU_FLAGS_SM_BLOCKS=$b"
  FGPU_FLAGS_SM_BLOCKS=$b FGPU_NO_OVERLAP=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-ceilings 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step_rank0']
print('%.3e'%d['value'], round(d['ms_per_step'],1), 'flags', k['scan_flags'], 'walk_stage', k['walk_stage'])"
done

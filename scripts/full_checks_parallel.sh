#!/bin/bash
# Runs ON THE GPU BOX: the larger-than-suite oracle checks side by side (each is seconds of GPU and minutes of one host core).
#   gpurun --timeout 1100 -- 'bash scripts/full_checks_parallel.sh'
out=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/full_checks
mkdir -p "$out"
python scripts/full_oracle_check.py 10000000 2 1000 > "$out/config2.log" 2>&1 &
python scripts/full_oracle_check.py 10000000 102 5002 > "$out/seeds_102_5002.log" 2>&1 &
python scripts/full_oracle_check.py 5000000 7 77 8000 > "$out/repeats.log" 2>&1 &
python scripts/full_oracle_check.py 3000000 5 55 0 150 0.05 two_hash > "$out/config5_shape.log" 2>&1 &
while [ -n "$(jobs -r)" ]; do sleep 30; echo "[$(date +%T)] still running: $(jobs -r | wc -l)"; done
for f in config2 seeds_102_5002 repeats config5_shape; do echo "--- $f"; grep -v amdgpu.ids "$out/$f.log" | tail -4; done
# ... and the variants of the suite's full-size / multi-variant checks that the driver-run `pytest -m gpu` leaves out (tests/conftest.py: marker `slow`)
python -m pytest tests -q -m "gpu and slow" > "$out/slow_variants.log" 2>&1; echo "--- slow variants"; tail -3 "$out/slow_variants.log"

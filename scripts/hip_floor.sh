#!/bin/bash
# the floor of any HIP process on this box, beside the command line's own clock (scripts/cli_phase_clock.sh)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && hipcc --offload-arch=gfx950 -O2 -o /tmp/hip_floor "$root/scripts/micro/hip_floor.hip" 2>/dev/null || exit 1
for i in 1 2 3 4; do
  s=$(date +%s.%N); /tmp/hip_floor; e=$(date +%s.%N); echo "   process, return from main: $(python3 -c "print(round(($e-$s)*1e3))") ms"
  s=$(date +%s.%N); /tmp/hip_floor x; e=$(date +%s.%N); echo "   process, _exit:            $(python3 -c "print(round(($e-$s)*1e3))") ms"
done

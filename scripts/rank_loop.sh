#!/bin/bash
# Runs ON THE GPU BOX: the N-rank pipeline (scripts/two_rank_check.py: product backend, N processes sharing the one GPU, gloo as the transport with
# event-ordered page-locked staging) repeated REPS times in both stream modes -- the library on a stream of its own (host fences) and on
# torch's stream (no host fence, what bench.py runs for N > 1).  One line per run; a wrong bloo2 / junction map or a hang shows up as FAIL.
#   gpurun -- 'bash scripts/rank_loop.sh 3 700000 20 > gpurun_out/rank_loop.txt'
ranks=${1:-3}; per=${2:-700000}; reps=${3:-20}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
export GLOO_SOCKET_IFNAME=lo
fail=0
for mode in own_stream torch_stream; do
  for i in $(seq 1 $reps); do
    port=$((20000 + RANDOM % 20000))
    if [ $mode = torch_stream ]; then export FAUCET_TORCH_STREAM=1; else unset FAUCET_TORCH_STREAM; fi
    out=$(timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$ranks --master-addr 127.0.0.1 --master-port $port scripts/two_rank_check.py $per 2>&1)
    rc=$?
    line=$(echo "$out" | grep -E "RESULT" | tail -1)
    inits=$(echo "$out" | grep -c "INIT OK")
    if echo "$line" | grep -q "RESULT PASS"; then verdict=PASS; else verdict=FAIL; fail=$((fail + 1)); fi
    echo "$mode run $i: rc=$rc ranks initialised $inits/$ranks  $verdict  ${line:0:160}"
  done
done
echo "failures: $fail of $((2 * reps)) runs ($ranks ranks x $per reads)"
exit $fail

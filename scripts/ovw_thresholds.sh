#!/bin/bash
# Runs ON THE GPU BOX: which clusters should the optimistic walk (k_ovw_round) take?  Config 2's step (bench.py) and config 3's shape through the
# CLI under several thresholds (FGPU_WALK_KO = pieces, FGPU_WALK_KO_WEIGHT = lk positions of a cluster).  Measurement aid.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
python3 - <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
from faucet_amd import synth_det as sd
fx = json.load(open("tests/golden/fullsize.json"))["config3"]
c = fx["params"]
dev = torch.device("cuda", 0)
g = sd.make_genome(c["genome"], c["genome_seed"], dev)
sd.plant_repeats(g, c["genome_seed"] + 100, *c["repeats"])
reads = sd.make_pairs(g, c["pairs"], c["read_len"], c["insert"][0], c["insert"][1], c["err"], c["read_seed"], dev)
sd.fasta_bytes(reads, fastq=True).cpu().numpy().tofile("/dev/shm/c3_reads.fq")
open("/dev/shm/c3_args.txt", "w").write(" ".join(fx["args"]))
PY
run_c3() {
  for i in 1 2; do
    env "$@" FGPU_CLI_TIMES=1 $root/faucet_amd/faucet -read_load_file /dev/shm/c3_reads.fq -read_scan_file /dev/shm/c3_reads.fq -file_prefix /dev/shm/c3_out $(cat /dev/shm/c3_args.txt) > /dev/null 2> /tmp/c3.err
    echo "config3 [$*] $(grep -E 'pass 1 \(|pass 2 \(' /tmp/c3.err | awk '{printf "%s %s ms  ", $2, $(NF-4)}') $(grep 'optimistically' /tmp/c3.err | sed 's/.*optimistically: //')"
  done
  sha256sum /dev/shm/c3_out.junctions /dev/shm/c3_out.long_pair_filter | cut -c1-16 | tr '\n' ' '; echo
}
run_c3 FGPU_X=0
run_c3 FGPU_WALK_KO_AVG=8 FGPU_WALK_KO_AVG_MIN=16
run_c3 FGPU_WALK_KO_AVG=6 FGPU_WALK_KO_AVG_MIN=12
run_c3 FGPU_WALK_KO_AVG=8 FGPU_WALK_KO_AVG_MIN=16 FGPU_WALK_KO=16
run_c3 FGPU_OVW_FOLLOWERS=0
run_c3 FGPU_OVW_FOLLOWERS=0 FGPU_WALK_KO_AVG=8 FGPU_WALK_KO_AVG_MIN=16
run_c3 FGPU_WALK_KO=8 FGPU_WALK_KO_WEIGHT=32
rm -f /dev/shm/c3_reads.fq /dev/shm/c3_out.* /dev/shm/c3_args.txt
for v in "FGPU_WALK_KO=32 FGPU_WALK_KO_ALWAYS=1 FGPU_WALK_KO_WEIGHT=128" "FGPU_WALK_KO=32 FGPU_WALK_KO_ALWAYS=1 FGPU_WALK_KO_WEIGHT=128 FGPU_WALK_KO_AVG=8 FGPU_WALK_KO_AVG_MIN=16" "FGPU_WALK_KO=32 FGPU_WALK_KO_ALWAYS=1 FGPU_WALK_KO_WEIGHT=128 FGPU_WALK_KO_AVG=6 FGPU_WALK_KO_AVG_MIN=12"; do
  env $v python3 bench.py --steps 10 --warmup 3 --no-cpu --no-ceilings --no-host-leg --no-full-size > /tmp/b.json 2> /tmp/b.err
  python3 - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
k = d["kernel_ms_per_step_rank0"]
print(f"config2 [{sys.argv[1]}] {d['ms_per_step']:.2f} ms/step  walk_stage {k.get('walk_stage')}  ovw pieces {d['outputs']['walk_key_ordered_pieces_rank0']}  junctions {d['outputs']['junctions']}")
PY
done

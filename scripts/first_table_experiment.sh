#!/bin/bash
# Runs ON THE GPU BOX: what a SMALLER table of first-set times would buy pass 1, measured before building the structure that
# would make it exact.  -DFGPU_FIRST_MASK_LOG2=n folds first[] (4 bytes per Bloom bit: 2 GiB on config 2) into 2^n entries: the atomicMin of
# k_load_mark and the loads of k_load_resolve then go to a 128 MiB / 4 MiB table (results are WRONG in these builds; only the kernels' times are
# looked at; pass 1 alone is run: scripts/load_only_times.py).  The sweep (k_carry_from_first) still reads the whole array: its time is what a smaller table would save outright.
#   gpurun -- 'bash scripts/first_table_experiment.sh > gpurun_out/first_table.txt'
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
mkdir -p gpurun_out
for v in "full_2GiB:" "fold_128MiB:-DFGPU_FIRST_MASK_LOG2=25" "fold_4MiB:-DFGPU_FIRST_MASK_LOG2=20" "full_again:"; do
  n=${v%%:*}; export FGPU_EXTRA_CXXFLAGS="${v#*:}"
  python -m faucet_amd.build --force > gpurun_out/ft_build_$n.log 2>&1 || { echo "$n: build failed"; continue; }
  timeout -k 10 300 python scripts/load_only_times.py $n 2> gpurun_out/ft_$n.err
done

#!/bin/bash
# (Round 3's experimental variants -- -DFGPU_KO_SCALAR, -DFGPU_KO_EAGER_GIVE, -DFGPU_KO_NO_PREFETCH, -DFGPU_KO_NO_CACHE, -DFGPU_KO_ONE_XCD, -DFGPU_KO_SLEEP=n --
# exist in the tree at commit 530a5c9; the walk in the tree now is round 2's algorithm, which measured fastest.  profiles/r03_ko_walk.txt.)
# Runs ON THE GPU BOX: the key-ordered walk built in several measurement variants; each runs one fuzz case with the walk forced onto every
# cluster (correctness) and scripts/pe_profile.py (time of walk_ko on the repeat-rich paired-end set).
#   gpurun -- 'bash scripts/ko_variants.sh "name:-DFLAG ..." ...'
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
mkdir -p gpurun_out
for v in "$@"; do
  n=${v%%:*}; export FGPU_EXTRA_CXXFLAGS="${v#*:}"
  # a variant named prev* is built from tmp_prev/scan_walk_prev.hip (an earlier version of the walk kept aside for a same-box comparison)
  if [[ $n == prev* ]]; then cp faucet_amd/csrc/scan_walk.hip /tmp/scan_walk_cur.hip; cp tmp_prev/scan_walk_prev.hip faucet_amd/csrc/scan_walk.hip; fi
  python -m faucet_amd.build --force > gpurun_out/kov_build_$n.log 2>&1 || { echo "$n: build failed"; continue; }
  FGPU_KO_WAIT_S=10 FGPU_WALK_KO=2 FGPU_WALK_KO_ALWAYS=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fuzz or key_ordered or tandem" > gpurun_out/kov_test_$n.log 2>&1
  echo "== $n [$FGPU_EXTRA_CXXFLAGS] tests: $(tail -1 gpurun_out/kov_test_$n.log)"
  FGPU_WALK_KO_ALWAYS=1 FGPU_PROFILE_WALK=1 timeout -k 10 300 python scripts/pe_profile.py > gpurun_out/kov_pe_$n.log 2>&1
  grep -E "^walk_ko |^walk_stage" gpurun_out/kov_pe_$n.log
  if [[ $n == prev* ]]; then cp /tmp/scan_walk_cur.hip faucet_amd/csrc/scan_walk.hip; fi
done

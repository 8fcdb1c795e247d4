#!/bin/bash
# ON THE GPU BOX: scripts/fuzz_vs_reference.py at values of k the suite's draws do not take: even k (k-mers that are their own reverse complement)
# and very small k.   bash scripts/fuzz_vs_reference_k.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
for k in 4 8 14 20 22 30 3 5 9 27; do
  echo "k = $k: $(FUZZ_K=$k python3 scripts/fuzz_vs_reference.py 2000 2030 2>&1 | tail -1)"
done

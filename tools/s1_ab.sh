#!/bin/bash
# Runs ON THE GPU BOX: tools/s1_only.py (the bottleneck tail's kernels alone) for the product build and the variant libraries given as arguments
cd "$GRAFT_REPO_ROOT"
for so in "" "$@"; do
  echo "== ${so:-product}"
  for shape in "256 56" "512 28" "1024 14" "2048 7"; do ALIGNQ_SO=$so python3 tools/s1_only.py $shape 20 2>/dev/null; done
done

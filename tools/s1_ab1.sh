#!/bin/bash
# Runs ON THE GPU BOX: tools/s1_only.py at ONE shape for the product build and the variant libraries given as arguments
cd "$GRAFT_REPO_ROOT"
for so in "" "$@"; do
  echo "== ${so:-product}"; ALIGNQ_SO=$so python3 tools/s1_only.py 256 56 20 2>/dev/null
done

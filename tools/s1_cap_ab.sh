cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do for so in "" tools/lib/lib_s1cap512.so tools/lib/lib_s1cap640.so; do echo "== ${so:-product}"; ALIGNQ_SO=$so python3 tools/s1_only.py 256 56 30 2>/dev/null; ALIGNQ_SO=$so python3 tools/s1_only.py 512 28 30 2>/dev/null; done; done

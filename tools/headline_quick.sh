#!/bin/bash
# Runs ON THE GPU BOX: the headline step time alone (ms_per_step of the default workload), e.g. for an A/B of two builds (ALIGNQ_SO).
cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps ${STEPS:-300} --warmup 20 --no-cpu-baseline --no-kernels --no-shapes --no-dp-probe --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline ms_per_step', d['ms_per_step'], 'value', d['value'])"

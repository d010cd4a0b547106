#!/bin/bash
# Runs ON THE GPU BOX: config 5's captured step time alone (one JSON line, ms_per_step), e.g. for an A/B of two builds (ALIGNQ_SO).
cd "$GRAFT_REPO_ROOT"
python3 bench.py --model resnet50_dann --batch 28 --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-kernels --no-shapes --no-dp-probe --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('office ms_per_step', d['ms_per_step'], 'value', d['value'])"

#!/bin/bash
# Runs ON THE GPU BOX: configuration 5's captured step alone and its data-parallel form at world size 1 (--dp-selftest: RCCL, buckets
# packed inside the captured backward, collectives ordered by flags) -> two "office ms_per_step" lines
cd "$GRAFT_REPO_ROOT"
for extra in "" "--dp-selftest"; do
python3 bench.py --model resnet50_dann --batch 28 --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-kernels --no-shapes --no-dp-probe --no-other-configs $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('office ${extra:-single-graph} ms_per_step', d['ms_per_step'], 'value', d['value'])"
done

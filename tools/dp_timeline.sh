#!/bin/bash
# Runs ON THE GPU BOX: configuration 5's captured data-parallel step at world size 1 (--dp-selftest) under rocprofv3 --kernel-trace: where in
# the replayed step the buckets' pack / flag / all-reduce kernels run (offsets from the step's first kernel) -> gpurun_out/dp_timeline.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/dptl && mkdir -p gpurun_out/dptl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dptl/tr -o run -- python3 bench.py --model resnet50_dann --batch 28 --steps 12 --warmup 3 --no-cpu-baseline --no-kernels --no-shapes --no-dp-probe --no-other-configs --dp-selftest > gpurun_out/dptl/log 2>&1
python3 - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/dptl/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
short = lambda n: re.sub(r'\(anonymous namespace\)::|alignq_site::|void ', '', n)[:60]
# a step starts at the counter bump
starts = [i for i, r in enumerate(rows) if 'dp_bump_kernel' in r[2]]
out = []
for a, b in list(zip(starts, starts[1:]))[-4:]:
    t0 = rows[a][0]
    seg = rows[a:b]
    wall = (rows[b][0] - t0) / 1e3
    busy = sum(e - s for s, e, _ in seg) / 1e3
    out.append(f"step: {len(seg)} kernels, period {wall:.1f} us, sum of kernel durations {busy:.1f} us")
    for s, e, n in seg:
        if any(k in n for k in ('nccl', 'rccl', 'mt_copy_kernel', 'dp_publish', 'dp_bump', 'mt_weight', 'mt_sgd', 'admm')):
            out.append(f"   +{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {short(n)}")
open('gpurun_out/dp_timeline.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[-60:]))
PY
rm -rf gpurun_out/dptl

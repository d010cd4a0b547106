#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): where configuration 5's step goes, kernel by kernel (ResNet-50-DANN, batch 28 + 28, eager
# launches so every kernel shows by name, MIOpen find mode as in the bench).  The find-mode trial kernels of the warm-up would
# swamp a whole-run --stats table, so the per-dispatch trace is aggregated over the LAST timed steps only.
# -> gpurun_out/prof_office/office_step_kernels.csv (name, calls per step, us per step)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_office && mkdir -p gpurun_out/prof_office
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_office/raw -o run -- python3 bench.py --model resnet50_dann --batch 28 --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-kernels $OFFICE_FLAGS > gpurun_out/prof_office/stats.log 2>&1
grep "^{\"metric" gpurun_out/prof_office/stats.log > gpurun_out/prof_office/bench_under_rocprof.json || true
python3 - <<'PY'
import csv, glob, json, re, collections
f = glob.glob('gpurun_out/prof_office/raw/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
ms = json.loads(open('gpurun_out/prof_office/bench_under_rocprof.json').read())['ms_per_step']
steps = 8
t_end = rows[-1][1]
t0 = t_end - int(steps * ms * 1e6)
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    if s >= t0:
        n = re.sub(r'\(anonymous namespace\)::|alignq_site::|void ', '', n)
        n = re.sub(r'<.*', '', n)[:70] if n.startswith(('ck::', '_ZN2ck', 'at::native')) else n[:90]
        agg[n][0] += 1
        agg[n][1] += e - s
tot = sum(v[1] for v in agg.values())
with open('gpurun_out/prof_office/office_step_kernels.csv', 'w') as fo:
    fo.write(f"# per step over the last {steps} steps ({ms:.2f} ms per step under rocprofv3, eager); kernel time per step {tot / steps / 1e6:.2f} ms\n")
    fo.write("name,calls_per_step,us_per_step\n")
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fo.write(f"\"{n}\",{c / steps:.1f},{d / steps / 1e3:.1f}\n")
print(open('gpurun_out/prof_office/office_step_kernels.csv').read()[:6000])
PY
rm -rf gpurun_out/prof_office/raw

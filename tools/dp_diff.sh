#!/bin/bash
# Runs ON THE GPU BOX: configuration 5's captured step, plain and data-parallel (world size 1, --dp-selftest), under rocprofv3 --kernel-trace;
# per kernel family: calls and summed duration per step in both forms and the difference -> gpurun_out/dp_diff.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/dpd && mkdir -p gpurun_out/dpd
for mode in plain dp; do
  extra=""; [ $mode == dp ] && extra="--dp-selftest"
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dpd/$mode -o run -- python3 bench.py --model resnet50_dann --batch 28 --steps 12 --warmup 3 --no-cpu-baseline --no-kernels --no-shapes --no-dp-probe --no-other-configs $extra > gpurun_out/dpd/$mode.log 2>&1
done
python3 - <<'PY'
import csv, glob, re, collections, json
def load(mode):
    f = glob.glob(f'gpurun_out/dpd/{mode}/**/*kernel_trace.csv', recursive=True)[0]
    rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
    # a step = from one stem forward (once per step in both forms) to the next; the last 6 complete steps
    marks = [i for i, r in enumerate(rows) if 'qstem7_fwd_kernel' in r[2]]
    lo, hi = marks[-7], marks[-1]
    steps = 6
    ms = (rows[hi][0] - rows[lo][0]) / steps / 1e6
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in rows[lo:hi]:
        n = re.sub(r'\(anonymous namespace\)::|alignq_site::|void ', '', n)
        n = re.sub(r'[<(].*', '', n)[:48]
        agg[n][0] += 1; agg[n][1] += e - s
    return ms, {k: (v[0] / steps, v[1] / steps / 1e3) for k, v in agg.items()}
mp, a = load('plain'); md, b = load('dp')
out = [f"step period under rocprofv3 (stem forward to stem forward, last 6 steps): plain {mp:.3f} ms, data-parallel (world 1) {md:.3f} ms",
       f"{'kernel':50s} {'calls':>12s} {'us per step':>22s} {'diff':>8s}"]
for k in sorted(set(a) | set(b), key=lambda k: -abs(b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1])):
    ca, ua = a.get(k, (0, 0.0)); cb, ub = b.get(k, (0, 0.0))
    if abs(ub - ua) < 3 and ca == cb: continue
    out.append(f"{k:50s} {ca:5.1f} -> {cb:5.1f} {ua:9.1f} -> {ub:9.1f} {ub - ua:+8.1f}")
out.append(f"sum of kernel durations: plain {sum(v[1] for v in a.values()):.1f}, dp {sum(v[1] for v in b.values()):.1f} us")
open('gpurun_out/dp_diff.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf gpurun_out/dpd

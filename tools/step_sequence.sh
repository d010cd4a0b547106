# ordered kernel list of ONE captured step (between two stem forward launches): bash tools/step_sequence.sh  (on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/seq && mkdir -p gpurun_out/seq
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seq/t -o run -- python3 bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-kernels --no-dp-probe --no-shapes > gpurun_out/seq/log 2>&1
python3 - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/seq/t/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [re.sub(r'\(anonymous namespace\)::|alignq_site::|void ', '', r['Kernel_Name'])[:90] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith('stem_fwd_kernel')]
a, b = idx[-2], idx[-1]
# rotate so that the listing starts at the first kernel after the previous step's last optimizer kernel
with open('gpurun_out/step_sequence.txt', 'w') as fo:
    prev_end = None
    for i in range(a, b):
        r = rows[i]
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (st - prev_end) / 1e3 if prev_end else 0.0
        fo.write(f"{i - a:4d} gap {gap:6.2f} us  dur {(en - st) / 1e3:7.2f} us  {names[i]}\n")
        prev_end = en
print(open('gpurun_out/step_sequence.txt').read())
PY
rm -rf gpurun_out/seq

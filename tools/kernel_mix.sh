# Runs ON THE GPU BOX: share of the step's kernel time per kernel: ARGS='--model resnet50_dann --batch 28 --steps 10 --warmup 2' bash tools/kernel_mix.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/mix
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mix -o run -- python3 bench.py $ARGS --no-cpu-baseline --no-kernels --no-dp-probe --no-shapes > gpurun_out/mix.log 2>&1
python3 - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/mix/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
ours = 0.0
out = []
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|alignq_site::|void ", "", r["Name"])
    mine = not (n.startswith("at::") or "miopen" in n.lower() or "Cijk" in n or "rocclr" in n or "ck::" in n or "igemm" in n.lower() or "naive_conv" in n or "batched_transpose" in n or "SubTensor" in n or "gridwise" in n)
    share = 100 * float(r["TotalDurationNs"]) / tot
    ours += share if mine else 0
    out.append("%-74s calls %6d avg %9.1f us  %5.1f %% %s" % (n[:74], int(r["Calls"]), float(r["AverageNs"]) / 1e3, share, "*" if mine else ""))
open("gpurun_out/kernel_mix.txt", "w").write("share of kernel time in this repository's kernels (*): %.1f %%\n" % ours + "\n".join(out[:60]) + "\n")
PY
rm -rf gpurun_out/mix

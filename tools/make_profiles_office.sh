#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): configuration 5's in-scope chains under rocprofv3:
#   1. kernel stats of tools/office_shapes.py (bnq_* / site1_* at the network's shapes: time per launch by kernel);
#   2. the two PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs, MI355X_MICROARCH.md) of the same program -> HBM bytes per launch;
#   3. the eager step kernel by kernel (tools/profile_office.sh) and the captured step's timeline (tools/step_timeline.sh).
# -> gpurun_out/profiles/${ROUND}_office_*
set -e
export ROUND=${ROUND:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_o && mkdir -p gpurun_out/prof_o gpurun_out/profiles
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_o/stats -o run -- python3 tools/office_shapes.py > gpurun_out/prof_o/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_o/fetch -o run -- python3 tools/office_shapes.py > gpurun_out/prof_o/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_o/write -o run -- python3 tools/office_shapes.py > gpurun_out/prof_o/write.log 2>&1
grep -E "^(bnq_|bn_site_)" gpurun_out/prof_o/stats.log > gpurun_out/profiles/${ROUND}_office_shapes_under_rocprof.txt || true
python3 - <<'PY'
import os; R = os.environ["ROUND"]
import csv, glob, collections, re
def one(pat):
    f = glob.glob(pat, recursive=True); assert f, pat; return f[0]
OURS = ("bnq_", "site1_", "slab_reduce", "site_prep", "qgemm")
rows = list(csv.reader(open(one("gpurun_out/prof_o/stats/**/*kernel_stats.csv"))))
with open(f"gpurun_out/profiles/{R}_office_kernel_stats.csv", "w") as fo:
    fo.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/office_shapes.py   (MI355X)\n")
    fo.write("# bench.py's kernels.office_shapes: bnq chains at [56,256,56,56] / [56,64,112,112], bottleneck tails at 2 x [28,100352] / 2 x [28,802816]\n")
    fo.write("# (4 rotating operand sets, 3 warm + 20 timed calls per chain); all shapes of a kernel are averaged together in this table\n")
    w = csv.writer(fo, quoting=csv.QUOTE_ALL); w.writerow(rows[0])
    w.writerows([r for r in rows[1:] if any(k in r[0] for k in OURS)])
def counters(which, name):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(one(f"gpurun_out/prof_o/{which}/**/*counter_collection.csv"))):
        if row.get("Counter_Name") != name: continue
        k = (re.sub(r"\(anonymous namespace\)::|alignq_site::|void ", "", row["Kernel_Name"])[:64], row.get("Grid_Size", ""))
        acc[k][0] += 1; acc[k][1] += float(row["Counter_Value"])
    return acc
fe, wr = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
with open(f"gpurun_out/profiles/{R}_office_pmc_hbm_bytes.csv", "w") as fo:
    fo.write("# HBM bytes per launch = 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, counters in 1 KiB units (MI355X_MICROARCH.md), separate\n")
    fo.write("# rocprofv3 --pmc passes of tools/office_shapes.py; one row per (kernel, grid size) = per shape\n")
    fo.write("kernel,grid,launches,fetch_MB_per_launch,write_MB_per_launch,total_MB_per_launch\n")
    for k in sorted(set(fe) | set(wr)):
        if not any(o in k[0] for o in OURS): continue
        nf, f = fe.get(k, [0, 0.0]); nw, w_ = wr.get(k, [0, 0.0])
        fmb = 2 * f * 1024 / max(nf, 1) / 1e6; wmb = w_ * 1024 / max(nw, 1) / 1e6
        fo.write(f"\"{k[0]}\",{k[1]},{max(nf, nw)},{fmb:.2f},{wmb:.2f},{fmb + wmb:.2f}\n")
print(open(f"gpurun_out/profiles/{R}_office_pmc_hbm_bytes.csv").read()[:3000])
PY
rm -rf gpurun_out/prof_o
bash tools/profile_office.sh > /dev/null 2>&1 || true
cp gpurun_out/prof_office/office_step_kernels.csv gpurun_out/profiles/${ROUND}_office_step_kernels_eager.csv || true
PMIN=500 PMAX=900 STEPS=30 REPS=12 EXTRA="--model resnet50_dann --batch 28" bash tools/step_timeline.sh > /dev/null 2>&1 || true
cp gpurun_out/step_timeline.txt gpurun_out/profiles/${ROUND}_office_step_timeline.txt || true
python3 tools/timeline_agg.py gpurun_out/profiles/${ROUND}_office_step_timeline.txt > gpurun_out/profiles/${ROUND}_office_step_by_kernel.txt || true
head -30 gpurun_out/profiles/${ROUND}_office_step_by_kernel.txt

#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel stats + the two PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs,
# MI355X_MICROARCH.md) of tools/roofline_shapes.py; condensed into gpurun_out/profiles/${ROUND}_shapes_*.
set -e
export ROUND=${ROUND:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_s && mkdir -p gpurun_out/prof_s gpurun_out/profiles
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s/stats -o run -- python3 tools/roofline_shapes.py > gpurun_out/prof_s/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_s/fetch -o run -- python3 tools/roofline_shapes.py > gpurun_out/prof_s/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_s/write -o run -- python3 tools/roofline_shapes.py > gpurun_out/prof_s/write.log 2>&1
python3 - <<'PY'
import os; R = os.environ["ROUND"]
import csv, glob, collections, json, re
def one(pat):
    f = glob.glob(pat, recursive=True); assert f, pat; return f[0]
OURS = ("site", "slab_reduce", "act_quant", "weight_", "bins_")
rows = list(csv.reader(open(one("gpurun_out/prof_s/stats/**/*kernel_stats.csv"))))
with open(f"gpurun_out/profiles/{R}_shapes_kernel_stats.csv", "w") as fo:
    fo.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/roofline_shapes.py   (MI355X)\n")
    fo.write("# shapes: site fwd/bwd at [128,524288] (site_fwd4<64,true,false,512>, site_bwd4<64,true,false,true,true>: the looped forms), [28,802816] and [28,100352]\n")
    fo.write("# (site1_*), weights [512,512,3,3], plain quantiser on 2^26 elements; 3 warm + 10-20 timed launches each\n")
    w = csv.writer(fo, quoting=csv.QUOTE_ALL); w.writerow(rows[0])
    w.writerows([r for r in rows[1:] if any(k in r[0] for k in OURS)])
# per-dispatch counters keyed by (kernel, grid) so that the three site shapes stay apart
def counters(which, name):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(one(f"gpurun_out/prof_s/{which}/**/*counter_collection.csv"))):
        if row.get("Counter_Name") != name: continue
        k = (re.sub(r"\(anonymous namespace\)::|alignq_site::|void ", "", row["Kernel_Name"])[:60], row.get("Grid_Size", ""))
        acc[k][0] += 1; acc[k][1] += float(row["Counter_Value"])
    return acc
fe, wr = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
with open(f"gpurun_out/profiles/{R}_shapes_pmc_hbm_bytes.csv", "w") as fo:
    fo.write("# rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 tools/roofline_shapes.py\n")
    fo.write("# KB per dispatch (mean); gfx950: fetch bytes = 2 x FETCH_SIZE for wide coalesced reads (MI355X_MICROARCH.md, HBM)\n")
    fo.write("kernel,grid_threads,dispatches,FETCH_SIZE_KB,fetch_corrected_MB,WRITE_SIZE_KB,write_MB,total_corrected_MB\n")
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1])):
        if not any(o in k[0] for o in OURS): continue
        nf, sf = fe.get(k, [0, 0.0]); nw, sw = wr.get(k, [0, 0.0])
        f_kb = sf / nf if nf else 0.0; w_kb = sw / nw if nw else 0.0
        fo.write('"%s",%s,%d,%.1f,%.2f,%.1f,%.2f,%.2f\n' % (k[0], k[1], max(nf, nw), f_kb, 2 * f_kb * 1024 / 1e6, w_kb, w_kb * 1024 / 1e6, (2 * f_kb + w_kb) * 1024 / 1e6))
PY
grep "^{" gpurun_out/prof_s/stats.log > gpurun_out/profiles/${ROUND}_shapes_under_rocprof.json || true
rm -rf gpurun_out/prof_s
cat gpurun_out/profiles/${ROUND}_shapes_pmc_hbm_bytes.csv | head -30

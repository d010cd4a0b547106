#!/bin/bash
# Runs ON THE GPU BOX: PMC bytes of the N2 pairs (tools/n2_bytes.py), FETCH_SIZE and WRITE_SIZE in separate passes.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_n2 && mkdir -p gpurun_out/prof_n2 gpurun_out/profiles
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_n2/fetch -o run -- python3 tools/n2_bytes.py > gpurun_out/prof_n2/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_n2/write -o run -- python3 tools/n2_bytes.py > gpurun_out/prof_n2/write.log 2>&1
python3 - <<'PY'
import csv, glob, re
def rows(which, name):
    f = glob.glob(f"gpurun_out/prof_n2/{which}/**/*counter_collection.csv", recursive=True)[0]
    out = []
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == name:
            out.append((int(r["Dispatch_Id"]), re.sub(r"\(anonymous namespace\)::|alignq_site::|void ", "", r["Kernel_Name"])[:48], float(r["Counter_Value"])))
    return sorted(out)
fe, wr = rows("fetch", "FETCH_SIZE"), rows("write", "WRITE_SIZE")
want = ("site_fwd4", "conv3x3_nhwc_kernel", "wgrad3x3_nhwc_kernel", "site_bwd4")
fe = [r for r in fe if any(w in r[1] for w in want)]
wr = [r for r in wr if any(w in r[1] for w in want)]
# drop the set-up site forward (first site_fwd4 dispatch), then groups of 3 identical launches
fe, wr = fe[1:], wr[1:]
labels = ["site fwd, fp32 out", "site fwd, int16 index out", "conv fwd, fp32 in", "conv fwd, int16 index in",
          "filter gradient, fp32 x", "filter gradient, int16 index x", "site bwd, mask from fp32 y", "site bwd, mask from int16 index"]
n = 128 * 16384
with open("gpurun_out/profiles/r02_n2_bytes.csv", "w") as fo:
    fo.write("# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python3 tools/n2_bytes.py   (B=128, C=16, 32x32, k=8)\n")
    fo.write("# mean of 3 launches; fetch_MB = 2 x FETCH_SIZE (gfx950 calibration for 16 B-per-lane reads; the 8 B-per-lane index reads may be over- or under-counted)\n")
    fo.write("launch,kernel,fetch_MB,write_MB,total_MB,bytes_per_element\n")
    for i, lab in enumerate(labels):
        f3, w3 = fe[3 * i:3 * i + 3], wr[3 * i:3 * i + 3]
        if len(f3) < 3 or len(w3) < 3: continue
        f_mb = 2 * sum(r[2] for r in f3) / 3 * 1024 / 1e6
        w_mb = sum(r[2] for r in w3) / 3 * 1024 / 1e6
        fo.write('"%s","%s",%.2f,%.2f,%.2f,%.2f\n' % (lab, f3[0][1], f_mb, w_mb, f_mb + w_mb, (f_mb + w_mb) * 1e6 / n))
print(open("gpurun_out/profiles/r02_n2_bytes.csv").read())
PY
rm -rf gpurun_out/prof_n2

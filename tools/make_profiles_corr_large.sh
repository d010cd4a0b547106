#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): tools/corr_large_bench.py (bench.py's kernels.corr_large) plain and under rocprofv3 kernel stats
# -> gpurun_out/profiles/${ROUND}_corr_large.txt (the bench's own lines, then the kernel table of the rows above 128)
set -e
export ROUND=${ROUND:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_c && mkdir -p gpurun_out/prof_c gpurun_out/profiles
python3 tools/corr_large_bench.py 2>/dev/null > gpurun_out/profiles/${ROUND}_corr_large.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c/stats -o run -- python3 tools/corr_large_bench.py > gpurun_out/prof_c/stats.log 2>&1
python3 - <<'PY'
import os; R = os.environ["ROUND"]
import csv, glob
f = glob.glob("gpurun_out/prof_c/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(f)))
with open(f"gpurun_out/profiles/{R}_corr_large.txt", "a") as fo:
    fo.write("\n# rocprofv3 --kernel-trace --stats -- python3 tools/corr_large_bench.py (all shapes of a kernel averaged together:\n")
    fo.write("# corr at [256,16384] and [1024,16384], the ADMM site at [256,16384], the fused site at [128,16384])\n")
    w = csv.writer(fo, quoting=csv.QUOTE_ALL); w.writerow(rows[0])
    w.writerows([r for r in rows[1:] if any(k in r[0] for k in ("corrl_", "sitel_", "admm_loss", "site_fwd4", "site_bwd4", "slab_reduce", "site_prep"))])
PY
rm -rf gpurun_out/prof_c
cat gpurun_out/profiles/${ROUND}_corr_large.txt | cut -c1-240

#!/usr/bin/env python3
"""Registers / scratch / LDS of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), optionally filtered:
    python tools/kernel_resources.py alignq_amd/csrc/site4_kernels.hip [substring of the mangled name]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", f"-I{root}/include",
       f"-I{root}/alignq_amd/csrc", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
for line in out.split("\n"):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        continue
    if cur is None:
        continue
    for key in ("VGPRs", "AGPRs", "ScratchSize \\[bytes/lane\\]", "Occupancy \\[waves/SIMD\\]", "VGPRs Spill", "LDS Size \\[bytes/block\\]"):
        m = re.search(r"remark: [^ ]* +" + key + r": (\d+)", line)
        if m:
            cur[key.split(" ")[0] if "Spill" not in key else "Spill"] = int(m.group(1))
    if "LDS" in cur:
        if pat in cur["name"]:
            sh = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            sh = re.sub(r"\(anonymous namespace\)::|alignq_site::", "", sh).split("(")[0]
            print(f"{sh:60s} vgpr {cur.get('VGPRs'):4d} agpr {cur.get('AGPRs', 0):3d} spill {cur.get('Spill', 0):3d} scratch {cur.get('ScratchSize', 0):4d} occ {cur.get('Occupancy')} lds {cur['LDS']}")
        cur = None

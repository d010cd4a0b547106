"""No kernel of the library spills registers or uses scratch memory (hipcc -Rpass-analysis=kernel-resource-usage over every
.hip file, cross-compiled for gfx950: no GPU needed).  A spill in a streaming kernel is scratch traffic in its inner loop; the
site kernels are sized against the 128 / 168 / 256 register steps on purpose."""
import concurrent.futures
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scan(src):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", f"-I{ROOT}/include",
           f"-I{ROOT}/alignq_amd/csrc", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out, name = [], None
    for line in err.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        for key in ("VGPRs Spill", r"ScratchSize \[bytes/lane\]"):
            m = re.search(r"remark: [^ ]* +" + key + r": (\d+)", line)
            if m and name:
                out.append((os.path.basename(src), name, key.split(" ")[0], int(m.group(1))))
    return out


def test_no_kernel_spills_or_uses_scratch():
    files = sorted(glob.glob(os.path.join(ROOT, "alignq_amd", "csrc", "*.hip")))
    assert len(files) >= 12
    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as pool:
        rows = [r for res in pool.map(_scan, files) for r in res]
    kernels = {(f, n) for f, n, _, _ in rows}
    assert len(kernels) > 100, len(kernels)          # the remarks were produced (every kernel reports both quantities)
    bad = [r for r in rows if r[3] != 0]
    assert not bad, bad

#!/usr/bin/env python3
"""Cost of ONE grid-wide barrier inside a kernel (agent-scope ticket + spin, co-resident workgroups) against the ~4.8 us a
dependent HIP-graph node costs - the measurement VERDICT r3 item 5 asks for before any kernel is made to span a layer
boundary.  Builds tools/src/grid_barrier.hip if the binary is missing (hipcc cross-compiles; the binary travels with gpurun),
runs it on the GPU box and writes gpurun_out/grid_barrier.json.

    gpurun -- python3 tools/grid_barrier_cost.py
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, BIN = os.path.join(ROOT, "tools", "src", "grid_barrier.hip"), os.path.join(ROOT, "tools", "bin", "grid_barrier")


def main():
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(BIN), exist_ok=True)
        subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", SRC, "-o", BIN], check=True)
    if "--build-only" in sys.argv:
        return
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    sys.stderr.write(r.stderr)
    res = json.loads(r.stdout)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "grid_barrier.json"), "w") as f:
        json.dump(res, f, indent=1)
    for e in res["results"]:
        print(f"{e['form']:5s} {e['workgroups']:5d} x {e['threads']:4d}: {e['us_per_barrier']:.3f} us per barrier "
              f"(empty kernel {e['empty_kernel_us']:.1f} us)")
    print("spin timeouts:", res["spin_timeouts"])
    sys.exit(r.returncode)


if __name__ == "__main__":
    main()

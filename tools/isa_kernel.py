#!/usr/bin/env python3
"""Print (or summarise) the gfx950 ISA of one kernel from a hipcc -save-temps .s file.

    python tools/isa_kernel.py <file.s> <substring of the mangled kernel name> [--count]

--count prints, per basic block, the number of vector-ALU, LDS, vector-memory and scalar instructions, so the
instructions per element of a streaming loop can be read off (NOTES.md quotes these numbers)."""
import re
import sys


def kernels(text):
    out, name, body = {}, None, []
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            name, body = m.group(1), []
            out[name] = body
        elif name is not None:
            body.append(line)
            if line.startswith(".Lfunc_end"):
                name = None
    return out


def main():
    path, pat = sys.argv[1], sys.argv[2]
    ks = {k: v for k, v in kernels(open(path).read()).items() if pat in k}
    for k, body in ks.items():
        print("==", k)
        if "--count" not in sys.argv:
            print("\n".join(body))
            continue
        blk, cnt = "entry", {}
        for line in body:
            m = re.match(r"^(\.LBB\w+):", line)
            if m:
                blk = m.group(1)
                continue
            t = line.strip().split(" ")[0].split("\t")[0]
            if not t or t.startswith(";") or t.startswith("."):
                continue
            c = cnt.setdefault(blk, {"valu": 0, "lds": 0, "vmem": 0, "salu": 0, "other": 0})
            if t.startswith("v_"):
                c["valu"] += 1
            elif t.startswith("ds_"):
                c["lds"] += 1
            elif t.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["vmem"] += 1
            elif t.startswith("s_"):
                c["salu"] += 1
            else:
                c["other"] += 1
        for b, c in cnt.items():
            print(f"  {b:14s} " + " ".join(f"{n}={v}" for n, v in c.items()))


if __name__ == "__main__":
    main()

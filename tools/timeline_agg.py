#!/usr/bin/env python3
"""Aggregate a step timeline (tools/step_timeline.sh output) by kernel family: python3 tools/timeline_agg.py file [file2 for a diff]"""
import collections
import re
import sys


def agg(fn):
    a = collections.defaultdict(lambda: [0, 0.0])
    for l in open(fn):
        if l.startswith('#'):
            continue
        m = re.match(r'\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+(.*)', l)
        if not m:
            continue
        n = re.sub(r'\(.*', '', m.group(4))[:60]
        if n.startswith('igemm'):
            n = n[:9]
        if n.startswith('_ZN2ck'):
            n = 'ck_conv'
        if n.startswith('at::native'):
            n = n[:58]
        if n.startswith('Cijk'):
            n = 'rocblas'
        a[n][0] += 1
        a[n][1] += float(m.group(2))
    return a


a = agg(sys.argv[1])
b = agg(sys.argv[2]) if len(sys.argv) > 2 else None
tot = sum(v[1] for v in a.values())
keys = sorted(set(a) | set(b or {}), key=lambda k: -(a.get(k, [0, 0])[1] + (b or {}).get(k, [0, 0])[1]))
for k in keys[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    c, d = a.get(k, [0, 0.0])
    if b is None:
        print(f"{d:9.1f} us {100 * d / tot:5.1f}% {c:4d}  {k}")
    else:
        c2, d2 = b.get(k, [0, 0.0])
        print(f"{d:9.1f} -> {d2:9.1f} us ({d2 - d:+8.1f}) {c:4d} -> {c2:4d}  {k}")
print("total", tot, (sum(v[1] for v in b.values()) if b else ""))

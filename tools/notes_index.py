#!/usr/bin/env python3
"""Regenerates the index at the top of NOTES.md (between the INDEX markers): one line per experiment of the "tried and not kept" /
"measured, tried" sections - NOTES.md line, round, kernel names (back-quoted identifiers) and the entry's first words.
    python3 tools/notes_index.py        (idempotent; run after editing NOTES.md)"""
import os
import re

P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "NOTES.md")
BEGIN, END = "<!-- INDEX:BEGIN -->", "<!-- INDEX:END -->"
text = open(P).read()
if BEGIN in text:
    text = text[:text.index(BEGIN)] + text[text.index(END) + len(END) + 1:]
lines = text.split("\n")
sect = re.compile(r"^#{2,3} .*(tried|Tried)")
entries, in_sec, cur = [], False, None
for i, l in enumerate(lines):
    if l.startswith("#"):
        if cur:
            entries.append(cur); cur = None
        in_sec = bool(sect.match(l))
        sec_name = l.lstrip("# ").strip()
        continue
    if not in_sec:
        continue
    if l.startswith("* "):
        if cur:
            entries.append(cur)
        cur = [i, sec_name, l[2:]]
    elif cur and l.startswith("  "):
        cur[2] += " " + l.strip()
    elif cur and not l.strip():
        entries.append(cur); cur = None
if cur:
    entries.append(cur)
out = [BEGIN, "## Index of experiments (tried and not kept / measured) — regenerate with `python3 tools/notes_index.py`", "",
       "| line | round | kernels / identifiers | entry |", "|---|---|---|---|"]
head_len = None
rows = []
for i, sec, body in entries:
    m = re.match(r"\(?(round \d[^)]*)\)", body)
    rnd = m.group(1) if m else (re.search(r"[Rr]ound \d", sec).group(0).lower() if re.search(r"[Rr]ound \d", sec) else "")
    ids = []
    for t in re.findall(r"`([^`]+)`", body):
        t = t.split("(")[0].split("<")[0].strip()
        if re.match(r"^[A-Za-z_][\w.:/-]*$", t) and t not in ids and len(t) > 3:
            ids.append(t)
    title = re.sub(r"\*\*|`", "", body)
    title = re.sub(r"^\(?round \d[^)]*\)\s*", "", title)[:120].rstrip()
    rows.append((i, rnd, ", ".join(ids[:4]), title.replace("|", "/")))
n_index = len(out) + len(rows) + 2
# NOTES.md starts with its title block; the index goes right behind the first heading's paragraph (before the second heading)
second = next(k for k, l in enumerate(lines) if k > 0 and l.startswith("#"))
for i, rnd, ids, title in rows:
    out.append(f"| {i + 1 + n_index} | {rnd} | {ids} | {title} |")
out += ["", END]
lines[second:second] = out
open(P, "w").write("\n".join(lines))
print(f"{len(rows)} entries indexed")

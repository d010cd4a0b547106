#!/usr/bin/env python3
"""Basic-block map of one kernel of a -save-temps .s file: per block the memory operations and where the waits on the
vector-memory counter sit (a `vmcnt(0)` right behind a lone load = a serialised round trip).
    python tools/isa_blocks.py <file.s> <substring of the mangled kernel name> [min instructions to list a block]"""
import re, sys, collections
text = open(sys.argv[1]).read()
name = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 30
i = text.index(name); i = text.index(':', i)
body = text[i:text.index('.Lfunc_end', i)]
blk = 'entry'; per = collections.defaultdict(collections.Counter); order = ['entry']; notes = collections.defaultdict(list); n = 0
for line in body.split('\n'):
    m = re.match(r'^(\.LBB\w+):', line)
    if m:
        blk = m.group(1); order.append(blk); n = 0; continue
    t = line.strip()
    if not t or t.startswith(';'): continue
    n += 1
    per[blk][t.split()[0]] += 1
    if 'vmcnt' in t or t.startswith('s_barrier'):
        notes[blk].append(f"{n}:{'barrier' if t.startswith('s_barrier') else re.search(r'vmcnt.[0-9]+.', t).group(0)}")
print(name, 'blocks', len(order))
for b in order:
    c = per[b]; tot = sum(c.values())
    gl = sum(v for k, v in c.items() if k.startswith(('global_load', 'buffer_load')))
    gs = sum(v for k, v in c.items() if k.startswith(('global_store', 'buffer_store')))
    if tot > minn or gl or gs or notes[b]:
        print(f"  {b:12s} {tot:4d} ld {gl:2d} st {gs:2d} valu {sum(v for k, v in c.items() if k.startswith('v_')):4d} lds {sum(v for k, v in c.items() if k.startswith('ds_')):3d} mfma {sum(v for k, v in c.items() if k.startswith('v_mfma')):3d}  {' '.join(notes[b][:10])}")

#!/usr/bin/env python3
"""tools/isa_blocks.py file.s kernel-substring -- basic blocks of one kernel of a gfx950 assembly listing: instructions, multiply-adds, scratch
(spill) accesses, LDS / global accesses, waits, branch targets, loop nesting comment.  Static counts."""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r"^(_Z\w*%s\w*):[^\n]*\n" % re.escape(sys.argv[2]), s, flags=re.M)
start = m.end()
end = s.index("s_endpgm", start)
blocks, cur = [], ["entry", "", []]
for l in s[start:end].split("\n"):
    mm = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?", l)
    if mm:
        blocks.append(cur)
        cur = [mm.group(1), (mm.group(2) or "").strip(), []]
    elif l.startswith("\t") and not l.strip().startswith((".", ";")):
        cur[2].append(l.strip())
blocks.append(cur)
print(m.group(1))
for name, note, ins in blocks:
    br = [i.split()[-1] for i in ins if i.startswith(("s_cbranch", "s_branch"))]
    calls = sum(1 for i in ins if i.startswith("s_swappc"))
    print(f"{name:10s} n={len(ins):5d} mad={sum(i.startswith('v_mad_u64') for i in ins):5d} scratch={sum('scratch' in i for i in ins):3d} ds={sum(i.startswith('ds_') for i in ins):3d} "
          f"global={sum(i.startswith('global_') for i in ins):3d} waits={sum(i.startswith('s_waitcnt') for i in ins):2d} calls={calls} br={br} {note}")

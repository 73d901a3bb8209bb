#!/usr/bin/env python3
"""tools/isa_stats.py file.s [substring ...] -- per-kernel instruction statistics of a gfx950 assembly listing (hipcc -save-temps):
total instructions, multiply-adds, scratch (spill) accesses, AGPR moves, s_nop, waits.  Static counts, not executed counts."""
import re
import sys
from collections import Counter


def kernels(text):
    cur, body = None, []
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            if line.strip().startswith("s_endpgm"):
                yield cur, body
                cur = None
                continue
            if line.startswith("\t") and not line.strip().startswith((".", ";")):
                body.append(line.strip())


def main():
    text = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    for name, body in kernels(text):
        if pats and not any(p in name for p in pats):
            continue
        c = Counter(l.split()[0] for l in body)
        scratch = {k: v for k, v in c.items() if "scratch" in k}
        print(f"{name}\n   instructions {len(body)}  v_mad_u64_u32 {c.get('v_mad_u64_u32', 0)}  scratch {sum(scratch.values())} {scratch}  "
              f"accvgpr {c.get('v_accvgpr_write_b32', 0) + c.get('v_accvgpr_read_b32', 0)}  s_nop {c.get('s_nop', 0)}  s_waitcnt {c.get('s_waitcnt', 0)}  "
              f"global_load_lds {c.get('global_load_lds_dwordx4', 0)}")


if __name__ == "__main__":
    main()

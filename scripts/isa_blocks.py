#!/usr/bin/env python3
"""per-basic-block instruction census of one kernel's assembly (scripts/isa_count.sh dumps /tmp/isa_*.s):
   isa_blocks.py file.s  ->  block label, #valu, #salu, #vmem, terminator"""
import re, sys
blk = "entry"; rows = []; cur = dict(v=0, s=0, m=0, w=0, term="")
for line in open(sys.argv[1]):
    m = re.match(r"^(\.LBB\S+):", line)
    if m:
        rows.append((blk, cur)); blk = m.group(1); cur = dict(v=0, s=0, m=0, w=0, term=""); continue
    t = line.strip().split()
    if not t or t[0].startswith((";", ".")): continue
    op = t[0]
    if op.startswith("v_"): cur["v"] += 1
    elif op.startswith("s_waitcnt"): cur["w"] += 1
    elif op.startswith(("s_cbranch", "s_branch")): cur["term"] += " %s->%s" % (op[2:], t[1])
    elif op.startswith("s_"): cur["s"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_", "ds_")): cur["m"] += 1
rows.append((blk, cur))
for b, c in rows:
    print("%-12s valu %4d salu %4d mem %3d wait %2d %s" % (b, c["v"], c["s"], c["m"], c["w"], c["term"]))

#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing.  usage: isa_blocks.py file.s kernel-substring [min_valu]"""
import re
import sys

src, key = sys.argv[1], sys.argv[2]
minv = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
cur, blocks = None, []
for l in lines[start + 1:]:
    if "s_endpgm" in l:
        break
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
    if m:
        cur = dict(name=m.group(1), hdr=m.group(2).strip(), valu=0, salu=0, lds=0, vmem=0, div=0, mov=0, cmp=0, depth="")
        d = re.search(r"Depth=(\d+)", m.group(2))
        cur["depth"] = d.group(1) if d else ""
        blocks.append(cur)
        continue
    if cur is None:
        continue
    t = l.strip()
    if not cur["depth"]:
        d = re.search(r"Depth=(\d+)", t) if t.startswith(";") else None
        if d:
            cur["depth"] = d.group(1)
    if t.startswith("v_"):
        cur["valu"] += 1
        if t.startswith(("v_div_", "v_rcp")):
            cur["div"] += 1
        if t.startswith(("v_mov", "v_cndmask", "v_pk_mov")):
            cur["mov"] += 1
        if t.startswith("v_cmp"):
            cur["cmp"] += 1
    elif t.startswith("s_"):
        cur["salu"] += 1
    elif t.startswith("ds_"):
        cur["lds"] += 1
    elif t.startswith(("global_", "scratch_", "buffer_", "flat_")):
        cur["vmem"] += 1
for b in blocks:
    if b["valu"] >= minv:
        print("%-10s depth %-2s valu %3d (mov %2d cmp %2d div %2d) salu %3d lds %2d vmem %2d" % (
            b["name"], b["depth"], b["valu"], b["mov"], b["cmp"], b["div"], b["salu"], b["lds"], b["vmem"]))

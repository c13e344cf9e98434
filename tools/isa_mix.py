#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -save-temps .s file (blocks that contain MFMAs first).
    python tools/isa_mix.py file.s <kernel-name-substring> [min_mfma]"""
import collections, re, sys
src, key = sys.argv[1], sys.argv[2]
minm = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(";")[0].strip().endswith(":"))
blocks, cur, name = [], collections.Counter(), "entry"
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith(".Lfunc_end"):
        break
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append((name, cur)); cur, name = collections.Counter(), t.split(":")[0]
        continue
    if not t or t.startswith((";", ".")):
        continue
    op = t.split()[0]
    if op.startswith("v_mfma"): k = "mfma"
    elif op.startswith("v_accvgpr"): k = "accvgpr"
    elif op.startswith("v_"): k = "valu"
    elif op.startswith("ds_"): k = "lds"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): k = "vmem"
    elif op.startswith("s_waitcnt"): k = "waitcnt"
    elif op.startswith("s_nop"): k = "nop"
    elif op.startswith("s_"): k = "salu"
    else: k = "other"
    cur[k] += 1
blocks.append((name, cur))
for n, c in blocks:
    if c["mfma"] >= minm:
        tot = sum(c.values())
        print(n, dict(c), "valu/mfma=%.2f" % ((c["valu"] + c["accvgpr"]) / max(c["mfma"], 1)), "total", tot)

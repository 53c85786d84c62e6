#!/usr/bin/env python3
"""Static branches of one kernel in a hipcc -S listing with their direction and distance
(development aid: a taken branch costs a lone wave ~26-33 clocks, tools/probe/branch_fetch_probe).
    python tools/asm_branches.py gpurun_out/asm/cur.s 'nuts_kernelINS_12EightSchoolsILi16'"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
needle = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and needle in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
labels, seq = {}, []
for l in lines[start:end]:
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        labels[m.group(1)] = len(seq)
        continue
    m = re.match(r"^\t([a-z_0-9]+)\s*(.*)", l)
    if m and not l.startswith("\t."):
        seq.append((m.group(1), m.group(2)))
print("instructions", len(seq))
br = [(i, op, arg) for i, (op, arg) in enumerate(seq) if op.startswith("s_cbranch") or op == "s_branch"]
print("static branches", len(br))
for i, op, arg in br:
    t = labels.get(arg.split()[0], -1)
    print("%5d %-18s -> %5d  (%+d)" % (i, op, t, t - i))

#!/usr/bin/env python3
"""Instruction census of the loops of a kernel in hipcc's -save-temps assembly (tools/asm_loops.py <file.s> [kernel-substring])."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "lnprob_kernelILb0E"
m = re.search(r"^(_Z\w*" + re.escape(pat) + r"\w*):", s, re.M)
start = m.start()
end = s.index(".Lfunc_end", start)
body = s[start:end].splitlines()
labels = {}
for i, l in enumerate(body):
    mm = re.match(r"^(\.LBB\d+_\d+):", l)
    if mm:
        labels[mm.group(1)] = i
seen = set()
for i, l in enumerate(body):
    mm = re.search(r"\b(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
    if mm and mm.group(2) in labels and labels[mm.group(2)] < i and mm.group(2) not in seen:
        seen.add(mm.group(2))
        a = labels[mm.group(2)]
        ins = [x.strip().split()[0] for x in body[a:i + 1] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
        c = Counter()
        for x in ins:
            if x.startswith("v_") and "f64" in x:
                c["v_f64"] += 1
            elif x.startswith("v_"):
                c["v_other"] += 1
            elif x.startswith("s_"):
                c["s_"] += 1
            elif x.startswith("ds_"):
                c["ds_"] += 1
            elif x.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["mem"] += 1
            else:
                c["other"] += 1
        print(mm.group(2), "instrs", len(ins), dict(c))

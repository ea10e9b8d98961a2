#!/usr/bin/env python3
"""Static instruction census of one kernel by SOURCE LINE (developer tool, CPU only).

    python tools/asm_by_line.py [kernel-substring] [--top N] [--file mp_eval.hpp]

Compiles mp_kernels.hip with -gline-tables-only to assembly and attributes every instruction of the chosen kernel to the
innermost source location of its .loc directive.  Inlined helpers (mp_math.hpp) are reported under their own lines; the
second table sums by source-line ranges of walker_eval's sections.  Static counts: code inside the tile loop runs once per
tile, code inside the sweep loop ~3 times per tile."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "magprop_amd", "csrc", "mp_kernels.hip")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pat = args[0] if args else "lnprob_kernelILb0ELi4ELb0E"
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-gline-tables-only", "-x", "hip",
                        "-S", "--cuda-device-only", SRC, "-o", out], check=True, stderr=subprocess.DEVNULL, cwd=td)
        s = open(out).read()
    files = {int(m.group(1)): m.group(3) for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', s)}
    files.update({int(m.group(1)): os.path.basename(m.group(2)) for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"\s+md5', s)})
    m = re.search(r"^(_Z\w*" + re.escape(pat) + r"\w*):", s, re.M)
    body = s[m.start():s.index(".Lfunc_end", m.start())].splitlines()
    cur = ("?", 0)
    by_line = collections.Counter()
    kinds = collections.defaultdict(collections.Counter)
    for l in body:
        t = l.strip()
        mm = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if mm:
            cur = (files.get(int(mm.group(1)), mm.group(1)), int(mm.group(2)))
            continue
        if not l.startswith("\t") or t.startswith((".", ";")) or not t:
            continue
        op = t.split()[0]
        k = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "mem")
        by_line[cur] += 1
        kinds[cur][k] += 1
    total = sum(by_line.values())
    print(f"{pat}: {total} instructions")
    byfile = collections.Counter()
    for (f, ln), n in by_line.items():
        byfile[f] += n
    print("by file:", dict(byfile))
    print(f"--- top {top} source lines")
    for (f, ln), n in by_line.most_common(top):
        print(f"{f}:{ln:5d}  {n:5d}  {dict(kinds[(f, ln)])}")
    # mp_math.hpp by function (line ranges)
    src = open(os.path.join(ROOT, "magprop_amd", "csrc", "mp_math.hpp")).read().splitlines()
    fn_at = {}
    name = "?"
    for i, l in enumerate(src, 1):
        mm = re.match(r"MP_DEV\s+[\w<>:, ]+?\s+(\w+)\(", l.strip())
        if mm:
            name = mm.group(1)
        fn_at[i] = name
    byfn = collections.Counter()
    for (f, ln), n in by_line.items():
        if f == "mp_math.hpp":
            byfn[fn_at.get(ln, "?")] += n
    print("--- mp_math.hpp by function:", dict(byfn.most_common()))


if __name__ == "__main__":
    main()

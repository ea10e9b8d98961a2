#!/usr/bin/env python3
"""Static census of EXPOSED LDS / scalar-memory round trips of one kernel (developer tool, CPU only).

    python tools/wait_census.py [kernel-substring] [--top N] [--asm file.s]

With one wavefront per SIMD nothing hides a memory round trip: an `s_waitcnt lgkmcnt(n)` that follows its ds_read / s_load
after a handful of instructions stalls the wave for the rest of the ~64 (LDS) / ~100+ (scalar cache) cycles.  For every such
wait this lists the source line of the wait, the line of the oldest operation it waits for, and the issue slots in between
(VALU ~4.5 cycles, everything else ~1): the estimate of what is exposed is max(0, latency - slots).  Static: code in the tile
loop runs once per tile (8 per walker near the truth), code in the sweep loop ~2.6 times per tile.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "magprop_amd", "csrc", "mp_kernels.hip")
LAT = {"ds": 64.0, "s": 110.0}


def main():
    argv = sys.argv[1:]
    args = [a for i, a in enumerate(argv) if not a.startswith("--") and (i == 0 or argv[i - 1] not in ("--top", "--asm"))]
    pat = args[0] if args else "lnprob_kernelILb0ELi4ELb0ELb0E"
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 50
    if "--asm" in sys.argv:
        s = open(sys.argv[sys.argv.index("--asm") + 1]).read()
    else:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-gline-tables-only", "-x", "hip",
                            "-S", "--cuda-device-only", SRC, "-o", out], check=True, stderr=subprocess.DEVNULL, cwd=td)
            s = open(out).read()
    files = {int(m.group(1)): os.path.basename(m.group(3)) for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', s)}
    m = re.search(r"^(_Z\w*" + re.escape(pat) + r"\w*):", s, re.M)
    body = s[m.start():s.index(".Lfunc_end", m.start())].splitlines()
    cur = ("?", 0)
    outstanding = []     # (kind, loc, slot index at issue) in issue order; LDS returns in order, scalar loads may not: lgkmcnt(0) only
    slots = 0.0
    sites = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    n_wait = n_ds = n_s = 0
    for l in body:
        t = l.strip()
        mm = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if mm:
            cur = (files.get(int(mm.group(1)), mm.group(1)), int(mm.group(2)))
            continue
        if not l.startswith("\t") or t.startswith((".", ";")) or not t:
            if re.match(r"^\.LBB", l):      # a label: what is outstanding across it is unknown; keep (conservative)
                pass
            continue
        op = t.split()[0]
        if op.startswith("ds_") and not op.startswith(("ds_write", "ds_swizzle", "ds_bpermute", "ds_permute")):
            outstanding.append(("ds", cur, slots)); n_ds += 1
            slots += 1.0
        elif op.startswith(("ds_write", "ds_bpermute", "ds_permute", "ds_swizzle")):
            outstanding.append(("dsw", cur, slots))
            slots += 1.0
        elif op.startswith(("s_load", "s_buffer_load")):
            outstanding.append(("s", cur, slots)); n_s += 1
            slots += 1.0
        elif op == "s_waitcnt":
            mm = re.search(r"lgkmcnt\((\d+)\)", t)
            if mm:
                n = int(mm.group(1))
                done = outstanding[:max(0, len(outstanding) - n)] if n else outstanding
                waited = [o for o in done if o[0] in ("ds", "s")]
                if waited:
                    exposed = max(max(0.0, LAT[k] - (slots - at)) for k, _, at in waited)
                    last = waited[-1]
                    n_wait += 1
                    st = sites[cur]
                    st[0] += 1
                    st[1] += exposed
                    st[2][(last[0], last[1])] += 1
                outstanding = outstanding[len(done):] if n else []
            slots += 1.0
        elif op.startswith("v_"):
            slots += 8.0 if op.endswith("_f64") and ("rcp" in op or "rsq" in op or "sqrt" in op) else 4.5
        elif op == "s_barrier":
            outstanding = []
            slots += 1.0
        else:
            slots += 1.0
    print(f"{pat}: {n_ds} LDS reads, {n_s} scalar loads, {n_wait} waits on them (static)")
    print(f"{'wait at':28s} {'count':>5s} {'exposed cycles (static sum)':>28s}   waits for (last op issued)")
    for loc, (c, e, srcs) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:top]:
        src = ", ".join(f"{k}@{f}:{ln} x{n}" for ((k, (f, ln)), n) in srcs.most_common(3))
        print(f"{loc[0] + ':' + str(loc[1]):28s} {c:5d} {e:28.0f}   {src}")
    print(f"total exposed (static, every site once): {sum(v[1] for v in sites.values()):.0f} cycles")


main()

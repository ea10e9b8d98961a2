#!/usr/bin/env python3
"""Do the collective's kernels overlap the lnprob kernels in time?  (GPU box, after
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --always-gather --overlap 1 --no-cpu-baseline --no-extra --no-mcmc --steps 200)
    python tools/overlap_from_trace.py DIR
Reads the kernel trace (start / end time stamps per dispatch) and reports, for every kernel that is not mp::lnprob_kernel
(the RCCL all-gather's kernels in a group of one), how much of its duration lies inside an lnprob kernel's interval, and the
gap between consecutive lnprob kernels with and without a collective kernel between them."""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
K.sort()
ln = [(a, b) for a, b, n in K if "lnprob_kernel" in n]
ln = ln[len(ln) // 4:]                       # the timed region and what follows (skip spin-up)
t_lo, t_hi = ln[0][0], ln[-1][1]
others = defaultdict(list)
for a, b, n in K:
    if "lnprob_kernel" in n or b < t_lo or a > t_hi:
        continue
    inside = sum(max(0, min(b, lb) - max(a, la)) for la, lb in ln if lb > a and la < b)
    others[n.split("(")[0][-70:]].append((b - a, inside))
print(f"{len(ln)} lnprob kernels, mean duration {sum(b - a for a, b in ln) / len(ln) / 1e3:.1f} us")
for n, v in others.items():
    dur = sum(x for x, _ in v)
    ins = sum(y for _, y in v)
    print(f"  {n}: {len(v)} dispatches, mean {dur / len(v) / 1e3:.2f} us, {100.0 * ins / max(dur, 1):.0f} % of their time inside an lnprob kernel's interval")
gaps = [ln[i + 1][0] - ln[i][1] for i in range(len(ln) - 1)]
gaps.sort()
print(f"gap between consecutive lnprob kernels: median {gaps[len(gaps) // 2] / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us, 90th percentile {gaps[int(0.9 * len(gaps))] / 1e3:.2f} us")
starts = [ln[i + 1][0] - ln[i][0] for i in range(len(ln) - 1)]
print(f"start-to-start: mean {sum(starts) / len(starts) / 1e3:.2f} us")

#!/usr/bin/env python3
"""Developer diagnostic (GPU box): when a coarse tile is given up early (mp_eval.hpp abort_tile), by how much had its first
lanes exceeded the bound, and what did the next attempt (one stride finer) do?  Basis of the rule that skips strides.
    make -C magprop_amd/csrc abort-study && MAGPROP_AMD_LIB=$PWD/magprop_amd/libmagprop_amd_abort.so python tools/abort_study.py [n]
(the shipped library records only WHETHER the excess was beyond the skip threshold: 255 or 0 in the word's lane field)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magprop_amd import LogProb
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
lo, hi = gs["prior_lower"], gs["prior_upper"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(11)
lp = LogProb(gs["Humped_x"], gs["Humped_y"], gs["Humped_yerr"])
lp.handle.tile_log(True)
rows = []          # (kind, sweep at abort, log2 ratio, next tile: kind, kept lanes, aborted?)
for rep in range(4):
    X = lo + (hi - lo) * rng.random((n, 6))
    out, st = lp.handle.lnprob_batch(X, want_status=True)
    for i in range(n):
        log = lp.handle.last_tile_log(i)
        for a, b in zip(log[:-1], log[1:]):
            k, s_, l, w = a
            if w == 0x40:
                rows.append((k, s_, l / 8.0, b[0], b[2], b[3] == 0x40))
rows = np.array(rows, float)
print(f"{len(rows)} aborted tiles in {4 * n} prior-wide walkers")
for kind in (4, 3, 2):
    r = rows[rows[:, 0] == kind]
    if not len(r):
        continue
    print(f"kind {'P1248'[kind]}: {len(r)} aborts")
    for lo2, hi2 in ((0, 2), (2, 5), (5, 7), (7, 10), (10, 15), (15, 40)):
        m = (r[:, 2] >= lo2) & (r[:, 2] < hi2)
        if m.sum() == 0:
            continue
        nxt = r[m]
        same_next = nxt[:, 3] == kind - 1
        useless = same_next & ((nxt[:, 5] == 1) | (nxt[:, 4] < 8))
        print(f"   log2(excess) in [{lo2:2d}, {hi2:2d}): {m.sum():5d} tiles; next tile one stride finer: {same_next.sum():5d}, of which it too kept nothing: {useless.sum():5d} "
              f"({100.0 * useless.sum() / max(same_next.sum(), 1):.0f} %), kept lanes median {np.median(nxt[same_next & ~useless][:, 4]) if (same_next & ~useless).any() else 0:.0f}")

"""Diagnostic (GPU box): the tile sequence (kind:sweeps:lanes kept) of the slowest prior-wide walkers and of a near-truth one."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magprop_amd import LogProb
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
lo, hi = gs["prior_lower"], gs["prior_upper"]
lp = LogProb(gs["Humped_x"], gs["Humped_y"], gs["Humped_yerr"])
lp.handle.tile_log(True)
rng = np.random.default_rng(1)
P = np.array([1.0, 5.0, -3.0, 2.0, -1.0, 0.0]) + 1e-4 * rng.standard_normal((1024, 6))
wide = lo + (hi - lo) * rng.random((1024, 6))
def fmt(log): return " ".join(f"{'P1248'[k]}:{s}:{l}:{w:x}" for k, s, l, w in log)
lp(P); print("near-truth walker 0:", fmt(lp.handle.last_tile_log(0)))
out, st = lp.handle.lnprob_batch(wide, want_status=True)
sw = lp.handle.last_sweeps(1024)
hist = {}
for i in range(1024):
    if st[i] != 0: continue
    for k, s, l, w_ in lp.handle.last_tile_log(i):
        hist.setdefault(k, []).append(s)
print("prior-wide: sweeps per tile by kind:", {('P1248'[k]): (len(v), round(float(np.mean(v)), 2), int(np.max(v))) for k, v in sorted(hist.items())})
for i in np.argsort(sw)[::-1][:6]:
    print(f"walker {i} status {st[i]} sweeps {sw[i]} pars {np.round(wide[i], 3).tolist()}:", fmt(lp.handle.last_tile_log(i)))

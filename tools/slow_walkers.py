#!/usr/bin/env python3
"""Which walkers make a prior-wide launch slow?  Per-walker Newton sweeps and single-walker kernel times over the prior box
(GPU box): python tools/slow_walkers.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magprop_amd import LogProb, synth  # noqa: E402

g = np.load(os.path.join(ROOT, "tests/golden/golden_synth.npz"))
lp = LogProb(g["Humped_x"], g["Humped_y"], g["Humped_yerr"])
rng = np.random.default_rng(20261004)
n = 1024
P = synth.PRIOR_LOWER + (synth.PRIOR_UPPER - synth.PRIOR_LOWER) * rng.random((n, 6))
out, st = lp.handle.lnprob_batch(P, want_status=True)
sw = lp.handle.last_sweeps(n)
tiles = 40
ok = st == 0
print("status counts", np.bincount(st, minlength=4), " sweeps/tile: mean %.2f  median %.2f  p90 %.2f  p99 %.2f  max %.2f" % (
    sw[ok].mean() / tiles, np.median(sw[ok]) / tiles, np.quantile(sw[ok], 0.9) / tiles, np.quantile(sw[ok], 0.99) / tiles, sw[ok].max() / tiles))
# time groups of 1024 copies of single walkers: cheapest, median, most expensive by sweeps
order = np.argsort(np.where(ok, sw, -1))
picks = {"min sweeps": order[np.nonzero(ok[order])[0][0]], "median": order[len(order) // 2], "p90": order[int(0.9 * n)],
         "p99": order[int(0.99 * n)], "max sweeps": order[-1]}
for name, i in picks.items():
    Q = np.tile(P[i], (n, 1))
    lp.handle.lnprob_batch(Q)
    t0 = time.perf_counter()
    for _ in range(10):
        lp.handle.lnprob_batch(Q)
    dt = (time.perf_counter() - t0) / 10
    print(f"{name:11s} walker {i:4d} sweeps/tile {sw[i] / tiles:5.2f}  1024 copies: {dt * 1e3:.3f} ms per launch (host entry)   pars {np.round(P[i], 3)}")

# every walker on its own (1024 copies each): cost against parameters
import torch  # noqa: E402
dev = torch.device("cuda", lp.handle.device)
cost = np.zeros(n)
outt = torch.empty(n, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream(dev)
for i in range(n):
    Q = torch.from_numpy(np.tile(P[i], (n, 1))).to(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lp.handle.lnprob_batch_dev(Q.data_ptr(), n, 6, outt.data_ptr(), stream=stream.cuda_stream)
    e0.record(stream)
    lp.handle.lnprob_batch_dev(Q.data_ptr(), n, 6, outt.data_ptr(), stream=stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize(dev)
    cost[i] = e0.elapsed_time(e1)
o = np.argsort(cost)
print("per-walker kernel ms (1024 copies): min %.3f  median %.3f  p90 %.3f  p99 %.3f  max %.3f" % (
    cost[o[0]], np.median(cost), np.quantile(cost, 0.9), np.quantile(cost, 0.99), cost[o[-1]]))
names = ["B", "P", "lgMdisc", "lgRdisc", "lgeps", "lgdelta"]
print("correlation of cost with parameters:", {k: round(float(np.corrcoef(P[ok, j], cost[ok])[0, 1]), 2) for j, k in enumerate(names)},
      "with sweeps", round(float(np.corrcoef(sw[ok], cost[ok])[0, 1]), 2))
for i in o[-8:]:
    print(f"  walker {i:4d} {cost[i]:.3f} ms  sweeps/tile {sw[i] / tiles:.2f} status {st[i]} pars {np.round(P[i], 3)}")

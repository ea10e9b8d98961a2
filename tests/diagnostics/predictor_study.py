#!/usr/bin/env python3
"""How good is the first guess of a tile?  (CPU; developer study behind the predictor of mp_eval.hpp.)
Trajectories of the serial C oracle for the four canonical parameter sets and 40 prior-wide ones; at tile starts spread over
the grid the values of omega at the n step ends of a tile of the given stride are predicted from the history at the tile's
spacing and compared with the trajectory:
  lin<p>  Newton backward-difference polynomial of degree p in the step index (round 3: p = 4)
  log<p>  the same on log(omega)
  dlog    round 4: log(omega) from the quadratic Newton polynomial of its DERIVATIVE, d log(omega)/dk = omega_dot t ln(Q) / omega, at
          the tile's start and the two points before it, integrated from 0 to k (no differencing of values)
Printed: log10 of the largest relative error over the tile: median, 90th percentile, maximum over the tile starts.
    python tests/diagnostics/predictor_study.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle as co
g = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
tgrid = np.logspace(0, 6, 10001)
lnq = np.log(tgrid[1] / tgrid[0])
cfg = co.cfg_synth()
rng = np.random.default_rng(3)
lo, hi = g["prior_lower"], g["prior_upper"]
cases = [np.array(v, float) for v in ([1, 5, -3, 2, -1, 0], [1, 5, -3, 3, -1, 0], [1, 1, -3, 2, 1, 1], [1, 5, -5, 2, -1, 2])]
cases += [lo + (hi - lo) * rng.random(6) for _ in range(40)]


def newton(v, k, order):
    g1 = v[0] - v[1]; g2 = g1 - (v[1] - v[2]); d2b = (v[1] - v[2]) - (v[2] - v[3]); g3 = g2 - d2b
    g4 = g3 - (d2b - ((v[2] - v[3]) - (v[3] - v[4])))
    c2 = 0.5 * k * (k + 1); c3 = c2 * (k + 2) / 3; c4 = c3 * (k + 3) / 4
    return v[0] + k * g1 + (c2 * g2 if order >= 2 else 0) + (c3 * g3 if order >= 3 else 0) + (c4 * g4 if order >= 4 else 0)


res = {}
for p in cases:
    phys = p.copy(); phys[2:] = 10.0 ** phys[2:]
    st, M, W = co.trajectory(cfg, phys, tgrid)
    if st != 0:
        continue
    F = np.array([co.rhs(cfg, phys, tgrid[i], M[i], W[i])[0][1] for i in range(0, 10001)])
    for stride, n in ((8, 256), (4, 256), (2, 256), (8, 128), (4, 128)):
        k = np.arange(1, n + 1, dtype=float)
        for start in range(64 + 4 * stride, 10000 - stride * n, 997):
            true = W[start + stride * np.arange(1, n + 1)]
            h = np.array([W[start - j * stride] for j in range(5)])
            a = np.array([F[start - j * stride] * tgrid[start - j * stride] * lnq * stride / W[start - j * stride] for j in range(3)])
            d1, d2 = a[0] - a[1], a[0] - 2 * a[1] + a[2]
            preds = {"lin4": newton(h, k, 4), "lin2": newton(h, k, 2), "log2": np.exp(newton(np.log(h), k, 2)),
                     "log3": np.exp(newton(np.log(h), k, 3)), "log4": np.exp(np.clip(newton(np.log(h), k, 4), -50, 50)),
                     "dlog": W[start] * np.exp(np.clip(a[0] * k + d1 * k * k / 2 + d2 * (k ** 3 / 6 + k * k / 4), -4, 4))}
            for m, pr in preds.items():
                res.setdefault((stride, n, m), []).append(np.max(np.abs(pr / true - 1.0)))
for key in sorted(res):
    v = np.log10(np.array(res[key]) + 1e-17)
    print(f"stride {key[0]} x {key[1]:3d} steps  {key[2]:5s}  tiles {len(v):4d}  median {np.median(v):5.1f}  90 % {np.percentile(v, 90):5.1f}  max {v.max():5.1f}")

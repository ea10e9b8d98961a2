"""Scan (GPU box): prior-wide and near-truth kernel time at 1 024 / 4 096 walkers for the current environment settings."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from magprop_amd import LogProb
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
lo, hi = gs["prior_lower"], gs["prior_upper"]
data = (gs["Humped_x"], gs["Humped_y"], gs["Humped_yerr"])
out_s = []
for n in (1024, 4096):
    rng = np.random.default_rng(1)
    P = np.array([1.0, 5.0, -3.0, 2.0, -1.0, 0.0]) + 1e-4 * rng.standard_normal((n, 6))
    wide = lo + (hi - lo) * rng.random((n, 6))
    lp = LogProb(*data)
    for nm, X in (("near", P), ("wide", wide)):
        dP = torch.from_numpy(X).cuda(); out = torch.empty(n, dtype=torch.float64, device="cuda")
        for _ in range(10): lp.lnprob_device(dP, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): lp.lnprob_device(dP, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        lp(X); sw = lp.handle.last_sweeps(n)
        out_s.append(f"n={n} {nm} {1e3*dt:.4f} ms tiles {lp.handle.last_mean_tiles:.2f} sw/tile {lp.handle.last_mean_sweeps:.2f} max_sweeps {sw.max()}")
print(" | ".join(out_s))

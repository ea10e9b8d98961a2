#!/usr/bin/env python3
"""Ad-hoc GPU sanity check: HIP path vs the C oracle on the golden parameter clouds + a timing."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magprop_amd import _capi, engine, synth  # noqa: E402
from oracle import c_oracle as co  # noqa: E402

g = np.load(os.path.join(ROOT, "tests/golden/golden_synth.npz"))
tarr = engine.grid(None)
cfg = co.cfg_synth()
for name in ("Humped", "Classic", "Sloped", "Stuttering"):
    x, y, yerr, P = g[name + "_x"], g[name + "_y"], g[name + "_yerr"], g[name + "_pars"]
    ref, rst = co.lnprob_batch(cfg, P, tarr, x, y, yerr, synth.PRIOR_LOWER, synth.PRIOR_UPPER, synth.LOG_MASK)
    out, st = synth._evaluate(P, x, y, yerr, synth.PRIOR_LOWER, synth.PRIOR_UPPER, want_status=True)
    ok = np.isfinite(ref)
    rel = np.abs(out[ok] / ref[ok] - 1.0)
    eng = engine.engine(_capi.cfg_synth())
    print(f"{name:10s} status mismatches {(st != rst).sum()}  inf mismatches {(np.isfinite(out) != ok).sum()}  "
          f"max rel err vs C oracle {rel.max():.3e}  vs reference {np.nanmax(np.abs(out[ok]/g[name+'_lnprob'][ok]-1)):.3e}"
          f"  mean sweeps/tile {eng.handle.last_mean_sweeps:.2f}")

# light curve
st, out, traj = engine.engine(_capi.cfg_synth()).handle.model_lc([1, 5, 1e-3, 100, .1, 1], want_traj=True)
sto, oo, to = co.model_lc(cfg, [1, 5, 1e-3, 100, .1, 1], tarr, want_traj=True)
print("model_lc status", st, "max rel: Ltot", np.abs(out[1] / oo[1] - 1).max(), "M", np.abs(traj[0] / to[0] - 1).max(),
      "omega", np.abs(traj[1] / to[1] - 1).max())

# timing, N_walk = 1024 near truth (synth_mcmc.py:175-176)
rng = np.random.default_rng(1)
for n in (64, 512, 1024, 4096):
    P = np.array([1, 5, -3, 2, -1, 0.0]) + 1e-4 * rng.standard_normal((n, 6))
    x, y, yerr = g["Humped_x"], g["Humped_y"], g["Humped_yerr"]
    synth.lnprob(P, x, y, yerr)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        synth.lnprob(P, x, y, yerr)
    dt = (time.perf_counter() - t0) / reps
    print(f"N_walk={n}: {dt*1e3:.3f} ms/batch (host buffers, incl. PCIe)  {n/dt:.0f} evals/s")
# prior-wide timing
f = np.load(os.path.join(ROOT, "tests/golden/golden_flagscan.npz"))
P = f["pars"][:1024]
synth.lnprob(P, x, y, yerr)
t0 = time.perf_counter()
out, st = synth._evaluate(P, x, y, yerr, synth.PRIOR_LOWER, synth.PRIOR_UPPER, want_status=True)
dt = time.perf_counter() - t0
print(f"prior-wide 1024: {dt*1e3:.3f} ms  status counts {np.bincount(st, minlength=4)}  flag agreement with reference "
      f"{(st[:1024] == f['status'][:1024]).mean():.4f}  mean sweeps {engine.engine(_capi.cfg_synth()).handle.last_mean_sweeps:.2f}")

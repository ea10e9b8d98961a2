#!/usr/bin/env python3
"""Long runs of the fused sampler on every kernel variant (GPU box): no hangs, finite chains, stored values reproduce.
    python tools/stress_sampler.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magprop_amd import EnsembleSampler, LogProb  # noqa: E402

g = np.load(os.path.join(ROOT, "tests/golden/golden_synth.npz"))
x, y, yerr = g["Humped_x"], g["Humped_y"], g["Humped_yerr"]
lo, hi = g["prior_lower"], g["prior_upper"]
rng = np.random.default_rng(5)
lp = LogProb(x, y, yerr)
for nwalk, nsteps, wide in ((24, 20000, False), (512, 12000, True), (1024, 6000, False), (2048, 3000, True), (8192, 1200, False)):
    truth = np.array([1, 5, -3, 2, -1, 0.0])
    pos = (lo + (hi - lo) * rng.random((nwalk, 6))) if wide else truth + 1e-4 * rng.standard_normal((nwalk, 6))
    s = EnsembleSampler(nwalk, 6, x, y, yerr, seed=nwalk)
    t0 = time.perf_counter()
    s.run_mcmc(pos, nsteps, store=False)
    s.run_mcmc(None, 3)                               # three stored steps at the end
    dt = time.perf_counter() - t0
    chain, lnp = s.get_chain(), s.get_log_prob()
    ref = lp(chain[-1])
    fin = np.isfinite(ref)
    ok = np.array_equal(np.isfinite(lnp[-1]), fin) and np.allclose(ref[fin], lnp[-1][fin], rtol=1e-8, atol=1e-9)
    print(f"{nwalk:5d} walkers x {nsteps + 3:6d} steps ({'prior-wide' if wide else 'truth ball'} start): {dt:6.2f} s, "
          f"{nwalk * (nsteps + 3) / dt / 1e6:5.2f} M walker-steps/s, acceptance {s.acceptance_fraction.mean():.3f}, "
          f"finite {fin.mean():.3f}, failed proposals {s.get_bad()[0]}, stored lnprob reproduces: {ok}", flush=True)
    assert ok
    s.close()
print("stress ok")

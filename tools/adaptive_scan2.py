"""(MAGPROP_AMD_* overrides: developer build only -- make -C magprop_amd/csrc experiments and
MAGPROP_AMD_LIB=$PWD/magprop_amd/libmagprop_amd_exp.so.)
Scan (GPU box): launch time of near-truth / burnt-in / prior-wide ensembles of 1 024 walkers under the solver's experiment
knobs (environment of the process: MAGPROP_AMD_COARSE_MAX_SWEEPS, MAGPROP_AMD_FINE_MAX_SWEEPS, MAGPROP_AMD_TROUBLE_LIMIT ...)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from magprop_amd import LogProb, EnsembleSampler
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
x, y, yerr = gs["Humped_x"], gs["Humped_y"], gs["Humped_yerr"]
lo, hi = gs["prior_lower"], gs["prior_upper"]
n = 1024
rng = np.random.default_rng(20261005)
near = np.array([1.0, 5.0, -3.0, 2.0, -1.0, 0.0]) + 1.0e-4 * rng.standard_normal((n, 6))
es = EnsembleSampler(n, 6, x, y, yerr, seed=20261005)
burnt = es.run_mcmc(near, 500, store=False)
es.close()
wide = lo + (hi - lo) * rng.random((n, 6))
lp = LogProb(x, y, yerr)
res = []
for nm, P in (("near", near), ("burnt", burnt), ("wide", wide)):
    dP = torch.from_numpy(P).cuda(); out = torch.empty(n, dtype=torch.float64, device="cuda")
    for _ in range(10): lp.lnprob_device(dP, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): lp.lnprob_device(dP, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    lp(P)
    res.append(f"{nm} {1e3 * dt:.4f} ms ({lp.handle.last_mean_tiles:.1f} x {lp.handle.last_mean_sweeps:.2f}, max sweeps {lp.handle.last_sweeps(n).max()})")
print(" | ".join(res))

import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from magprop_amd import LogProb, EnsembleSampler
gs = np.load("tests/golden/golden_synth.npz")
rng = np.random.default_rng(1)
for grb, truth in (("Humped",[1.0,5.0,-3.0,2.0,-1.0,0.0]),("Classic",[1.0,5.0,-3.0,3.0,-1.0,0.0])):
    lp = LogProb(gs[grb+"_x"], gs[grb+"_y"], gs[grb+"_yerr"])
    lp.handle.tile_log(True)
    P = np.array(truth) + 1e-4 * rng.standard_normal((2048, 6))
    def fmt(log): return " ".join(f"{'P1248'[k]}:{s}:{l}:{w:x}" for k, s, l, w in log)
    lp(P); print(grb, "near-truth SPL=2 walker 0:", fmt(lp.handle.last_tile_log(0)))
    print("   tiles", lp.handle.last_mean_tiles, "sweeps/tile", lp.handle.last_mean_sweeps)
    es = EnsembleSampler(2048, 6, gs[grb+"_x"], gs[grb+"_y"], gs[grb+"_yerr"], seed=7)
    burnt = es.run_mcmc(P, 300, store=False); es.close()
    lp(burnt)
    tl = lp.handle.last_tiles(2048); print("   burnt: tiles mean", tl.mean(), "max", tl.max())
    for i in np.argsort(tl)[::-1][:2]: print("   burnt walker", i, fmt(lp.handle.last_tile_log(i)))

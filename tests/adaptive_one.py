"""Diagnostic (GPU box): one golden point under several solver settings / kernel variants."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from magprop_amd import LogProb
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz")); gf = np.load(os.path.join(ROOT, "tests", "golden", "golden_flagscan.npz"))
i = int(sys.argv[1]) if len(sys.argv) > 1 else 299
p = gf["pars"][i]; tight = gf["lnprob_tight"][i]
data = (gs["Humped_x"], gs["Humped_y"], gs["Humped_yerr"])
for kw in ({}, {"sweep_tol": 1e-9}, {"max_stride": 2}, {"max_stride": 1}, {"sweep_tol": 1e-9, "max_stride": 1}):
    for n in (1, 1500):
        lp = LogProb(*data, **kw)
        P = np.tile(p, (n, 1))
        out = lp(P)
        print(kw, "batch", n, "lnprob", repr(out[0]), "rel dev from tight %.3e" % (abs(out[0] - tight) / abs(tight)), "tiles", lp.handle.last_tiles(1)[0], "sweeps", lp.handle.last_sweeps(1)[0], flush=True)

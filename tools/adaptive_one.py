"""Diagnostic (GPU box): one golden point under several solver settings, with its tile sequence.
    python tools/adaptive_one.py <set> <index> [batch]      set: flagscan | flagscan2_<type>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import TYPES
from magprop_amd import LogProb
G = os.path.join(ROOT, "tests", "golden")
gs, gf, gf2 = (np.load(os.path.join(G, f)) for f in ("golden_synth.npz", "golden_flagscan.npz", "golden_flagscan2.npz"))
name, idx = sys.argv[1], int(sys.argv[2])
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
if name == "flagscan":
    P, tight, ds = gf["pars"], gf["lnprob_tight"], 0
else:
    d = TYPES.index(name.split("_")[1]); m = gf2["ds"] == d - 1
    P, tight, ds = gf2["pars"][m], gf2["lnprob_tight"][m], d
data = [(gs[n + "_x"], gs[n + "_y"], gs[n + "_yerr"]) for n in TYPES]
def fmt(log): return " ".join(f"{'P1248'[k]}:{s}:{l}:{w:x}" for k, s, l, w in log)
for label, kw in (("default", {}), ("sweep_tol 1e-9", {"sweep_tol": 1e-9}), ("max_stride 4", {"max_stride": 4}), ("max_stride 1", {"max_stride": 1})):
    lp = LogProb(*data[0], **kw)
    for s in data[1:]:
        lp.add_dataset(*s)
    lp.handle.tile_log(True)
    a = (idx // batch) * batch
    out = lp.handle.lnprob_batch(P[a:a + batch], ds_id=ds)
    o = out[idx - a]
    print(f"{label}: hip {o!r} tight {tight[idx]!r} ratio {abs(o - tight[idx]) / (1e-7 + 1e-7 * abs(tight[idx])):.3f}  {fmt(lp.handle.last_tile_log(idx - a))}", flush=True)

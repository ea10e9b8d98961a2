"""Diagnostic (GPU box): the golden point with the largest deviation from the reference's tight value, per kernel variant,
next to the serial restatement's adaptive mode on the same point."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import TYPES
from magprop_amd import LogProb
from oracle import c_oracle as co
G = os.path.join(ROOT, "tests", "golden")
gs, gf, gf2 = (np.load(os.path.join(G, f)) for f in ("golden_synth.npz", "golden_flagscan.npz", "golden_flagscan2.npz"))
tarr = np.logspace(0.0, 6.0, 10001); lo, hi = gs["prior_lower"], gs["prior_upper"]
data = [(gs[n + "_x"], gs[n + "_y"], gs[n + "_yerr"]) for n in TYPES]
lp = LogProb(*data[0])
for s in data[1:]:
    lp.add_dataset(*s)
sets = [("flagscan", gf["pars"], gf["lnprob_tight"], gf["status"], 0)]
for d, n in enumerate(TYPES[1:]):
    m = gf2["ds"] == d
    sets.append(("flagscan2_" + n, gf2["pars"][m], gf2["lnprob_tight"][m], gf2["status"][m], d + 1))
for name, P, tight, rst, ds in sets:
    for batch in (1500, 750):
        out = np.empty(len(P)); tl = np.empty(len(P), int); sw = np.empty(len(P), int)
        for a in range(0, len(P), batch):
            out[a:a + batch] = lp.handle.lnprob_batch(P[a:a + batch], ds_id=ds)
            n_ = len(P[a:a + batch]); tl[a:a + batch] = lp.handle.last_tiles(n_); sw[a:a + batch] = lp.handle.last_sweeps(n_)
        ok = (rst == 0) & np.isfinite(tight) & np.isfinite(out)
        r = np.where(ok, np.abs(out - tight) / (1e-7 + 1e-7 * np.abs(tight)), 0.0)
        i = int(np.argmax(r))
        o4, _ = co.lnprob_batch(co.cfg_synth(), P[i], tarr, *data[ds], lo, hi, 0b111100, mode="adaptive", spl=4)
        o2, _ = co.lnprob_batch(co.cfg_synth(), P[i], tarr, *data[ds], lo, hi, 0b111100, mode="adaptive", spl=2)
        of, _ = co.lnprob_batch(co.cfg_synth(), P[i], tarr, *data[ds], lo, hi, 0b111100)
        print(f"{name} batch {batch}: worst {r[i]:.3f} at {i}: hip {out[i]!r} tight {tight[i]!r} oracle adaptive spl4 {o4[0]!r} spl2 {o2[0]!r} fixed {of[0]!r} "
              f"tiles {tl[i]} sweeps {sw[i]} pars {np.round(P[i], 5).tolist()}", flush=True)

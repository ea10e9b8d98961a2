"""Diagnostic (GPU box): per-walker tiles / sweeps of the adaptive solver on prior-wide walkers; status mismatches."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import TYPES
from magprop_amd import LogProb, _capi
G = os.path.join(ROOT, "tests", "golden")
gs, gf, gf2 = (np.load(os.path.join(G, f)) for f in ("golden_synth.npz", "golden_flagscan.npz", "golden_flagscan2.npz"))
lo, hi = gs["prior_lower"], gs["prior_upper"]
data = [(gs[n + "_x"], gs[n + "_y"], gs[n + "_yerr"]) for n in TYPES]
rng = np.random.default_rng(1)
rng.standard_normal((1024, 6))
wide = lo + (hi - lo) * rng.random((1024, 6))
for label, kw in (("adaptive", {}), ("fixed", {"max_stride": 1})):
    lp = LogProb(*data[0], **kw)
    out, st = lp.handle.lnprob_batch(wide, want_status=True)
    sw, tl = lp.handle.last_sweeps(1024), lp.handle.last_tiles(1024)
    order = np.argsort(sw)[::-1][:8]
    print(label, "sweeps: mean %.1f max %d; tiles mean %.1f max %d" % (sw.mean(), sw.max(), tl.mean(), tl.max()))
    for i in order:
        print("   walker", i, "sweeps", sw[i], "tiles", tl[i], "status", st[i], "pars", np.round(wide[i], 4).tolist())
# status mismatches of the adaptive mode on the golden scans
lp = LogProb(*data[0])
for s in data[1:]:
    lp.add_dataset(*s)
for name, P, rst, ds in [("flagscan", gf["pars"], gf["status"], np.zeros(1500, np.int32)), ("flagscan2", gf2["pars"], gf2["status"], (gf2["ds"] + 1).astype(np.int32))]:
    out, st = lp.handle.lnprob_batch(P, ds_id=ds, want_status=True)
    bad = np.nonzero(st != rst)[0]
    sw, tl = lp.handle.last_sweeps(len(P)), lp.handle.last_tiles(len(P))
    print(name, "mismatches", bad.tolist(), [(int(st[i]), int(rst[i]), int(sw[i]), int(tl[i])) for i in bad], [np.round(P[i], 5).tolist() for i in bad])

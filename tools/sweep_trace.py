#!/usr/bin/env python3
"""Developer diagnostic (GPU box): how the converged region of a slowly converging tile grows, sweep by sweep.
    make -C magprop_amd/csrc sweep-trace && MAGPROP_AMD_LIB=$PWD/magprop_amd/libmagprop_amd_trace.so python tools/sweep_trace.py
Prints, for the slowest walkers of a prior-wide batch, the first tile over single intervals that took more than 12 sweeps:
per sweep "first pending lane / pending lanes / lanes beyond the break-up limit"."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magprop_amd import LogProb, _capi
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
lo, hi = gs["prior_lower"], gs["prior_upper"]
rng = np.random.default_rng(20261004)
_ = rng.standard_normal((1024, 6)), rng.standard_normal((1024, 6))          # the draws tools/ab_tail.py makes before its prior-wide sets
wide = lo + (hi - lo) * rng.random((1024, 6))
extra = np.array([[8.021, 7.738, -4.792, 3.258, 1.92, 1.486], [7.113, 1.018, -3.261, 2.594, 1.82, 0.298], [9.706, 6.854, -4.537, 3.208, 1.653, 1.239]])
X = np.concatenate([extra, wide])
lp = LogProb(gs["Humped_x"], gs["Humped_y"], gs["Humped_yerr"])
lp.handle.tile_log(True)
out, st = lp.handle.lnprob_batch(X, want_status=True)
sw = lp.handle.last_sweeps(len(X))
buf = np.zeros(96, dtype=np.int32)
for i in list(range(3)) + list(np.argsort(sw)[::-1][:4]):
    m = lp.handle._L.mp_last_tile_log(lp.handle._h, int(i), _capi._iptr(buf), 96)
    words = [int(w) for w in buf[:max(m, 0)]]
    print(f"walker {i} status {st[i]} total sweeps {sw[i]} pars {np.round(X[i], 3).tolist()}")
    print("   tile", (words[-1] >> 24) & 0xFF if words else None, " ".join(f"{w & 0xFF}/{(w >> 8) & 0xFF}/{(w >> 16) & 0xFF}" for w in words))

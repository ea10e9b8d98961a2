"""Diagnostic (GPU box): ONE walker (sampler coordinates on the command line, dataset index) under several solver settings,
in a batch of 4 096 copies (2 steps per lane) and of 64 (4 steps per lane).
    python tools/adaptive_walker.py <ds> p0 p1 p2 p3 p4 p5"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import TYPES
from magprop_amd import LogProb, _capi
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
ds = int(sys.argv[1]); p = np.array([float(v) for v in sys.argv[2:8]])
sets = [(gs[n + "_x"], gs[n + "_y"], gs[n + "_yerr"]) for n in TYPES]
def fmt(log): return " ".join(f"{'P1248'[k]}:{s}:{l}:{w:x}" for k, s, l, w in log)
ref = None
for label, kw in (("fixed strict", {"sweep_tol": _capi.SWEEP_TOL_STRICT, "max_stride": 1}), ("default", {}), ("sweep_tol 1e-9", {"sweep_tol": 1e-9}),
                  ("max_stride 4", {"max_stride": 4}), ("max_stride 2", {"max_stride": 2}), ("stride_tol 1e-8", {"stride_tol": 1e-8})):
    for n in (4096, 64):
        lp = LogProb(*sets[ds], **kw)
        lp.handle.tile_log(True)
        o = lp.handle.lnprob_batch(np.tile(p, (n, 1)))[0]
        if ref is None: ref = o
        print(f"{label:16s} n={n:5d}: {o!r} rel {abs(o - ref) / max(abs(ref), 1.0):.3e}  {fmt(lp.handle.last_tile_log(0))}", flush=True)

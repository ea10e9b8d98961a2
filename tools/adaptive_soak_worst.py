"""Diagnostic (GPU box): the soak's walkers (tests/test_gpu_soak.py, any N) at the product's defaults against the fixed-step
strict kernel; the worst walkers with their tile sequences.   python tools/adaptive_soak_worst.py [N]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import TYPES
import test_gpu_soak as tgs
from magprop_amd import LogProb, _capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
rng = np.random.default_rng(20261003)
lo, hi = gs["prior_lower"], gs["prior_upper"]
P = tgs._walkers(rng, N, lo, hi)
ids = rng.integers(0, 4, N).astype(np.int32)
sets = [(gs[n + "_x"], gs[n + "_y"], gs[n + "_yerr"]) for n in TYPES]
def fmt(log): return " ".join(f"{'P1248'[k]}:{s}:{l}:{w:x}" for k, s, l, w in log)
def evaluate(batch, **kw):
    lp = LogProb(*sets[0], **kw)
    for s in sets[1:]: lp.add_dataset(*s)
    out = np.empty(N); st = np.empty(N, dtype=np.int32)
    for a in range(0, N, batch):
        out[a:a + batch], st[a:a + batch] = lp.handle.lnprob_batch(P[a:a + batch], ds_id=ids[a:a + batch], want_status=True)
    return lp, out, st
_, ref, rst = evaluate(4096, sweep_tol=_capi.SWEEP_TOL_STRICT, max_stride=1)
for batch in (4096, 1024):
    lp, out, st = evaluate(batch)
    both = (rst == 0) & (st == 0)
    rel = np.where(both, np.abs(out - ref) / np.maximum(np.abs(ref), 1.0), 0.0)
    print(f"batch {batch}: status mismatches {int((st != rst).sum())}, max {rel.max():.3e}, above 1e-7: {int((rel > 1e-7).sum())}, above 5e-8: {int((rel > 5e-8).sum())}")
    lp.handle.tile_log(True)
    for i in np.argsort(rel)[::-1][:4]:
        a = (i // batch) * batch
        o = lp.handle.lnprob_batch(P[a:a + batch], ds_id=ids[a:a + batch])
        print(f"   walker {i} ds {ids[i]} rel {rel[i]:.3e} pars {np.round(P[i], 4).tolist()} hip {out[i]!r} fixed {ref[i]!r}: {fmt(lp.handle.last_tile_log(i - a))}")
    lp.handle.tile_log(False)

#!/usr/bin/env python3
"""Same-box A/B of the slow-walker tail (GPU box):  MAGPROP_AMD_LIB=$PWD/ab/libX.so python tools/ab_tail.py [tag]
Kernel time (HIP events over back-to-back launches) of 1 024 walkers near the truth, uniform over the prior box and burnt in
(500 sampler steps), and of 8 192 prior-wide walkers, with the distribution of tiles and sweeps per walker."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from magprop_amd import EnsembleSampler, LogProb
tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("MAGPROP_AMD_LIB", "default"))
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
lo, hi = gs["prior_lower"], gs["prior_upper"]
data = (gs["Humped_x"], gs["Humped_y"], gs["Humped_yerr"])
truth = np.array([1.0, 5.0, -3.0, 2.0, -1.0, 0.0])
rng = np.random.default_rng(20261004)
lp = LogProb(*data)


def timed(X, reps=60):
    dP = torch.from_numpy(np.ascontiguousarray(X)).cuda()
    out = torch.empty(len(X), dtype=torch.float64, device="cuda")
    for _ in range(8):
        lp.lnprob_device(dP, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        lp.lnprob_device(dP, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    _, st = lp.handle.lnprob_batch(X, want_status=True)
    ok = st == 0
    tl, sw = lp.handle.last_tiles(len(X))[ok], lp.handle.last_sweeps(len(X))[ok]
    if os.environ.get("AB_TAIL_LOG"):
        lp.handle.tile_log(True)
        lp.handle.lnprob_batch(X)
        allsw = lp.handle.last_sweeps(len(X))
        for i in np.argsort(allsw)[::-1][:2]:
            print("   walker", i, "sweeps", allsw[i], np.round(X[i], 3).tolist(), " ".join(f"{'P1248'[k]}:{s_}:{l}:{w:x}" for k, s_, l, w in lp.handle.last_tile_log(i)))
        lp.handle.tile_log(False)
    return ms, tl, sw


es = EnsembleSampler(1024, 6, *data, seed=7)
burnt = es.run_mcmc(truth + 1e-4 * rng.standard_normal((1024, 6)), 500, store=False)
es.close()
bench_wide = lo + (hi - lo) * np.random.default_rng(20261003 + 1).random((1024, 6))      # bench.py's kernel_ms.prior_wide set
sets = [("near 1024", truth + 1e-4 * rng.standard_normal((1024, 6))), ("wide bench", bench_wide), ("wide 1024", lo + (hi - lo) * rng.random((1024, 6))),
        ("wide 1024 b", lo + (hi - lo) * rng.random((1024, 6))), ("burnt 1024", burnt), ("wide 8192", lo + (hi - lo) * rng.random((8192, 6)))]
for name, X in sets:
    ms, tl, sw = timed(X)
    print(f"{tag:28s} {name:12s} {ms:.4f} ms  {len(X) / ms / 1e3:6.2f} M/s | tiles mean {tl.mean():5.2f} p99 {np.percentile(tl, 99):4.0f} max {tl.max():3d} | "
          f"sweeps mean {sw.mean():5.1f} p99 {np.percentile(sw, 99):4.0f} max {sw.max():3d}", flush=True)

"""Diagnostic (GPU box): a burnt-in ensemble (500 sampler steps from the 1e-4 ball) through the solver at max_stride 8 / 4:
launch time, tiles, sweeps, and the tile sequences of its slowest walkers."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from magprop_amd import LogProb, EnsembleSampler
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
x, y, yerr = gs["Humped_x"], gs["Humped_y"], gs["Humped_yerr"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(20261005)
es = EnsembleSampler(n, 6, x, y, yerr, seed=20261005)
burnt = es.run_mcmc(np.array([1.0, 5.0, -3.0, 2.0, -1.0, 0.0]) + 1.0e-4 * rng.standard_normal((n, 6)), 500, store=False)
es.close()
def fmt(log): return " ".join(f"{'P1248'[k]}:{s}:{l}:{w:x}" for k, s, l, w in log)
for ms in (8, 4):
    lp = LogProb(x, y, yerr, max_stride=ms)
    lp.handle.tile_log(True)
    dP = torch.from_numpy(burnt).cuda(); out = torch.empty(n, dtype=torch.float64, device="cuda")
    for _ in range(10): lp.lnprob_device(dP, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): lp.lnprob_device(dP, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    lp(burnt)
    sw, tl = lp.handle.last_sweeps(n), lp.handle.last_tiles(n)
    print(f"max_stride {ms}: {1e3 * dt:.4f} ms; tiles mean {tl.mean():.2f} max {tl.max()}; sweeps mean {sw.mean():.1f} p90 {np.percentile(sw, 90):.0f} max {sw.max()}")
    for i in np.argsort(sw)[::-1][:4]:
        print(f"   walker {i} sweeps {sw[i]} pars {np.round(burnt[i], 3).tolist()}: {fmt(lp.handle.last_tile_log(i))}")

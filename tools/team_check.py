"""GPU box: the team kernels (2 / 4 wavefronts per walker, batches of <= n_simd/2, <= n_simd/4 walkers) against the
4-steps-per-lane kernel on the same walkers -- values, statuses, tiles and sweeps per walker -- and their launch times.

    python tools/team_check.py [reps]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from magprop_amd import LogProb, _capi  # noqa: E402

gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
lo, hi = gs["prior_lower"], gs["prior_upper"]
data = (gs["Humped_x"], gs["Humped_y"], gs["Humped_yerr"])
truth = np.array([1.0, 5.0, -3.0, 2.0, -1.0, 0.0])
rng = np.random.default_rng(1)
N = 1024
near = truth + 1e-4 * rng.standard_normal((N, 6))
wide = lo + (hi - lo) * rng.random((N, 6))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def run(lp, X, batch):
    out = np.empty(len(X))
    st = np.empty(len(X), dtype=np.int32)
    tiles = np.empty(len(X), dtype=np.int32)
    sweeps = np.empty(len(X), dtype=np.int32)
    for a in range(0, len(X), batch):
        o, s = lp.handle.lnprob_batch(X[a:a + batch], want_status=True)
        out[a:a + batch], st[a:a + batch] = o, s
        tiles[a:a + batch] = lp.handle.last_tiles(len(o))
        sweeps[a:a + batch] = lp.handle.last_sweeps(len(o))
    return out, st, tiles, sweeps


for label, kw in (("product defaults", {}), ("strict, max_stride 1", {"sweep_tol": _capi.SWEEP_TOL_STRICT, "max_stride": 1})):
    lp = LogProb(*data, **kw)
    for nm, X in (("near", near), ("wide", wide)):
        base = run(lp, X, 1024)
        for batch in (512, 256):
            o, s, t, sw = run(lp, X, batch)
            ok = (base[1] == 0) & (s == 0)
            rel = np.abs(o[ok] - base[0][ok]) / np.maximum(np.abs(base[0][ok]), 1.0)
            print(f"{label:22s} {nm:5s} batch {batch:4d} vs 1024: status mismatches {int(np.sum(s != base[1]))}, "
                  f"max rel diff {rel.max():.2e}, p99 {np.quantile(rel, 0.99):.2e}, walkers with other tile count "
                  f"{int(np.sum(t != base[2]))}, other sweep count {int(np.sum(sw != base[3]))}; tiles {t[ok].mean():.2f} "
                  f"(1024: {base[2][ok].mean():.2f}), sweeps {sw[ok].mean():.2f} ({base[3][ok].mean():.2f})", flush=True)

lp = LogProb(*data)
for nm, X in (("near", near), ("wide", wide)):
    for n in (128, 256, 512, 1024):
        dP = torch.from_numpy(X[:n].copy()).cuda()
        out = torch.empty(n, dtype=torch.float64, device="cuda")
        for _ in range(10):
            lp.lnprob_device(dP, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            lp.lnprob_device(dP, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"time {nm:5s} n={n:5d}: {1e3 * dt:.4f} ms = {n / dt / 1e6:.2f} M evals/s", flush=True)

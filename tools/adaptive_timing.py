"""(MAGPROP_AMD_* overrides: developer build only -- make -C magprop_amd/csrc experiments and
MAGPROP_AMD_LIB=$PWD/magprop_amd/libmagprop_amd_exp.so.)
Timing (GPU box): kernel time at several batch sizes / steps per lane / max_stride, with tiles and sweeps."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import TYPES
from magprop_amd import LogProb
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
lo, hi = gs["prior_lower"], gs["prior_upper"]
grb = sys.argv[1] if len(sys.argv) > 1 else "Humped"
data = (gs[grb + "_x"], gs[grb + "_y"], gs[grb + "_yerr"])
truth = {"Humped": [1.0, 5.0, -3.0, 2.0, -1.0, 0.0], "Classic": [1.0, 5.0, -3.0, 3.0, -1.0, 0.0]}[grb]
rng = np.random.default_rng(1)
for n in (512, 1024, 2048, 4096, 8192):
    P = np.array(truth) + 1e-4 * rng.standard_normal((n, 6))
    wide = lo + (hi - lo) * rng.random((n, 6))
    for ms in (8, 4):
        lp = LogProb(*data, max_stride=ms)
        for nm, X in (("near", P), ("wide", wide)):
            dP = torch.from_numpy(X).cuda()
            out = torch.empty(n, dtype=torch.float64, device="cuda")
            for _ in range(10):
                lp.lnprob_device(dP, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 100
            for _ in range(reps):
                lp.lnprob_device(dP, out=out)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            lp(X)
            print(f"SPL={os.environ.get('MAGPROP_AMD_SPL', 'auto')} n={n} max_stride={ms} {nm}: {1e3 * dt:.4f} ms = {n / dt / 1e6:.2f} M evals/s; "
                  f"tiles/walker {lp.handle.last_mean_tiles:.2f}, sweeps/tile {lp.handle.last_mean_sweeps:.2f}", flush=True)

#!/usr/bin/env python3
"""Same-box A/B of small launches (GPU box):  MAGPROP_AMD_LIB=$PWD/ab/libX.so python tools/ab_small.py [tag]
Kernel time (HIP events over back-to-back launches) of 128 ... 1 024 walkers near the truth and uniform over the prior box:
the launches the team kernels serve (four wavefronts per walker up to n_simd / 2 walkers)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from magprop_amd import LogProb
tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("MAGPROP_AMD_LIB", "default"))
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
lo, hi = gs["prior_lower"], gs["prior_upper"]
lp = LogProb(gs["Humped_x"], gs["Humped_y"], gs["Humped_yerr"])
truth = np.array([1.0, 5.0, -3.0, 2.0, -1.0, 0.0])
rng = np.random.default_rng(20261005)
near = truth + 1e-4 * rng.standard_normal((1024, 6))
wide = lo + (hi - lo) * rng.random((1024, 6))
row = []
for name, X in (("near", near), ("wide", wide)):
    for n in (128, 256, 512, 1024):
        dP = torch.from_numpy(np.ascontiguousarray(X[:n])).cuda()
        out = torch.empty(n, dtype=torch.float64, device="cuda")
        for _ in range(10):
            lp.lnprob_device(dP, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(200):
            lp.lnprob_device(dP, out=out)
        e1.record()
        torch.cuda.synchronize()
        row.append(f"{name}{n} {e0.elapsed_time(e1) / 200:.4f}")
print(f"{tag:20s} " + " | ".join(row), flush=True)

#!/usr/bin/env python3
"""GPU box: ms per step of the device-resident sampler for small ensembles (whole steps and half-steps):
    MAGPROP_AMD_LIB=$PWD/ab/libX.so python tools/ab_sampler_small.py [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magprop_amd import EnsembleSampler  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
truth = np.array([1.0, 5.0, -3.0, 2.0, -1.0, 0.0])
tag = os.path.basename(os.environ.get("MAGPROP_AMD_LIB", "default"))
for n in [int(v) for v in os.environ.get("NS", "24 64 128 170 256 340 512").split()]:
    p0 = truth + 1.0e-4 * np.random.default_rng(5).standard_normal((n, 6))
    row = []
    for w in (True, False):
        es = EnsembleSampler(n, 6, g["Humped_x"], g["Humped_y"], g["Humped_yerr"], seed=11, whole_step=w)
        es.run_mcmc(p0, 5, store=False)
        t = time.perf_counter()
        es.run_mcmc(None, steps, store=False)
        t = time.perf_counter() - t
        row.append(f"{'whole' if w else 'half '} {1e3 * t / steps:.4f} ms/step {n * steps / t / 1e6:6.3f} M w-steps/s")
        es.close()
    print(f"{tag:14s} {n:4d} walkers: " + " | ".join(row), flush=True)

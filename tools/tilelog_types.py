#!/usr/bin/env python3
"""Developer diagnostic (GPU box): tile sequences (kind:sweeps:lanes kept:why) of a walker near each of the four truths, for the
4-steps-per-lane (1 024 walkers) and the 2-steps-per-lane (2 048 walkers) kernels, with tiles and sweeps per walker.
    python tools/tilelog_types.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magprop_amd import LogProb
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
TRUTH = {"Humped": [1.0, 5.0, -3.0, 2.0, -1.0, 0.0], "Classic": [1.0, 5.0, -3.0, 3.0, -1.0, 0.0],
         "Sloped": [1.0, 1.0, -3.0, 2.0, 1.0, 1.0], "Stuttering": [1.0, 5.0, -5.0, 2.0, -1.0, 2.0]}
rng = np.random.default_rng(1)
fmt = lambda log: " ".join(f"{'P1248'[k]}:{s}:{l}:{w:x}" for k, s, l, w in log)
for grb, truth in TRUTH.items():
    lp = LogProb(gs[grb + "_x"], gs[grb + "_y"], gs[grb + "_yerr"])
    lp.handle.tile_log(True)
    for n in (1024, 2048):
        P = np.array(truth) + 1e-4 * rng.standard_normal((n, 6))
        lp(P)
        print(f"{grb:10s} n={n}: tiles {lp.handle.last_mean_tiles:.2f} sweeps/tile {lp.handle.last_mean_sweeps:.2f} | {fmt(lp.handle.last_tile_log(0))}")
    lp.handle.close()

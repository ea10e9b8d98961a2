#!/usr/bin/env python3
"""End-to-end check of the fused sampler on the four seeded synthetic datasets (what synth_mcmc.py + plot_synth.py
do in the reference): 512 walkers x 3000 steps from the 1e-4 ball around the truths, posterior medians and 16-84 %
intervals after burn-in against the generating parameters (synth_mcmc.py:16-21).  python tools/posterior_check.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magprop_amd import EnsembleSampler  # noqa: E402

TRUTHS = {"Humped": [1.0, 5.0, -3.0, 2.0, -1.0, 0.0], "Classic": [1.0, 5.0, -3.0, 3.0, -1.0, 0.0],
          "Sloped": [1.0, 1.0, -3.0, 2.0, 1.0, 1.0], "Stuttering": [1.0, 5.0, -5.0, 2.0, -1.0, 2.0]}
NAMES = ["B", "P", "log MdiscI", "log RdiscI", "log eps", "log delta"]


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
    # (python tools/posterior_check.py [nwalk nstep burn]: 128 walkers and fewer run the sampler's team kernels)
    nwalk, nstep, burn = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 3000, 1000)
    for grb, truth in TRUTHS.items():
        x, y, yerr = g[grb + "_x"], g[grb + "_y"], g[grb + "_yerr"]
        rng = np.random.default_rng(7)
        pos = np.array(truth) + 1.0e-4 * rng.standard_normal((nwalk, 6))
        s = EnsembleSampler(nwalk, 6, x, y, yerr, seed=11)
        t0 = time.perf_counter()
        s.run_mcmc(pos, nstep)
        dt = time.perf_counter() - t0
        chain = s.get_chain()[burn:].reshape(-1, 6)
        lo, med, hi = np.percentile(chain, [16, 50, 84], axis=0)
        tau = s.get_autocorr_time(quiet=True)
        inside = np.sum((np.array(truth) >= np.percentile(chain, 2.5, axis=0)) & (np.array(truth) <= np.percentile(chain, 97.5, axis=0)))
        print(f"{grb}: {nwalk * nstep / dt / 1e6:.2f} M walker-steps/s, acceptance {s.acceptance_fraction.mean():.3f}, "
              f"mean tau {np.mean(tau):.0f}, truths inside the central 95 %: {inside}/6")
        for k in range(6):
            print(f"   {NAMES[k]:11s} truth {truth[k]:7.3f}   median {med[k]:7.3f}  (+{hi[k] - med[k]:.3f} / -{med[k] - lo[k]:.3f})")
        s.close()


if __name__ == "__main__":
    main()

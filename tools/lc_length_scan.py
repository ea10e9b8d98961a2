"""Kernel time of one 1024-walker lnprob batch against the length of the observed light curve (GPU box).

Real GRB light curves have 8..1944 points (data/real_data/); every synthetic set has 50.  Usage:
    python tools/lc_length_scan.py [--nwalk 1024] [--reps 50]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nwalk", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--lengths", type=int, nargs="*", default=[8, 50, 64, 65, 128, 256, 512, 1024, 1944])
    a = ap.parse_args()
    import torch

    import magprop_amd as mpa
    from magprop_amd import funcs, synth

    rng = np.random.default_rng(11)
    truth = np.array([1.0, 5.0, -3.0, np.log10(100.0), 0.1, 1.0])
    truth_phys = truth.copy()
    truth_phys[2:] = 10.0 ** truth_phys[2:]
    lo, hi = np.asarray(synth.PRIOR_LOWER), np.asarray(synth.PRIOR_UPPER)
    pars = truth + 0.02 * (hi - lo) * rng.standard_normal((a.nwalk, 6))
    pars = np.clip(pars, lo + 1e-6, hi - 1e-6)
    d_pars = torch.tensor(pars, dtype=torch.float64, device="cuda")
    for n_obs in a.lengths:
        x = np.sort(10.0 ** rng.uniform(0.3, 5.7, size=n_obs))
        y = funcs.model_lum(truth_phys, xdata=x)
        yerr = 0.1 * np.abs(y) + 1e-6
        y = y + yerr * rng.standard_normal(n_obs)
        lp = mpa.LogProb(x, y, yerr)
        out = torch.empty(a.nwalk, dtype=torch.float64, device="cuda")
        for _ in range(5):
            lp.lnprob_device(d_pars, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            lp.lnprob_device(d_pars, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        host = lp(pars[:8])
        print(f"n_obs={n_obs:5d}  ms/batch={ms:.4f}  evals/s={a.nwalk / ms * 1e3:,.0f}  lnp[0]={host[0]:.6f}", flush=True)


if __name__ == "__main__":
    main()

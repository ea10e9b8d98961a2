#!/usr/bin/env python3
"""Config 1 of BASELINE.json end to end on the GPU: what `synth_mcmc.py --grb Humped -w 24 -s 50` does
(code/synthetic_datasets/synth_mcmc.py:132-226), with the reference's output files.

    python tools/run_synth_mcmc.py --grb Humped -w 24 -s 50 --out /tmp/synth_out [--data path/to/Humped.csv]
    python tools/run_synth_mcmc.py --grb Humped --re-run --out /tmp/synth_out      # synth_mcmc.py:139-149

--re-run takes Npars/Nwalk/Nstep/seed from <out>/<grb>_info.json and reproduces the chain bit for bit (the
initial ball, the random splits and the move all derive from the stored seed).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magprop_amd import EnsembleSampler, mcmc_io  # noqa: E402

TRUTHS = {"Humped": [1.0, 5.0, -3.0, 2.0, -1.0, 0.0], "Classic": [1.0, 5.0, -3.0, 3.0, -1.0, 0.0],
          "Sloped": [1.0, 1.0, -3.0, 2.0, 1.0, 1.0], "Stuttering": [1.0, 5.0, -5.0, 2.0, -1.0, 2.0]}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--grb", required=True, choices=list(TRUTHS))
    ap.add_argument("-s", "--n-step", type=int, default=50)
    ap.add_argument("-w", "--n-walk", type=int, default=24)
    ap.add_argument("--seed", type=int, default=20261003)
    ap.add_argument("--data", default=None, help="x,y,yerr CSV (default: the seeded golden dataset)")
    ap.add_argument("--out", default="synth_out")
    ap.add_argument("-r", "--re-run", action="store_true", help="repeat the run recorded in <out>/<grb>_info.json")
    a = ap.parse_args(argv)
    if a.re_run:
        rec = mcmc_io.read_info(os.path.join(a.out, a.grb + "_info.json"))
        if rec["Npars"] != 6:
            sys.exit("run_synth_mcmc: the synthetic fits have 6 parameters")
        a.n_walk, a.n_step, a.seed = int(rec["Nwalk"]), int(rec["Nstep"]), int(rec["seed"])

    if a.data:
        x, y, yerr = mcmc_io.read_dataset(a.data)
    else:
        g = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
        x, y, yerr = g[a.grb + "_x"], g[a.grb + "_y"], g[a.grb + "_yerr"]
    os.makedirs(a.out, exist_ok=True)
    base = os.path.join(a.out, a.grb)
    rng = np.random.default_rng(a.seed)
    pos = np.array(TRUTHS[a.grb]) + 1.0e-4 * rng.standard_normal((a.n_walk, 6))      # synth_mcmc.py:175-176
    s = EnsembleSampler(a.n_walk, 6, x, y, yerr, seed=a.seed)
    t0 = time.perf_counter()
    s.run_mcmc(pos, a.n_step)
    dt = time.perf_counter() - t0
    mcmc_io.write_chain_files(base, s.get_chain(), s.get_log_prob())
    tau = s.get_autocorr_time(quiet=True)
    info = mcmc_io.write_info(base + "_info.json", 6, a.n_walk, a.n_step, a.seed, s.acceptance_fraction, tau)
    print(f"{a.grb}\nMean acceptance fraction: {info['acceptance_fraction']}\nAverage auto-correlation time: {np.mean(tau):.3f}")
    print(f"{a.n_walk * (a.n_step + 1)} lnprob evaluations in {dt:.3f} s ({a.n_walk * a.n_step / dt:.0f} walker-steps/s); files under {a.out}/")


if __name__ == "__main__":
    main()

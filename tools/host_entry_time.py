"""Wall time per call of the host-buffer entry (numpy in, numpy out: PCIe both ways + synchronisation), the path
`emcee.EnsembleSampler(..., vectorize=True)` drives.  python tools/host_entry_time.py [--reps 200]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    import magprop_amd as mpa
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "golden_synth.npz"))
    lp = mpa.LogProb(g["Humped_x"], g["Humped_y"], g["Humped_yerr"])
    rng = np.random.default_rng(1)
    for n in (24, 512, 1024, 4096):
        P = np.array([1.0, 5.0, -3.0, 2.0, -1.0, 0.0]) + 1.0e-4 * rng.standard_normal((n, 6))
        for _ in range(10):
            lp(P)
        t0 = time.perf_counter()
        for _ in range(a.reps):
            lp(P)
        dt = (time.perf_counter() - t0) / a.reps
        print(f"n={n:5d}  {dt * 1e3:.4f} ms/call  {n / dt:,.0f} evals/s", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""(MAGPROP_AMD_* overrides: developer build only -- make -C magprop_amd/csrc experiments and
MAGPROP_AMD_LIB=$PWD/magprop_amd/libmagprop_amd_exp.so.)
Experiment: accuracy and speed against the Newton-sweep tolerance (MAGPROP_AMD_SWEEP_TOL), per kernel variant.
Accuracy is measured against the kernel's own result at tolerance 1e-13 on the golden parameter clouds and the
1 500-point prior-wide scan; speed on a near-truth batch.  python tools/tol_scan.py   (GPU box)"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TYPES = ("Humped", "Classic", "Sloped", "Stuttering")


def child(batch):
    import time
    from magprop_amd import _capi, engine, synth
    g = np.load(os.path.join(ROOT, "tests/golden/golden_synth.npz"))
    f = np.load(os.path.join(ROOT, "tests/golden/golden_flagscan.npz"))
    h = _capi.Handle(_capi.cfg_synth(), engine.grid(None))
    h.set_prior(synth.PRIOR_LOWER, synth.PRIOR_UPPER, synth.LOG_MASK)
    for k, name in enumerate(TYPES):
        h.set_dataset(k, g[name + "_x"], g[name + "_y"], g[name + "_yerr"])
    P = np.concatenate([g[name + "_pars"] for name in TYPES] + [f["pars"]])
    ids = np.concatenate([np.full(len(g[name + "_pars"]), k) for k, name in enumerate(TYPES)] + [np.zeros(len(f["pars"]))]).astype(np.int32)
    reps = -(-batch // len(P))
    Pb, ib = np.tile(P, (reps, 1))[:batch], np.tile(ids, reps)[:batch]       # same walkers, batch-size selects the variant
    out, st = h.lnprob_batch(Pb, ds_id=ib, want_status=True)
    sweeps_wide = h.last_mean_sweeps
    rng = np.random.default_rng(1)
    T = np.array([1, 5, -3, 2, -1, 0.0]) + 1e-4 * rng.standard_normal((batch, 6))
    h.lnprob_batch(T, ds_id=0)
    t0 = time.perf_counter()
    for _ in range(20):
        h.lnprob_batch(T, ds_id=0)
    dt = (time.perf_counter() - t0) / 20
    n = min(batch, len(P))
    print(json.dumps({"out": out[:n].tolist(), "st": st[:n].tolist(), "sweeps_wide": sweeps_wide,
                      "sweeps_truth": h.last_mean_sweeps, "ms": dt * 1e3}))


if len(sys.argv) > 2 and sys.argv[1] == "child":
    child(int(sys.argv[2]))
else:
    for batch in (1024, 4096, 256):
        ref = None
        for tol in ("1e-13", "1e-9", "1e-8", "1e-7", "3e-7", "1e-6"):
            env = dict(os.environ, MAGPROP_AMD_SWEEP_TOL=tol)
            r = subprocess.run([sys.executable, __file__, "child", str(batch)], env=env, capture_output=True, text=True)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            out, st = np.array(d["out"]), np.array(d["st"])
            if ref is None:
                ref = (out, st)
            ok = (st == 0) & (ref[1] == 0)
            rel = np.abs(out[ok] - ref[0][ok]) / (np.abs(ref[0][ok]) + 10.0)
            print(f"batch {batch:5d} tol {tol:6s}  ms {d['ms']:.4f}  sweeps/tile truth {d['sweeps_truth']:.3f} wide {d['sweeps_wide']:.3f}  "
                  f"max rel {rel.max():.2e}  p99 {np.quantile(rel, 0.99):.2e}  status mismatches {int((st != ref[1]).sum())}", flush=True)

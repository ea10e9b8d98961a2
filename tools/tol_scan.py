#!/usr/bin/env python3
"""Experiment: accuracy vs the C oracle and sweeps per tile as a function of the Newton-sweep tolerance."""
import os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time
    from magprop_amd import _capi, engine, synth
    from oracle import c_oracle as co
    g = np.load(os.path.join(ROOT, "tests/golden/golden_synth.npz")); f = np.load(os.path.join(ROOT, "tests/golden/golden_flagscan.npz"))
    tarr = engine.grid(None); cfg = co.cfg_synth()
    worst = 0.0; sw = []; mism = 0
    h = _capi.Handle(_capi.cfg_synth(), tarr); h.set_prior(synth.PRIOR_LOWER, synth.PRIOR_UPPER, synth.LOG_MASK)
    for k, name in enumerate(("Humped", "Classic", "Sloped", "Stuttering")):
        x, y, yerr, P = g[name + "_x"], g[name + "_y"], g[name + "_yerr"], g[name + "_pars"]
        h.set_dataset(k, x, y, yerr)
        ref, rst = co.lnprob_batch(cfg, P, tarr, x, y, yerr, synth.PRIOR_LOWER, synth.PRIOR_UPPER, synth.LOG_MASK)
        out, st = h.lnprob_batch(P, ds_id=k, want_status=True)
        ok = np.isfinite(ref); mism += int((st != rst).sum())
        worst = max(worst, np.max(np.abs(out[ok] - ref[ok]) / (np.abs(ref[ok]) + 10.0))); sw.append(h.last_mean_sweeps)
    x, y, yerr = g["Humped_x"], g["Humped_y"], g["Humped_yerr"]
    ref, rst = co.lnprob_batch(cfg, f["pars"], tarr, x, y, yerr, synth.PRIOR_LOWER, synth.PRIOR_UPPER, synth.LOG_MASK)
    out, st = h.lnprob_batch(f["pars"], ds_id=0, want_status=True)
    ok = np.isfinite(ref); mism += int((st != rst).sum())
    wide = np.max(np.abs(out[ok] - ref[ok]) / (np.abs(ref[ok]) + 10.0)); swide = h.last_mean_sweeps
    rng = np.random.default_rng(1)
    P = np.array([1, 5, -3, 2, -1, 0.0]) + 1e-4 * rng.standard_normal((1024, 6))
    h.lnprob_batch(P, ds_id=0); t0 = time.perf_counter()
    for _ in range(10): h.lnprob_batch(P, ds_id=0)
    dt = (time.perf_counter() - t0) / 10
    print(json.dumps({"tol": os.environ.get("MAGPROP_AMD_SWEEP_TOL"), "worst_clouds": worst, "worst_wide": wide, "status_mismatch": mism,
                      "sweeps_clouds": float(np.mean(sw)), "sweeps_wide": swide, "sweeps_truth": h.last_mean_sweeps, "ms_1024": dt * 1e3}))
else:
    for tol in ("1e-9", "1e-8", "1e-7", "1e-6", "1e-5"):
        env = dict(os.environ, MAGPROP_AMD_SWEEP_TOL=tol)
        print(subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip())

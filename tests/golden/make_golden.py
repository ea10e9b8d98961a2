#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the real reference.

Runs only in the development container, where the reference checkout is mounted read-only at
/root/reference (it does not exist on the GPU box and never travels).  Nothing from the
reference is copied: this script calls its functions and stores inputs + outputs as data.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py [--flag-scan N]

What is called (paths relative to /root/reference):
  synth variant : code/synthetic_datasets/funcs.py  model_lum (:146), ODEs (:75), init_conds (:51), tarr (:19)
                  code/synthetic_datasets/mcmc_eqns.py  lnprior (:28)
                  lnlike/lnprob cannot be called under numpy>=2 (`mod == 'flag'` on an ndarray, :22), so the
                  three arithmetic lines :16-17 and :25 are applied here around the real model_lum.
  lib variant   : magnetar/funcs.py  model_lc (:105), odes (:33), init_conds (:17)
                  magnetar/mcmc_eqns.py:37 chi-square line applied around the real model_lc.
  data recipe   : code/synthetic_datasets/generate_data.py:61-67 with np.random.seed(20261003 + k).
  in-tree fixtures (data the reference's tests hold): tests/test_data/odes_integrated_by_odeint.csv,
                  tests/test_data/model_light_curve.csv — decimated (every 20th row) copies.
`*_tight` arrays: the same reference code with its odeint call given rtol=atol=1e-12 (integrator noise
removed; see tight_lsoda).
Outputs: golden_synth.npz, golden_lib.npz, golden_flagscan.npz, golden_corners.npz, golden_libscan.npz, golden_longlc.npz,
golden_flagscan2.npz, golden_libscan2.npz, golden_rhs.npz, golden_libkw.npz, golden_swift.npz, golden_fig3.npz (the two models of
code/figure_3.py: piroott :40-102, bucciantini :105-165), golden_holdout.npz / golden_holdout2.npz (--only), MANIFEST.json.
`--only tight` (make_tight) adds to the scan fixtures the tight-integrator value of EVERY successful point and the
enumerated LSODA-noise points (`*lsoda_noise_idx`): where the reference's default run is itself off by more than the
SURVEY.md 8(c) contract 1e-5 + 2e-6 |ref|.
"""
import argparse
import contextlib
import io
import json
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "code", "synthetic_datasets"))
sys.path.insert(0, REF)

import funcs as sf            # noqa: E402  (reference, synth variant)
import mcmc_eqns as sm        # noqa: E402
import magnetar as lib        # noqa: E402  (reference, lib variant)
from scipy.integrate import odeint  # noqa: E402

GRB_PARS = {  # generate_data.py:10-15 (inputs to the reference; restated as data)
    "Humped": [1.0, 5.0, 1.0e-3, 100.0, 0.1, 1.0],
    "Classic": [1.0, 5.0, 1.0e-3, 1000.0, 0.1, 1.0],
    "Sloped": [1.0, 1.0, 1.0e-3, 100.0, 10.0, 10.0],
    "Stuttering": [1.0, 5.0, 1.0e-5, 100.0, 0.1, 100.0],
}
TRUTHS = {  # synth_mcmc.py:16-21
    "Humped": [1.0, 5.0, -3.0, 2.0, -1.0, 0.0],
    "Classic": [1.0, 5.0, -3.0, 3.0, -1.0, 0.0],
    "Sloped": [1.0, 1.0, -3.0, 2.0, 1.0, 1.0],
    "Stuttering": [1.0, 5.0, -5.0, 2.0, -1.0, 2.0],
}
TYPES = list(GRB_PARS)
SEED0 = 20261003
LOWER = np.array([1.0e-3, 0.69, -6.0, np.log10(50.0), -2.0, -1.0])   # mcmc_eqns.py:41
UPPER = np.array([10.0, 10.0, -2.0, np.log10(2000.0), 2.0, 3.0])     # mcmc_eqns.py:40
DECIM = 50


def quiet(fn, *a, **k):
    with contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def synth_dataset(name, seed):
    """generate_data.py:55-67 around the real model_lum, with a fixed legacy seed."""
    model = sf.model_lum(np.array(GRB_PARS[name]))
    np.random.seed(seed)
    inx = np.sort(np.random.randint(low=0, high=model.shape[1], size=50))
    x = model[0, inx].copy()
    y = model[1, inx].copy()
    yerr = 0.25 * y
    y += np.random.normal(loc=0.0, scale=yerr, size=len(yerr))
    return inx, x, y, yerr


@contextlib.contextmanager
def tight_lsoda(tol=1.0e-12, mxstep=100000):
    """Run the reference's model_lum with its odeint call (funcs.py:169-170) given tight tolerances:
    same reference RHS / luminosity / interpolation code, integrator noise removed."""
    orig = sf.odeint

    def wrapped(func, y0, t, **kw):
        kw.update(rtol=tol, atol=tol, mxstep=mxstep)
        return orig(func, y0, t, **kw)
    sf.odeint = wrapped
    try:
        yield
    finally:
        sf.odeint = orig


@contextlib.contextmanager
def tight_lsoda_lib(tol=1.0e-12, mxstep=100000):
    """The same for the library variant: magnetar/funcs.py:150-151 calls the odeint its module imported."""
    mod = sys.modules["magnetar.funcs"]
    orig = mod.odeint

    def wrapped(func, y0, t, **kw):
        kw.update(rtol=tol, atol=tol, mxstep=mxstep)
        return orig(func, y0, t, **kw)
    mod.odeint = wrapped
    try:
        yield
    finally:
        mod.odeint = orig


# SURVEY.md 8(c) contract against the reference at its DEFAULT integrator tolerance: |d| <= 1e-5 + 2e-6 |ref|.
# Where the reference's own two runs (default vs rtol = atol = 1e-12) differ by more than 0.9 of that, the default value
# is LSODA noise: such points are enumerated per fixture (`*lsoda_noise_idx`) and the tests judge them against the tight
# value only (every point with a tight value is held to 1e-7 + 1e-7 |tight| anyway).
REF_ATOL, REF_RTOL = 1.0e-5, 2.0e-6


def noise_idx(default, tight):
    default, tight = np.asarray(default, float), np.asarray(tight, float)
    both = np.isfinite(default) & np.isfinite(tight)
    bad = both & (np.abs(default - tight) > 0.9 * (REF_ATOL + REF_RTOL * np.abs(default)))
    return np.nonzero(bad)[0].astype(np.int32)


def _tight_synth_one(job):
    p, x, y, yerr = job
    with tight_lsoda():
        return synth_lnprob(p, x, y, yerr)[0]


def _tight_lib_one(job):
    p, x, y, yerr, kind = job
    with tight_lsoda_lib():
        return _libscan_one((p, x, y, yerr, kind))[0]


def _tight_lib_lnlike_one(job):
    import pandas as pd
    row, x, y, yerr, kind = job
    p = row[np.isfinite(row)]
    with tight_lsoda_lib():
        return lib.lnlike(p, pd.DataFrame({"t": x, "Lum50": y, "Lum50err": yerr}), kind)


def _resave(name, extra):
    path = os.path.join(HERE, name)
    g = dict(np.load(path))
    g.update(extra)
    np.savez_compressed(path, **g)


def make_tight():
    """Second pass over the scan fixtures: the reference re-run with a tight integrator at EVERY successful point
    (`*_tight`), and the enumerated LSODA-noise points of each fixture.  Existing arrays are kept as they are."""
    import multiprocessing as mp
    os.chdir(REF)
    report = {}
    with mp.Pool(8) as pool:
        def tight_synth(P, st, sets, ds=None):
            jobs = [(P[i], *(sets[ds[i]] if ds is not None else sets)) for i in range(len(P)) if st[i] == 0]
            out = np.full(len(P), np.nan)
            out[st == 0] = pool.map(_tight_synth_one, jobs, chunksize=8)
            return np.where(np.isfinite(out), out, np.nan)       # a point the tight run flags: no tight value
        # golden_synth: tight values exist; lists per type
        g = np.load(os.path.join(HERE, "golden_synth.npz"))
        extra = {}
        for name in TYPES:
            extra[f"{name}_lsoda_noise_idx"] = noise_idx(g[f"{name}_lnprob"], g[f"{name}_lnprob_tight"])
            report[f"synth/{name}"] = len(extra[f"{name}_lsoda_noise_idx"])
        with tight_lsoda():                                   # light curves: canonical sets and the prior-wide ones
            for name in TYPES:
                extra[f"{name}_lc_tight"] = sf.model_lum(np.array(GRB_PARS[name]))[:, ::DECIM]
            extra["wide_lc_tight"] = np.array([quiet(sf.model_lum, p)[1:, ::DECIM] for p in g["wide_pars_physical"]])
        _resave("golden_synth.npz", extra)
        # prior-wide scans, Humped then the other three datasets
        g = np.load(os.path.join(HERE, "golden_flagscan.npz"))
        hs = synth_dataset("Humped", SEED0)[1:]
        t = tight_synth(g["pars"], g["status"], hs)
        keep = np.isfinite(g["lnprob_tight"])
        assert np.array_equal(t[keep], g["lnprob_tight"][keep])   # the 400 values of the first pass reproduce
        _resave("golden_flagscan.npz", {"lnprob_tight": t, "lsoda_noise_idx": noise_idx(g["lnprob"], t)})
        report["flagscan"] = int(len(noise_idx(g["lnprob"], t)))
        g = np.load(os.path.join(HERE, "golden_flagscan2.npz"))
        sets = [synth_dataset(nm, SEED0 + 1 + i)[1:] for i, nm in enumerate(["Classic", "Sloped", "Stuttering"])]
        t = tight_synth(g["pars"], g["status"], sets, g["ds"])
        _resave("golden_flagscan2.npz", {"lnprob_tight": t, "lsoda_noise_idx": noise_idx(g["lnprob"], t)})
        report["flagscan2"] = int(len(noise_idx(g["lnprob"], t)))
        g = np.load(os.path.join(HERE, "golden_corners.npz"))
        t = tight_synth(g["pars"], g["status"], hs)
        _resave("golden_corners.npz", {"lnprob_tight": t, "lsoda_noise_idx": noise_idx(g["lnprob"], t)})
        report["corners"] = int(len(noise_idx(g["lnprob"], t)))
        g = np.load(os.path.join(HERE, "golden_longlc.npz"))
        extra = {}
        for n in (112, 410, 1944):
            extra[f"synth{n}_lsoda_noise_idx"] = noise_idx(g[f"synth{n}_lnprob"], g[f"synth{n}_lnprob_tight"])
            report[f"longlc/synth{n}"] = len(extra[f"synth{n}_lsoda_noise_idx"])
        x, y, yerr = g["libS1944_ds"]
        t = np.array(pool.map(_tight_lib_lnlike_one, [(np.asarray(p, float), x, y, yerr, "S") for p in g["libS1944_pars"]]))
        extra["libS1944_lnlike_tight"] = t
        extra["libS1944_lsoda_noise_idx"] = noise_idx(g["libS1944_lnlike"], t)
        report["longlc/libS1944"] = len(extra["libS1944_lsoda_noise_idx"])
        _resave("golden_longlc.npz", extra)
        # library variant: scans on both grids, the 6/7/8/9-parameter likelihoods
        gl = np.load(os.path.join(HERE, "golden_lib.npz"))
        for fname, kind in (("golden_libscan.npz", "L"), ("golden_libscan2.npz", "S")):
            g = np.load(os.path.join(HERE, fname))
            x, y, yerr = gl["ds_" + kind]
            st = g["status"]
            t = np.full(len(st), np.nan)
            t[st == 0] = pool.map(_tight_lib_one, [(p, x, y, yerr, kind) for p, s_ in zip(g["pars_physical"], st) if s_ == 0], chunksize=8)
            t = np.where(np.isfinite(t), t, np.nan)
            _resave(fname, {"lnlike_tight": t, "lsoda_noise_idx": noise_idx(g["lnlike"], t)})
            report[fname] = int(len(noise_idx(g["lnlike"], t)))
        extra = {}
        for kind in ("L", "S"):
            x, y, yerr = gl["ds_" + kind]
            t = np.array(pool.map(_tight_lib_lnlike_one, [(row, x, y, yerr, kind) for row in gl[f"lnlike_{kind}_pars"]]))
            extra[f"lnlike_{kind}_tight"] = t
            extra[f"lnlike_{kind}_lsoda_noise_idx"] = noise_idx(gl[f"lnlike_{kind}"], t)
            report[f"lib/lnlike_{kind}"] = len(extra[f"lnlike_{kind}_lsoda_noise_idx"])
        _resave("golden_lib.npz", extra)
    print("LSODA-noise points per fixture:", report)
    return report


def synth_lnprob(p, x, y, yerr):
    """mcmc_eqns.py:52-81 with :16-17, :22-25 restated around the real model_lum/lnprior."""
    if not np.isfinite(sm.lnprior(np.asarray(p))):
        return -np.inf, 3
    arr = np.array(p, dtype=float)
    arr[2:] = 10.0 ** arr[2:]
    mod = quiet(sf.model_lum, arr, xdata=x)
    if isinstance(mod, str):
        return -np.inf, 1
    ll = -0.5 * np.sum(((y - mod) / yerr) ** 2.0)
    if not np.isfinite(ll):
        return -np.inf, 2
    return ll, 0


def synth_param_cloud(name, rng):
    """64 sampler-coordinate vectors: near-truth, +-5 %, prior-wide, edges."""
    t = np.array(TRUTHS[name])
    out = [t.copy()]
    out += [t + 1.0e-4 * rng.standard_normal(6) for _ in range(23)]          # synth_mcmc.py:175-176
    out += [t + 0.05 * np.maximum(np.abs(t), 0.5) * rng.standard_normal(6) for _ in range(12)]
    out += [LOWER + (UPPER - LOWER) * rng.random(6) for _ in range(22)]
    e = t.copy(); e[5] = 3.5; out.append(e)                                  # outside prior (upper)
    e = t.copy(); e[0] = 1.0e-4; out.append(e)                               # outside prior (lower)
    e = t.copy(); e[1] = 0.69; out.append(e)                                 # on the lower P edge
    e = t.copy(); e[3] = np.log10(2000.0); out.append(e)                     # on the upper Rdisc edge
    out.append(np.array([1.8171068, 3.68147895, -2.61786801, 1.99840102, -0.33083576, 2.95613803]))  # SURVEY flag
    out.append(np.array([10.0, 0.69, -2.0, np.log10(50.0), -2.0, 3.0]))      # prior corner
    return np.array(out[:64])


def make_synth():
    tarr = sf.tarr.copy()
    g = {"tarr_first_last": np.array([tarr[0], tarr[1], tarr[-2], tarr[-1]]), "prior_lower": LOWER,
         "prior_upper": UPPER, "decim": np.array(DECIM)}
    rng = np.random.default_rng(SEED0)
    for k, name in enumerate(TYPES):
        inx, x, y, yerr = synth_dataset(name, SEED0 + k)
        g[f"{name}_inx"], g[f"{name}_x"], g[f"{name}_y"], g[f"{name}_yerr"] = inx, x, y, yerr
        # full light curve + trajectory of the canonical parameter set
        pars = np.array(GRB_PARS[name])
        lc = sf.model_lum(pars)
        g[f"{name}_lc"] = lc[:, ::DECIM]
        y0 = sf.init_conds(pars[2], pars[1])
        args = (pars[0], pars[2], pars[3], pars[4], pars[5], 10.0, 0.1, 1.0, 0.9)
        soln, info = odeint(sf.ODEs, y0, tarr, args=args, full_output=True)
        g[f"{name}_traj"] = soln[::DECIM].T.copy()
        g[f"{name}_lsoda_counts"] = np.array([info["nst"][-1], info["nfe"][-1], info["nje"][-1]])
        tight = odeint(sf.ODEs, y0, tarr, args=args, rtol=1e-12, atol=1e-12, mxstep=5000)
        g[f"{name}_traj_tight"] = tight[::DECIM].T.copy()
        # posterior values on the parameter cloud
        P = synth_param_cloud(name, rng)
        lnp = np.empty(len(P)); st = np.empty(len(P), dtype=np.int32)
        for i, p in enumerate(P):
            lnp[i], st[i] = synth_lnprob(p, x, y, yerr)
        g[f"{name}_pars"], g[f"{name}_lnprob"], g[f"{name}_status"] = P, lnp, st
        with tight_lsoda():
            g[f"{name}_lnprob_tight"] = np.array([synth_lnprob(p, x, y, yerr)[0] if s == 0 else np.nan
                                                  for p, s in zip(P, st)])
        print(name, "lnprob(truth) =", repr(lnp[0]), "flags:", int((st == 1).sum()), "prior:", int((st == 3).sum()))
    # a few prior-wide light curves (model_lum with xdata=None) for curve-level parity
    wide = []
    wide_lc = []
    while len(wide) < 6:
        p = LOWER + (UPPER - LOWER) * rng.random(6)
        arr = p.copy(); arr[2:] = 10.0 ** arr[2:]
        lc = quiet(sf.model_lum, arr)
        if isinstance(lc, str):
            continue
        wide.append(arr); wide_lc.append(lc[1:, ::DECIM])
    g["wide_pars_physical"] = np.array(wide)
    g["wide_lc"] = np.array(wide_lc)
    np.savez_compressed(os.path.join(HERE, "golden_synth.npz"), **g)
    return g


def make_flagscan(n):
    """Oracle flag status over the uniform prior box (Humped dataset): validates the deterministic rule."""
    rng = np.random.default_rng(SEED0 + 77)
    _, x, y, yerr = synth_dataset("Humped", SEED0)
    P = LOWER + (UPPER - LOWER) * rng.random((n, 6))
    lnp = np.empty(n); st = np.empty(n, dtype=np.int32)
    for i, p in enumerate(P):
        lnp[i], st[i] = synth_lnprob(p, x, y, yerr)
        if i % 200 == 0:
            print("flagscan", i, n, flush=True)
    ntight = min(n, 400)
    tight = np.full(n, np.nan)
    with tight_lsoda():
        for i in range(ntight):
            if st[i] == 0:
                tight[i] = synth_lnprob(P[i], x, y, yerr)[0]
    np.savez_compressed(os.path.join(HERE, "golden_flagscan.npz"), pars=P, lnprob=lnp, status=st,
                        lnprob_tight=tight)
    print("flag rate", (st == 1).mean())


def _scan_one(job):
    p, x, y, yerr = job
    return synth_lnprob(p, x, y, yerr)


def make_flagscan2(n_each=1500):
    """The same scan against the other three synthetic datasets (a different posterior surface each), 3 x n_each more
    reference evaluations over the uniform prior box."""
    import multiprocessing as mp
    rng = np.random.default_rng(SEED0 + 177)
    names = ["Classic", "Sloped", "Stuttering"]
    P = LOWER + (UPPER - LOWER) * rng.random((3 * n_each, 6))
    ds = np.repeat(np.arange(3), n_each).astype(np.int32)
    sets = [synth_dataset(nm, SEED0 + 1 + i)[1:] for i, nm in enumerate(names)]   # seeds as in make_synth: SEED0 + k
    jobs = [(P[i], *sets[ds[i]]) for i in range(len(P))]
    with mp.Pool(8) as pool:
        res = pool.map(_scan_one, jobs, chunksize=20)
    lnp = np.array([r[0] for r in res])
    st = np.array([r[1] for r in res], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "golden_flagscan2.npz"), pars=P, ds=ds, ds_names=np.array(names), lnprob=lnp, status=st)
    print("flagscan2 flag rate", (st == 1).mean(), "n", len(P))


HOLDOUT_SEED = 41020261   # generated at commit de88a87 (round 4), BEFORE the last ~20 policy commits of that round: those commits
                          # were accepted with "hold-out summaries unchanged" among their gates, so this set served as an
                          # acceptance gate while the policy was still moving -- it is a regression set, not an untouched one
HOLDOUT2_SEED = 51020265  # round 5: drawn after every constant of the stride policy was final (no policy commit follows it);
                          # consulted ONCE, by tests/test_gpu_holdout.py, after it was written


def make_holdout(n_synth=900, n_lib=600, seed=HOLDOUT_SEED, out_name="golden_holdout.npz"):
    """HOLD-OUT sets.  golden_holdout.npz (round 4): prior-wide points drawn while the last policy changes of that round were
    still being accepted against it (see HOLDOUT_SEED): a regression gate.  golden_holdout2.npz (round 5, --only holdout2):
    a fresh seed after the policy was final.  4 x n_synth prior-wide points,
    one block per synthetic dataset (seeds as in make_synth), and 2 x n_lib library-variant points (grids "L" and "S") over
    the library's prior box; at every point the reference at its default LSODA tolerance AND at rtol = atol = 1e-12, plus
    the enumerated LSODA-noise points.  tests/test_gpu_holdout.py holds the product defaults to both bounds and to exact
    status; tests/test_oracle.py does the same for the serial restatement."""
    import multiprocessing as mp
    import pandas as pd
    HOLDOUT_SEED_ = seed
    rng = np.random.default_rng(seed)
    P = LOWER + (UPPER - LOWER) * rng.random((4 * n_synth, 6))
    ds = np.repeat(np.arange(4), n_synth).astype(np.int32)
    sets = [synth_dataset(nm, SEED0 + i)[1:] for i, nm in enumerate(TYPES)]
    jobs = [(P[i], *sets[ds[i]]) for i in range(len(P))]
    with mp.Pool(8) as pool:
        res = pool.map(_scan_one, jobs, chunksize=20)
        lnp = np.array([r[0] for r in res]); st = np.array([r[1] for r in res], dtype=np.int32)
        tight = np.full(len(P), np.nan)
        ok = np.nonzero(st == 0)[0]
        tight[ok] = pool.map(_tight_synth_one, [jobs[i] for i in ok], chunksize=20)
    out = {"synth_pars": P, "synth_ds": ds, "synth_ds_names": np.array(TYPES), "synth_lnprob": lnp, "synth_status": st,
           "synth_lnprob_tight": tight, "synth_lsoda_noise_idx": noise_idx(lnp, tight), "seed": np.array([HOLDOUT_SEED_])}
    print("holdout synth: flags", int((st == 1).sum()), "noise points", len(out["synth_lsoda_noise_idx"]), "of", len(P), flush=True)
    os.chdir(REF)
    lims = pd.read_csv(os.path.join(REF, "magnetar/mcmc_limits.csv"), index_col="pars")
    lo, hi = lims["lower"].values[:6], lims["upper"].values[:6]
    gl = np.load(os.path.join(HERE, "golden_lib.npz"))
    for kind in ("L", "S"):
        x, y, yerr = gl["ds_" + kind]
        Pl = lo + (hi - lo) * rng.random((n_lib, 6))
        phys = Pl.copy(); phys[:, 2:] = 10.0 ** phys[:, 2:]
        jobs = [(p, x, y, yerr, kind) for p in phys]
        with mp.Pool(8) as pool:
            res = pool.map(_libscan_one, jobs, chunksize=10)
            lnl = np.array([r[0] for r in res]); stl = np.array([r[1] for r in res], dtype=np.int32)
            tl = np.full(n_lib, np.nan)
            okl = np.nonzero(stl == 0)[0]
            tl[okl] = pool.map(_tight_lib_one, [jobs[i] for i in okl], chunksize=10)
        out.update({f"lib{kind}_pars_sampler": Pl, f"lib{kind}_pars_physical": phys, f"lib{kind}_lnlike": lnl, f"lib{kind}_status": stl,
                    f"lib{kind}_lnlike_tight": tl, f"lib{kind}_lsoda_noise_idx": noise_idx(lnl, tl)})
        print(f"holdout lib {kind}: flags", int((stl == 1).sum()), "noise points", len(out[f"lib{kind}_lsoda_noise_idx"]), "of", n_lib, flush=True)
    np.savez_compressed(os.path.join(HERE, out_name), **out)


def make_fig3(n_sets=12, n_rhs=400):
    """code/figure_3.py: the script's two right-hand sides, `piroott` (:40-102) and `bucciantini` (:105-165), and the two
    trajectories it integrates (:194-201).  The script plots at import: it is imported ONCE, in a scratch directory with a
    non-interactive backend (its `plots/` folder lands there), and its functions and results are used as they are.  Stored:
    the script's own (default LSODA) trajectories decimated, the same at rtol = atol = 1e-12, both models over `n_sets`
    further parameter sets (tight, decimated; the break-up verdict where the integrator gives up), and both right-hand
    sides at `n_rhs` states along and around those trajectories."""
    import importlib.util
    import tempfile
    os.environ["MPLBACKEND"] = "Agg"
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        spec = importlib.util.spec_from_file_location("ref_figure_3", os.path.join(REF, "code", "figure_3.py"))
        fig = importlib.util.module_from_spec(spec)
        with contextlib.redirect_stdout(io.StringIO()):
            spec.loader.exec_module(fig)
        os.chdir(cwd)
    tarr = fig.tarr
    dec = 10
    out = {"decim": np.array([dec]), "tarr": tarr[::dec],
           "script_pars": np.array([fig.B, fig.P, fig.MdiscI, fig.RdiscI, fig.epsilon, fig.delta]),
           "script_piroott": np.stack([fig.po_Mdisc, fig.po_omega])[:, ::dec],
           "script_bucciantini": np.stack([fig.b_Mdisc, fig.b_omega])[:, ::dec]}
    args0 = (fig.B, fig.MdiscI, fig.RdiscI, fig.epsilon, fig.delta)
    y0 = fig.init_conds(fig.MdiscI, fig.P)
    for name, f in (("piroott", fig.piroott), ("bucciantini", fig.bucciantini)):
        sol = odeint(f, y0, tarr, args=args0, rtol=1e-12, atol=1e-12, mxstep=100000)
        out[f"script_{name}_tight"] = sol.T[:, ::dec]
    rng = np.random.default_rng(SEED0 + 303)
    P = np.column_stack([rng.uniform(0.5, 5.0, n_sets), rng.uniform(1.0, 10.0, n_sets), 10.0 ** rng.uniform(-4.0, -2.0, n_sets),
                         rng.uniform(100.0, 1000.0, n_sets), 10.0 ** rng.uniform(-1.0, 1.0, n_sets), 10.0 ** rng.uniform(-0.5, 1.0, n_sets)])
    out["pars"] = P
    states = {"piroott": [], "bucciantini": []}
    for name, f in (("piroott", fig.piroott), ("bucciantini", fig.bucciantini)):
        trajs, ok = [], []
        for p in P:
            a_ = (p[0], p[2], p[3], p[4], p[5])
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                sol, info = odeint(f, fig.init_conds(p[2], p[1]), tarr, args=a_, rtol=1e-12, atol=1e-12, mxstep=100000, full_output=True)
            good = info["message"] == "Integration successful." and np.all(np.isfinite(sol))
            ok.append(good)
            trajs.append(sol.T[:, ::50] if good else np.full((2, len(tarr[::50])), np.nan))
            if good:
                idx = rng.integers(0, len(tarr), 40)
                for i in idx:
                    states[name].append((p, tarr[i], sol[i] * (1.0 + 0.02 * rng.standard_normal(2))))
        out[f"{name}_tight_dec50"] = np.array(trajs)
        out[f"{name}_ok"] = np.array(ok)
    for name, f in (("piroott", fig.piroott), ("bucciantini", fig.bucciantini)):
        st = states[name]
        pick = rng.choice(len(st), min(n_rhs, len(st)), replace=False)
        pp = np.array([st[i][0] for i in pick]); tt = np.array([st[i][1] for i in pick]); yy = np.array([st[i][2] for i in pick])
        dy = np.array([f(yy[i], tt[i], pp[i, 0], pp[i, 2], pp[i, 3], pp[i, 4], pp[i, 5]) for i in range(len(pick))])
        out[f"rhs_{name}_pars"], out[f"rhs_{name}_t"], out[f"rhs_{name}_y"], out[f"rhs_{name}_dydt"] = pp, tt, yy, dy
    np.savez_compressed(os.path.join(HERE, "golden_fig3.npz"), **out)
    print("fig3: sets ok", int(out["piroott_ok"].sum()), int(out["bucciantini_ok"].sum()), "of", n_sets, "; rhs points",
          len(out["rhs_piroott_t"]), len(out["rhs_bucciantini_t"]), flush=True)


def make_corners():
    """All 64 corners of the synth prior box on the Humped dataset, plus the largest rotation parameter the
    reference's own trajectory reaches (so tests can tell 'rode the break-up limit but LSODA survived')."""
    import itertools
    _, x, y, yerr = synth_dataset("Humped", SEED0)
    P = np.array([[UPPER[i] if b else LOWER[i] for i, b in enumerate(bits)] for bits in itertools.product([0, 1], repeat=6)])
    lnp = np.empty(len(P)); st = np.empty(len(P), dtype=np.int32); rot = np.full(len(P), np.nan)
    inertia = sf.I
    modW = (0.6 * sf.M * sf.c ** 2.0 * ((sf.GM / (sf.R * sf.c ** 2.0)) / (1.0 - 0.5 * (sf.GM / (sf.R * sf.c ** 2.0)))))
    for i, p in enumerate(P):
        lnp[i], st[i] = synth_lnprob(p, x, y, yerr)
        arr = p.copy(); arr[2:] = 10.0 ** arr[2:]
        y0 = sf.init_conds(arr[2], arr[1])
        soln = quiet(odeint, sf.ODEs, y0, sf.tarr, args=(arr[0], arr[2], arr[3], arr[4], arr[5], 10.0, 0.1, 1.0, 0.9))
        rot[i] = np.nanmax(0.5 * inertia * soln[:, 1] ** 2.0 / modW)
    np.savez_compressed(os.path.join(HERE, "golden_corners.npz"), pars=P, lnprob=lnp, status=st, max_rot=rot)
    print("corners: flags", int((st == 1).sum()), "rode break-up but survived", int(((st == 0) & (rot >= 0.27)).sum()))


def make_libscan(n=300):
    """Library variant over its own prior box (magnetar/mcmc_limits.csv rows 1-6, log rows un-logged): model_lc + the
    chi-square line of magnetar/mcmc_eqns.py:37 on the golden "L" dataset; 'flag' where model_lc returns it."""
    os.chdir(REF)
    import pandas as pd
    lims = pd.read_csv(os.path.join(REF, "magnetar/mcmc_limits.csv"), index_col="pars")
    lo, hi = lims["lower"].values[:6], lims["upper"].values[:6]
    gl = np.load(os.path.join(HERE, "golden_lib.npz"))
    x, y, yerr = gl["ds_L"]
    rng = np.random.default_rng(SEED0 + 303)
    P = lo + (hi - lo) * rng.random((n, 6))
    phys = P.copy(); phys[:, 2:] = 10.0 ** phys[:, 2:]
    lnl = np.empty(n); st = np.empty(n, dtype=np.int32)
    for i, p in enumerate(phys):
        mod = quiet(lib.model_lc, p, xdata=x, GRBtype="L")
        if isinstance(mod, str):
            lnl[i], st[i] = -np.inf, 1
        else:
            v = -0.5 * np.sum(((y - mod) / yerr) ** 2.0)
            lnl[i], st[i] = (v, 0) if np.isfinite(v) else (-np.inf, 2)
    np.savez_compressed(os.path.join(HERE, "golden_libscan.npz"), pars_sampler=P, pars_physical=phys, lnlike=lnl, status=st)
    print("libscan: flags", int((st == 1).sum()), "of", n)


def _libscan_one(job):
    p, x, y, yerr, kind = job
    mod = quiet(lib.model_lc, p, xdata=x, GRBtype=kind)
    if isinstance(mod, str):
        return -np.inf, 1
    v = -0.5 * np.sum(((y - mod) / yerr) ** 2.0)
    return (v, 0) if np.isfinite(v) else (-np.inf, 2)


def make_libscan2(n=900):
    """Library variant on the short-GRB grid (GRBtype "S", 1e-3..1e6 s) over its prior box, golden "S" dataset."""
    import multiprocessing as mp
    os.chdir(REF)
    import pandas as pd
    lims = pd.read_csv(os.path.join(REF, "magnetar/mcmc_limits.csv"), index_col="pars")
    lo, hi = lims["lower"].values[:6], lims["upper"].values[:6]
    gl = np.load(os.path.join(HERE, "golden_lib.npz"))
    x, y, yerr = gl["ds_S"]
    rng = np.random.default_rng(SEED0 + 404)
    P = lo + (hi - lo) * rng.random((n, 6))
    phys = P.copy(); phys[:, 2:] = 10.0 ** phys[:, 2:]
    with mp.Pool(8) as pool:
        res = pool.map(_libscan_one, [(p, x, y, yerr, "S") for p in phys], chunksize=10)
    lnl = np.array([r[0] for r in res]); st = np.array([r[1] for r in res], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "golden_libscan2.npz"), pars_sampler=P, pars_physical=phys, lnlike=lnl, status=st)
    print("libscan2 (S grid): flags", int((st == 1).sum()), "of", n)


def make_rhs(n=1500):
    """The right-hand sides themselves at random states: ODEs (synth, code/synthetic_datasets/funcs.py:75-142) and odes
    (lib, magnetar/funcs.py:33-101) over the prior box, all branches (capped Alfven radius, Rm < R, beyond break-up)."""
    rng = np.random.default_rng(SEED0 + 909)
    lo = np.array([1.0e-3, 0.69, 1.0e-6, 50.0, 1.0e-2, 1.0e-1])
    hi = np.array([10.0, 10.0, 1.0e-2, 2000.0, 1.0e2, 1.0e3])
    g = {}
    for name, fn, nn in (("synth", sf.ODEs, 10.0), ("lib", lib.odes, 1.0)):
        P = np.exp(np.log(lo) + (np.log(hi) - np.log(lo)) * rng.random((n, 6)))        # log-uniform, physical units
        t = 10.0 ** rng.uniform(-3.0 if name == "lib" else 0.0, 6.0, n)
        Md = 10.0 ** rng.uniform(17.0, 34.0, n)
        om = 10.0 ** rng.uniform(-1.2, 4.25, n)                                          # up to 1.8e4 rad/s: beyond break-up
        k = 0.9 * np.ones(n)
        k[: n // 5] = rng.uniform(0.3, 1.0, n // 5)                                      # other capping fractions
        al = 0.1 * np.ones(n)
        al[n // 5: 2 * n // 5] = rng.uniform(0.01, 0.5, n // 5)
        out = np.array([fn(np.array([Md[i], om[i]]), t[i], P[i, 0], P[i, 2], P[i, 3], P[i, 4], P[i, 5], nn, al[i], 1.0, k[i])
                        for i in range(n)])
        g[name + "_pars"], g[name + "_t"], g[name + "_y"] = P, t, np.stack([Md, om], axis=1)
        g[name + "_k"], g[name + "_alpha"], g[name + "_dydt"] = k, al, out
    np.savez_compressed(os.path.join(HERE, "golden_rhs.npz"), **g)
    print("rhs:", g["synth_dydt"][:2], g["lib_dydt"][:2])


def make_libkw():
    """model_lc with alpha / cs7 / k / n away from their defaults (magnetar/funcs.py:105-106): the reference integrates
    with the defaults (:150-151) and lights with the given values (:157-185)."""
    os.chdir(REF)
    g = {}
    pars = np.array(GRB_PARS["Humped"])
    wide = np.array([3.0, 1.2, 5.0e-2, 300.0, 2.0, 0.5])
    cases = [dict(alpha=0.3), dict(cs7=2.5), dict(k=0.5), dict(alpha=0.02, cs7=0.4, k=0.99, n=25.0),
             dict(alpha=0.5, k=0.3, dipeff=0.2, propeff=0.9, f_beam=7.0)]
    g["n_cases"] = np.array(len(cases))
    for i, kw in enumerate(cases):
        g[f"kw{i}_names"] = np.array(sorted(kw))
        g[f"kw{i}_values"] = np.array([kw[k] for k in sorted(kw)])
        for kind in ("L", "S"):
            for tag, p in (("humped", pars), ("wide", wide)):
                with np.errstate(all="ignore"):
                    g[f"kw{i}_{kind}_{tag}"] = quiet(lib.model_lc, p, GRBtype=kind, **kw)[:, ::DECIM]
    g["pars_humped"], g["pars_wide"] = pars, wide
    x = np.array([1.0, 3.7, 250.0, 9.0e4, 1.0e6])
    g["xdata"] = x
    g["kw3_L_humped_xdata"] = quiet(lib.model_lc, pars, xdata=x, GRBtype="L", **cases[3])
    np.savez_compressed(os.path.join(HERE, "golden_libkw.npz"), **g)
    same = all(np.array_equal(g[f"kw{i}_L_humped"], lib.model_lc(pars, GRBtype="L", **{k: v for k, v in cases[i].items()
               if k in ("dipeff", "propeff", "f_beam")})[:, ::DECIM]) for i in range(len(cases)))
    print("libkw: curves with alpha/cs7/k/n changed equal the default-keyword curves bit for bit:", same)


def make_lib():
    os.chdir(REF)  # magnetar/mcmc_eqns.py:55 reads a cwd-relative CSV
    import pandas as pd
    g = {}
    pars = np.array(GRB_PARS["Humped"])
    for kind in ("L", "S"):
        lc = lib.model_lc(pars, GRBtype=kind)
        g[f"lc_{kind}"] = lc[:, ::DECIM]
    g["lc_None_equals_L"] = np.array(np.array_equal(lib.model_lc(pars), lib.model_lc(pars, GRBtype="L")))
    # non-default keyword arguments (n only reaches the luminosity stage: funcs.py:150-151 vs :185)
    g["lc_L_n10_dip1_prop1"] = lib.model_lc(pars, GRBtype="L", n=10.0, dipeff=1.0, propeff=1.0)[:, ::DECIM]
    g["lc_L_fbeam"] = lib.model_lc(pars, GRBtype="L", f_beam=25.0, dipeff=0.3, propeff=0.7)[:, ::DECIM]
    # in-tree fixtures of the reference's own tests, decimated (data, not source)
    ode = pd.read_csv(os.path.join(REF, "tests/test_data/odes_integrated_by_odeint.csv"), index_col=False)
    mlc = pd.read_csv(os.path.join(REF, "tests/test_data/model_light_curve.csv"), index_col=False)
    g["intree_odes"] = np.array([ode["Mdisc"].values, ode["omega"].values, ode["t"].values])[:, ::20]
    g["intree_odes_pars"] = np.array([1.0, 1.0, 1.0e-3, 100.0, 1.0, 10.0])  # tests/test_funcs.py:36-41 (B,P,MdiscI,RdiscI,eps,delta)
    g["intree_lc"] = np.array([mlc["Ldip"].values, mlc["Lprop"].values, mlc["Ltot"].values, mlc["t"].values])[:, ::20]
    g["intree_lc_pars"] = pars                                               # tests/test_funcs.py:55
    # lib-variant likelihood on physical parameters (magnetar/mcmc_eqns.py:17-37) for 6/7/8/9 parameters
    rng = np.random.default_rng(SEED0 + 5)
    for kind in ("L", "S"):
        lc = lib.model_lc(pars, GRBtype=kind)
        np.random.seed(SEED0 + (10 if kind == "L" else 11))
        inx = np.sort(np.random.randint(0, lc.shape[1], 60))
        x = lc[0, inx] * (1.0 + 3.0e-4 * rng.random(60))      # off-knot times, inside the grid
        x = np.clip(x, lc[0, 0], lc[0, -1])
        y0 = np.interp(x, lc[0], lc[1])
        yerr = 0.2 * y0
        y = y0 + rng.normal(0.0, yerr)
        data = pd.DataFrame({"t": x, "Lum50": y, "Lum50err": yerr})
        g[f"ds_{kind}"] = np.array([x, y, yerr])
        cases = []
        vals = []
        for nd in (6, 7, 8, 9):
            for _ in range(4):
                p6 = pars * (1.0 + 0.2 * rng.standard_normal(6))
                p6 = np.abs(p6)
                extra = {6: [], 7: [rng.uniform(1, 600)], 8: [rng.uniform(0.01, 1), rng.uniform(0.01, 1)],
                         9: [rng.uniform(0.01, 1), rng.uniform(0.01, 1), rng.uniform(1, 600)]}[nd]
                p = np.concatenate([p6, extra])
                ll = lib.lnlike(p, data, kind)
                row = np.full(9, np.nan); row[:nd] = p
                cases.append(row); vals.append(ll)
        g[f"lnlike_{kind}_pars"] = np.array(cases)
        g[f"lnlike_{kind}"] = np.array(vals)
    # lnprior behaviour incl. the 7-parameter special case (magnetar/mcmc_eqns.py:64-79)
    lims = pd.read_csv(os.path.join(REF, "magnetar/mcmc_limits.csv"), index_col="pars")
    g["limits_lower"] = lims["lower"].values
    g["limits_upper"] = lims["upper"].values
    pri_cases = [
        [1.0, 5.0, -2.0, 2.0, 0.0, 0.0],
        [1.0, 5.0, -3.5, 2.0, 0.0, 0.0],
        [1.0, 5.0, -2.0, 2.0, 0.0, 0.0, 300.0],
        [1.0, 5.0, -2.0, 2.0, 0.0, 0.0, 0.5],
        [1.0, 5.0, -2.0, 2.0, 0.0, 0.0, 0.5, 0.5],
        [1.0, 5.0, -2.0, 2.0, 0.0, 0.0, 0.5, 1.5],
        [1.0, 5.0, -2.0, 2.0, 0.0, 0.0, 0.5, 0.5, 300.0],
        [1.0, 5.0, -2.0, 2.0, 0.0, 0.0, 0.5, 0.5, 0.5],
    ]
    rows = []; vals = []
    for p in pri_cases:
        row = np.full(9, np.nan); row[:len(p)] = p
        rows.append(row); vals.append(lib.lnprior(np.array(p)))
    g["lnprior_pars"] = np.array(rows)
    g["lnprior"] = np.array(vals)
    # fit statistics (magnetar/fit_stats.py:6-62) on the reference's own fixture (tests/test_funcs.py:167-182)
    ng = pd.read_csv(os.path.join(REF, "tests/test_data/noisy_gaussian.csv"))
    yd, ye, ym = ng["ydata"].values, ng["yerr"].values, ng["ymod"].values
    g["fit_ydata"], g["fit_yerr"], g["fit_ymod"] = yd, ye, ym
    g["fit_redchisq_sd"] = np.array(lib.redchisq(yd, ym, sd=ye))
    g["fit_redchisq_sd_deg6"] = np.array(lib.redchisq(yd, ym, deg=6, sd=ye))
    g["fit_redchisq_plain"] = np.array(lib.redchisq(yd, ym))
    g["fit_aicc_2"] = np.array(lib.aicc(yd, ym, ye, 2))
    g["fit_aicc_6"] = np.array(lib.aicc(yd, ym, ye, 6))
    np.savez_compressed(os.path.join(HERE, "golden_lib.npz"), **g)
    print("lib lnlike L:", g["lnlike_L"][:4])


def make_longlc():
    """Light curves of real-GRB length (data/SGRBS/*_raw.txt have 8..1944 rows; the synthetic sets all have 50):
    the reference's own interpolation + chi-square over 112 / 410 / 1944 observed times, both variants."""
    import pandas as pd
    rng = np.random.default_rng(SEED0 + 77)
    g = {}
    truth = np.array(TRUTHS["Humped"])
    model = sf.model_lum(np.array(GRB_PARS["Humped"]))
    for n in (112, 410, 1944):
        x = 10.0 ** rng.uniform(np.log10(1.2), np.log10(9.0e5), n)
        k = n // 4                                                     # early-time cluster: many points per grid interval
        x[:k] = rng.uniform(model[0, 3000], model[0, 3010], k)
        x = np.sort(x)
        x[0], x[-1] = model[0, 0], model[0, -1]                        # exactly the first / last knot
        y0 = np.interp(x, model[0], model[1])
        yerr = 0.25 * y0
        y = y0 + rng.normal(0.0, yerr)
        P = [truth.copy()] + [truth + 1.0e-4 * rng.standard_normal(6) for _ in range(5)]
        P += [truth + 0.05 * np.maximum(np.abs(truth), 0.5) * rng.standard_normal(6) for _ in range(3)]
        P += [LOWER + (UPPER - LOWER) * rng.random(6) for _ in range(3)]
        P = np.array(P)
        res = [synth_lnprob(p, x, y, yerr) for p in P]
        with tight_lsoda():
            tight = [synth_lnprob(p, x, y, yerr) for p in P]
        g[f"synth{n}_ds"] = np.array([x, y, yerr])
        g[f"synth{n}_pars"] = P
        g[f"synth{n}_lnprob"] = np.array([r[0] for r in res])
        g[f"synth{n}_status"] = np.array([r[1] for r in res], dtype=np.int32)
        g[f"synth{n}_lnprob_tight"] = np.array([r[0] for r in tight])
    # lib variant, short-GRB grid (1e-3 .. 1e6 s), 1944 points, physical parameters (magnetar/mcmc_eqns.py:17-37)
    os.chdir(REF)
    pars = np.array(GRB_PARS["Humped"])
    lc = lib.model_lc(pars, GRBtype="S")
    n = 1944
    x = np.sort(10.0 ** rng.uniform(-2.9, 5.9, n))
    x[0], x[-1] = lc[0, 0], lc[0, -1]
    y0 = np.interp(x, lc[0], lc[1])
    yerr = 0.2 * y0
    y = y0 + rng.normal(0.0, yerr)
    data = pd.DataFrame({"t": x, "Lum50": y, "Lum50err": yerr})
    P = np.array([np.abs(pars * (1.0 + 0.1 * rng.standard_normal(6))) for _ in range(8)])
    g["libS1944_ds"] = np.array([x, y, yerr])
    g["libS1944_pars"] = P
    g["libS1944_lnlike"] = np.array([lib.lnlike(p, data, "S") for p in P])
    np.savez_compressed(os.path.join(HERE, "golden_longlc.npz"), **g)
    print("longlc synth1944:", g["synth1944_lnprob"][:3], g["synth1944_lnprob_tight"][:3])


def _swift_times(name):
    """Time stamps (s since trigger) of a real Swift light curve, data/SGRBS/<name>_raw.txt (data only: the first
    column of the numeric rows; the files mix BAT / XRT blocks with text markers)."""
    t = []
    for line in open(os.path.join(REF, "data", "SGRBS", name + "_raw.txt")):
        parts = line.split()
        if len(parts) == 6:
            try:
                t.append(float(parts[0]))
            except ValueError:
                pass
    return np.sort(np.array(t))


def make_swift():
    """Light curves whose observation TIMES are those of real Swift bursts (densely clustered early, sparse late;
    SURVEY.md section 2 marks data/SGRBS/*_raw.txt as the realistic shapes): GRB 060614 (1 944 rows) and GRB 051016B
    (80 rows).  Fluxes are synthetic (the model of a canonical parameter set + noise, generate_data.py:61-67 style), the
    values are the reference's: synth variant on the times inside its grid [1, 1e6] s, library variant on the short-GRB
    grid [1e-3, 1e6] s with every time stamp."""
    import pandas as pd
    os.chdir(REF)
    rng = np.random.default_rng(SEED0 + 606)
    g = {}
    for name, grb in (("060614", "Humped"), ("051016B", "Classic")):
        t_all = _swift_times(name)
        g[f"swift_{name}_n_rows"] = np.array(len(t_all))
        # ---- synth variant, "L" grid
        x = t_all[(t_all >= sf.tarr[0]) & (t_all <= sf.tarr[-1])]
        model = sf.model_lum(np.array(GRB_PARS[grb]))
        y0 = np.interp(x, model[0], model[1])
        yerr = 0.25 * y0
        y = y0 + rng.normal(0.0, yerr)
        truth = np.array(TRUTHS[grb])
        P = [truth.copy()] + [truth + 1.0e-4 * rng.standard_normal(6) for _ in range(5)]
        P += [truth + 0.05 * np.maximum(np.abs(truth), 0.5) * rng.standard_normal(6) for _ in range(3)]
        P += [LOWER + (UPPER - LOWER) * rng.random(6) for _ in range(3)]
        P = np.array(P)
        res = [synth_lnprob(p, x, y, yerr) for p in P]
        with tight_lsoda():
            tight = np.array([synth_lnprob(p, x, y, yerr)[0] for p in P])
        g[f"swift_{name}_ds"] = np.array([x, y, yerr])
        g[f"swift_{name}_pars"] = P
        g[f"swift_{name}_lnprob"] = np.array([r[0] for r in res])
        g[f"swift_{name}_status"] = np.array([r[1] for r in res], dtype=np.int32)
        g[f"swift_{name}_lnprob_tight"] = np.where(np.isfinite(tight), tight, np.nan)
        g[f"swift_{name}_lsoda_noise_idx"] = noise_idx(g[f"swift_{name}_lnprob"], tight)
        # ---- library variant, "S" grid: all time stamps from 1e-3 s on
        pars = np.array(GRB_PARS[grb])
        lc = lib.model_lc(pars, GRBtype="S")
        xs = t_all[(t_all >= lc[0, 0]) & (t_all <= lc[0, -1])]
        y0 = np.interp(xs, lc[0], lc[1])
        yerr = 0.2 * y0
        y = y0 + rng.normal(0.0, yerr)
        data = pd.DataFrame({"t": xs, "Lum50": y, "Lum50err": yerr})
        Pl = np.array([np.abs(pars * (1.0 + 0.1 * rng.standard_normal(6))) for _ in range(6)])
        g[f"swift_{name}_libS_ds"] = np.array([xs, y, yerr])
        g[f"swift_{name}_libS_pars"] = Pl
        g[f"swift_{name}_libS_lnlike"] = np.array([lib.lnlike(p, data, "S") for p in Pl])
        with tight_lsoda_lib():
            g[f"swift_{name}_libS_lnlike_tight"] = np.array([lib.lnlike(p, data, "S") for p in Pl])
        g[f"swift_{name}_libS_lsoda_noise_idx"] = noise_idx(g[f"swift_{name}_libS_lnlike"], g[f"swift_{name}_libS_lnlike_tight"])
        print("swift", name, "rows", len(t_all), "on L grid", len(x), "on S grid", len(xs), "lnprob", g[f"swift_{name}_lnprob"][:2])
    np.savez_compressed(os.path.join(HERE, "golden_swift.npz"), **g)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flag-scan", type=int, default=1500)
    ap.add_argument("--only", choices=["all", "lib", "libkw", "corners", "libscan", "longlc", "flagscan2", "libscan2", "rhs", "tight", "swift", "holdout", "holdout2", "fig3"], default="all", help="regenerate only one file")
    a = ap.parse_args()
    import scipy, pandas
    if a.only == "all":
        make_synth()
        make_flagscan(a.flag_scan)
    if a.only in ("all", "corners"):
        make_corners()
    if a.only in ("all", "lib"):
        make_lib()
    if a.only in ("all", "libkw"):
        make_libkw()
    if a.only in ("all", "libscan"):
        make_libscan()
    if a.only in ("all", "longlc"):
        make_longlc()
    if a.only in ("all", "flagscan2"):
        make_flagscan2()
    if a.only in ("all", "libscan2"):
        make_libscan2()
    if a.only in ("all", "rhs"):
        make_rhs()
    noise_report = make_tight() if a.only in ("all", "tight") else None
    if a.only == "holdout":          # (not part of "all": the hold-out set is generated once, after a policy freeze)
        make_holdout()
    if a.only == "holdout2":         # round 5: fresh seed, policy final
        make_holdout(n_synth=600, n_lib=400, seed=HOLDOUT2_SEED, out_name="golden_holdout2.npz")
    if a.only in ("all", "swift"):
        make_swift()
    if a.only in ("all", "fig3"):
        make_fig3()
    manifest = {
        "generator": "tests/golden/make_golden.py",
        "reference": "sgibson91/magprop mounted at /root/reference (magnetar v%s)" % lib.__version__
        if hasattr(lib, "__version__") else "sgibson91/magprop mounted at /root/reference",
        "versions": {"python": sys.version.split()[0], "numpy": np.__version__, "scipy": scipy.__version__,
                     "pandas": pandas.__version__},
        "seed0": SEED0, "holdout_seed": HOLDOUT_SEED, "decimation": DECIM, "flag_scan_n": a.flag_scan,
    }
    old = {}
    if os.path.exists(os.path.join(HERE, "MANIFEST.json")):
        old = json.load(open(os.path.join(HERE, "MANIFEST.json")))
    manifest["lsoda_noise_points"] = noise_report if noise_report is not None else old.get("lsoda_noise_points")
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()

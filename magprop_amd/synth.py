"""Posterior of the synthetic-dataset variant, signatures of ``code/synthetic_datasets/mcmc_eqns.py``.

``lnlike(pars, x, y, yerr)`` (:5-25), ``lnprior(pars)`` (:28-49), ``lnprob(pars, x, y, yerr, fbad)``
(:52-81) — the callable ``synth_mcmc.py:180-185`` hands to ``emcee.EnsembleSampler``.  ``pars`` may be
2-D ``(n_walkers, 6)`` (emcee ``vectorize=True``): one kernel launch for the whole batch.
"""
import numpy as np

from . import _capi, engine

PRIOR_UPPER = np.array([10.0, 10.0, -2.0, np.log10(2000.0), 2.0, 3.0])   # :40
PRIOR_LOWER = np.array([1.0e-3, 0.69, -6.0, np.log10(50.0), -2.0, -1.0])  # :41
LOG_MASK = 0b111100                                                       # arr[2:] = 10**arr[2:]  (:16-17)


_cfg_cache = {}


def _cfg():
    """The synthetic variant's model configuration (one object per pair of solver defaults in force: building the ctypes
    structure and its cache key costs more than the rest of a call's Python)."""
    key = (_capi.DEFAULT_SWEEP_TOL, _capi.DEFAULT_MAX_STRIDE)
    c = _cfg_cache.get(key)
    if c is None:
        c = _cfg_cache[key] = _capi.cfg_synth()
        c._mp_key = engine._cfg_key(c)
    return c


def _evaluate(pars, x, y, yerr, lower, upper, device=-1, want_status=False):
    p = np.asarray(pars, dtype=np.float64)
    scalar = p.ndim == 1
    p2 = np.atleast_2d(p)
    if p2.shape[1] != 6:
        raise ValueError("pars must have 6 entries: B, P, log10 MdiscI, log10 RdiscI, log10 epsilon, log10 delta")
    eng = engine.acquire(_cfg(), None, device)      # (what `with engine.use(...)` does, without the generator: this is the hot entry)
    try:
        slot = eng.dataset_slot(x, y, yerr)
        eng.set_prior(lower, upper, LOG_MASK)
        out, st = eng.handle.lnprob_batch(p2, ds_id=slot, want_status=True)
    finally:
        engine.release(eng)
    if scalar:
        return (float(out[0]), int(st[0])) if want_status else float(out[0])
    return (out, st) if want_status else out


def lnlike(pars, x, y, yerr, device=-1):
    return _evaluate(pars, x, y, yerr, None, None, device)


def lnprior(pars):
    p = np.asarray(pars, dtype=np.float64)
    inside = np.all(p <= PRIOR_UPPER, axis=-1) & np.all(p >= PRIOR_LOWER, axis=-1)
    if p.ndim == 1:
        return 0.0 if inside else -np.inf
    return np.where(inside, 0.0, -np.inf)


def lnprob(pars, x, y, yerr, fbad=None, device=-1):
    """Parameter sets inside the prior whose likelihood is not finite are appended to ``fbad`` (:72-79)."""
    out, st = _evaluate(pars, x, y, yerr, PRIOR_LOWER, PRIOR_UPPER, device, want_status=True)
    if fbad is not None:
        bad = np.atleast_1d(st)
        rows = np.atleast_2d(np.asarray(pars, dtype=np.float64))[(bad == _capi.STATUS_FLAG) |
                                                                (bad == _capi.STATUS_NONFINITE)]
        if len(rows):
            with open(fbad, "a") as f:
                for r in rows:
                    f.write(", ".join(f"{v}" for v in r) + "\n")
    return out

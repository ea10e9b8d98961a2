"""Posterior of the library variant, signatures of ``magnetar/mcmc_eqns.py``.

``lnlike(pars, data, GRBtype)`` (:6-37), ``lnprior(pars, custom_lims=None)`` (:40-84),
``lnprob(pars, data, GRBtype, custom_lims=None)`` (:87-119).  ``pars`` may also be 2-D
``(n_walkers, ndim)``: the whole batch is one kernel launch (emcee ``vectorize=True``).

Reference quirk Q1: the reference's ``lnprob`` checks a log-space prior box
(``mcmc_limits.csv`` rows 3-6) but hands the same numbers to ``model_lc`` un-exponentiated.  Here
``lnprob`` implements the documented intent — parameters 3-6 are log10 in sampler coordinates and are
un-logged before the model — while ``lnlike`` keeps the reference behaviour (physical parameters).
"""
import os

import numpy as np

from . import _capi, engine

# Default prior box: the values of the reference's magnetar/mcmc_limits.csv:2-10 (rows B, P, log_MdiscI, log_RdiscI,
# log_epsilon, log_delta, dipeff, propeff, f_beam), kept in code so that nothing depends on the working directory
# (the reference re-reads a cwd-relative CSV on every call, magnetar/mcmc_eqns.py:55).
DEFAULT_LIMITS_LOWER = np.array([1.0e-3, 0.69, -3.0, np.log10(50.0), -1.0, -5.0, 0.01, 0.01, 1.0])
DEFAULT_LIMITS_UPPER = np.array([10.0, 10.0, -1.0, np.log10(2000.0), 3.0, np.log10(50.0), 1.0, 1.0, 600.0])
_limits_cache = {}
LIB_LOG_MASK = 0b111100


def _read_limits(path):
    """rows `pars,lower,upper`; cached per (path, mtime) — the reference re-reads it on every call (:55)."""
    try:
        key = (path, os.path.getmtime(path))
    except OSError:
        raise ValueError("Please provide a valid file path.")
    hit = _limits_cache.get(key)
    if hit is None:
        import pandas as pd
        try:
            lims = pd.read_csv(path, index_col="pars")
        except ValueError:
            raise ValueError("Please provide a valid file path.")
        hit = (lims["lower"].values.astype(float), lims["upper"].values.astype(float))
        _limits_cache[key] = hit
    return hit


def _bounds(ndim, custom_lims=None):
    lo, hi = (DEFAULT_LIMITS_LOWER, DEFAULT_LIMITS_UPPER) if custom_lims is None else _read_limits(custom_lims)
    if ndim == 7:  # :64-75: the 7th parameter is f_beam, the last row
        return np.append(lo[:6], lo[-1]), np.append(hi[:6], hi[-1])
    return lo[:ndim].copy(), hi[:ndim].copy()


def _columns(data):
    return (np.asarray(data["t"], dtype=np.float64), np.asarray(data["Lum50"], dtype=np.float64),
            np.asarray(data["Lum50err"], dtype=np.float64))


def _evaluate(pars, data, GRBtype, lower, upper, log_mask, device=-1, want_status=False):
    p = np.asarray(pars, dtype=np.float64)
    scalar = p.ndim == 1
    p2 = np.atleast_2d(p)
    if not 6 <= p2.shape[1] <= 9:
        raise ValueError("pars must have 6, 7, 8 or 9 entries")
    x, y, yerr = _columns(data)
    with engine.use(_capi.cfg_lib(), GRBtype, device) as eng:
        slot = eng.dataset_slot(x, y, yerr)
        eng.set_prior(lower, upper, log_mask)
        out, st = eng.handle.lnprob_batch(p2, ds_id=slot, want_status=True)
    if want_status:
        return (float(out[0]) if scalar else out), st
    return float(out[0]) if scalar else out


def lnlike(pars, data, GRBtype, device=-1, reference_quirk=False):
    """-0.5*chi^2 of model_lc against data (physical parameters; 6/7/8/9-parameter dispatch of :22-34).
    A failed integration gives -inf.  The reference has no branch for it: its ``model_lc`` returns the string 'flag'
    and ``ydata - 'flag'`` raises (:37); ``reference_quirk=True`` raises the same TypeError."""
    out, st = _evaluate(pars, data, GRBtype, None, None, 0, device, want_status=True)
    if reference_quirk and np.any(st == _capi.STATUS_FLAG):
        raise TypeError("unsupported operand type(s) for -: 'float' and 'str'  (model_lc returned 'flag', "
                        "magnetar/mcmc_eqns.py:37)")
    return out


def lnprior(pars, custom_lims=None):
    p = np.asarray(pars, dtype=np.float64)
    lo, hi = _bounds(p.shape[-1], custom_lims)
    inside = np.all(p >= lo, axis=-1) & np.all(p <= hi, axis=-1)
    if p.ndim == 1:
        return 0.0 if inside else -np.inf
    return np.where(inside, 0.0, -np.inf)


def lnprob(pars, data, GRBtype, custom_lims=None, device=-1, reference_quirk=False):
    """Default: the documented intent (module docstring) — box prior in sampler coordinates, parameters
    3-6 un-logged before the model.  ``reference_quirk=True`` evaluates what the reference's code literally does: the
    same box test, then the SAME numbers handed to ``model_lc`` un-exponentiated (:108-113), so a point inside the
    log-space box reaches the model with e.g. MdiscI = -2.5; the model state goes non-finite and the result is -inf
    (:115-116) wherever the reference would return -inf or NaN."""
    p = np.asarray(pars, dtype=np.float64)
    lo, hi = _bounds(p.shape[-1], custom_lims)
    return _evaluate(p, data, GRBtype, lo, hi, 0 if reference_quirk else LIB_LOG_MASK, device)

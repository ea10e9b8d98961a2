"""Fit statistics of the reference (`magnetar/fit_stats.py`): `redchisq` (:6-33) and `aicc` (:36-62), plus the
same numbers for a parameter vector straight from the kernel's chi-square (no model light curve on the host).

The first two are O(N_obs) array arithmetic and stay on the host with the reference's signatures; `fit_statistics`
is the epilogue used after a fit (plot_synth.py:183-189): chi^2 = -2 * lnlike comes from the HIP path.
"""
import numpy as np


def redchisq(ydata, ymod, deg=None, sd=None):
    """chi^2 (sum of squared, optionally sd-scaled residuals), divided by nu = N - 1 - deg when deg is given."""
    ydata = np.asarray(ydata, dtype=float)
    ymod = np.asarray(ymod, dtype=float)
    if sd is not None:
        chisq = np.sum(((ydata - ymod) / np.asarray(sd, dtype=float)) ** 2.0)
    else:
        chisq = np.sum((ydata - ymod) ** 2.0)
    if deg is not None:
        return chisq / (ydata.size - 1.0 - deg)
    return chisq


def aicc(ydata, ymod, yerr, Npars):
    """Corrected Akaike information criterion as the reference defines it: -chi^2 + 2k + 2k(k+1)/(N-k-1)."""
    ydata, ymod, yerr = (np.asarray(v, dtype=float) for v in (ydata, ymod, yerr))
    if not (ydata.size == ymod.size == yerr.size):
        raise ValueError("ydata, ymod and yerr should all be the same length")
    k, n = float(Npars), float(ydata.size)
    chisq = np.sum(np.square((ydata - ymod) / yerr))
    # (summed in the reference's order, -chi^2, then 2k, then the small-sample term, so that the value agrees with
    # magnetar/fit_stats.py:58-62 to the last bit and not just to rounding)
    return (-chisq + 2.0 * k) + (2.0 * k * (k + 1.0)) / (n - k - 1.0)


def fit_statistics(pars, x, y, yerr, variant="synth", GRBtype=None, device=-1):
    """{'chisq', 'redchisq', 'aicc', 'lnlike'} of sampler-coordinate `pars` against (x, y, yerr): the chi-square is the
    kernel's (-2 * lnlike); `pars` may be (n, ndim) for n parameter sets at once.  Failed models give inf."""
    from . import mcmc_eqns, synth
    p = np.asarray(pars, dtype=float)
    npars = p.shape[-1]
    n = np.asarray(x).size
    if variant == "synth":
        arr = np.atleast_2d(p).copy()
        # lnlike of the synthetic variant un-logs pars[2:] itself (code/synthetic_datasets/mcmc_eqns.py:16-17)
        ll = synth._evaluate(arr, x, y, yerr, None, None, device) if p.ndim == 2 else \
            np.atleast_1d(synth._evaluate(arr, x, y, yerr, None, None, device))
    elif variant == "lib":
        import pandas as pd
        data = pd.DataFrame({"t": x, "Lum50": y, "Lum50err": yerr})
        ll = np.atleast_1d(mcmc_eqns.lnlike(np.atleast_2d(p), data, GRBtype, device))
    else:
        raise ValueError("variant must be 'synth' or 'lib'")
    chisq = -2.0 * ll
    out = {"lnlike": ll, "chisq": chisq, "redchisq": chisq / (n - 1.0 - npars),
           "aicc": -chisq + 2.0 * npars + (2.0 * npars * (npars + 1.0)) / (n - npars - 1.0)}
    if p.ndim == 1:
        out = {k: float(v[0]) for k, v in out.items()}
    return out

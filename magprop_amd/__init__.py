"""magprop_amd — MI355X-native (gfx950, fp64) log-posterior hot path of sgibson91/magprop.

``from magprop_amd import *`` gives the names ``from magnetar import *`` gives for this path
(``init_conds``, ``model_lc``, ``lnlike``, ``lnprior``, ``lnprob``); ``magprop_amd.synth`` mirrors
``code/synthetic_datasets/mcmc_eqns.py`` and ``magprop_amd.funcs.model_lum`` its ``model_lum``.
Everything is evaluated by hand-written HIP kernels behind the C ABI of include/magprop_amd.h; there
is no CPU fallback.
"""
from . import _capi, engine, figure_3, fit_stats, funcs, mcmc_eqns, synth  # noqa: F401
from .ensemble import EnsembleSampler  # noqa: F401
from .logprob import LogProb  # noqa: F401
from ._capi import MagpropAmdError  # noqa: F401
from .fit_stats import aicc, redchisq  # noqa: F401
from .funcs import ODEs, init_conds, model_lc, model_lum, odes  # noqa: F401
from .mcmc_eqns import lnlike, lnprior, lnprob  # noqa: F401

__version__ = "0.1.0"
__all__ = ["init_conds", "model_lc", "model_lum", "redchisq", "aicc", "lnlike", "lnprior", "lnprob", "synth", "LogProb", "EnsembleSampler", "MagpropAmdError"]

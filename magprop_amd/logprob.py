"""Batched, device-resident log-posterior callables (the emcee ``log_prob_fn`` drop-in).

``LogProb`` binds a model variant, one or more observed light curves and the prior to a GPU once, and
is then called with ``(n_walkers, ndim)`` arrays (``emcee.EnsembleSampler(..., vectorize=True)``) or a
single parameter vector.  ``lnprob_device`` works on torch tensors already resident in HBM.
"""
import threading

import numpy as np

from . import _capi, engine, synth


class LogProb:
    def __init__(self, x, y, yerr, variant="synth", GRBtype=None, lower="default", upper="default", log_mask=None,
                 device=-1, fbad=None, sweep_tol=None, max_stride=None):
        """sweep_tol: Newton-sweep tolerance of the time-parallel solver (0 = the library default,
        include/magprop_amd.h MP_SWEEP_TOL_DEFAULT).  max_stride: grid intervals one step of the solver may span (1, 2, 4, 8;
        0 = the library default 8, MP_MAX_STRIDE_DEFAULT; 1 = every grid interval is a step)."""
        tol_kw = {} if sweep_tol is None else {"sweep_tol": float(sweep_tol)}   # None: _capi.DEFAULT_SWEEP_TOL
        if max_stride is not None:                                              # None: _capi.DEFAULT_MAX_STRIDE
            tol_kw["max_stride"] = int(max_stride)
        if variant == "synth":
            cfg = _capi.cfg_synth(**tol_kw)
            lo, hi, mask = synth.PRIOR_LOWER, synth.PRIOR_UPPER, synth.LOG_MASK
        elif variant == "lib":
            from . import mcmc_eqns
            cfg = _capi.cfg_lib(**tol_kw)
            lo, hi = mcmc_eqns._bounds(6)
            mask = mcmc_eqns.LIB_LOG_MASK
        else:
            raise ValueError("variant must be 'synth' or 'lib'")
        if not isinstance(lower, str):
            lo, hi = lower, upper
        if log_mask is not None:
            mask = log_mask
        # a private handle: the prior and datasets of this object are never swapped out by other callers
        self.handle = _capi.Handle(cfg, engine.grid(GRBtype), device)
        self.handle.set_prior(lo, hi, mask)
        self.n_datasets = 0
        self._lock = threading.Lock()     # host-buffer calls from several Python threads (ctypes drops the GIL)
        self.add_dataset(x, y, yerr)
        self.fbad = fbad

    def add_dataset(self, x, y, yerr):
        """Register a further light curve (mixed lengths allowed); returns its ds_id."""
        with self._lock:
            slot = self.n_datasets
            self.handle.set_dataset(slot, x, y, yerr)
            self.n_datasets += 1
        return slot

    def __call__(self, pars, ds_id=None):
        p = np.asarray(pars, dtype=np.float64)
        scalar = p.ndim == 1
        with self._lock:
            out, st = self.handle.lnprob_batch(p[None, :] if scalar else p, ds_id=ds_id, want_status=True)
        if self.fbad is not None:         # code/synthetic_datasets/mcmc_eqns.py:72-79
            rows = np.atleast_2d(p)[(st == _capi.STATUS_FLAG) | (st == _capi.STATUS_NONFINITE)]
            if len(rows):
                with open(self.fbad, "a") as f:
                    for r in rows:
                        f.write(", ".join(f"{v}" for v in r) + "\n")
        return float(out[0]) if scalar else out

    def lnprob_and_curves(self, pars, ds_id=None):
        """(lnprob[n], status[n], Ltot[n, n_grid]): the log-posterior together with every walker's model light curve
        on the grid in 1e50 erg/s (what model_lum(pars)[1] returns, code/synthetic_datasets/funcs.py:229-231); rows of
        walkers that did not finish (status != 0) are NaN."""
        p = np.atleast_2d(np.asarray(pars, dtype=np.float64))
        with self._lock:
            return self.handle.lnprob_batch(p, ds_id=ds_id, want_status=True, want_ltot=True)

    def lnprob_device(self, pars, out=None, ds_id=None, status=None, ltot=None):
        """pars: contiguous float64 CUDA tensor (n, ndim) on this handle's device.  Asynchronous on torch's
        current stream; returns the lnprob tensor.  ltot: optional (n, n_grid) float64 tensor that receives the model
        light curves (NaN rows for walkers that did not finish)."""
        import torch
        assert pars.is_cuda and pars.dtype == torch.float64 and pars.is_contiguous()
        n, nd = pars.shape
        if out is None:
            out = torch.empty(n, dtype=torch.float64, device=pars.device)
        stream = torch.cuda.current_stream(pars.device).cuda_stream
        self.handle.lnprob_batch_dev(pars.data_ptr(), n, nd, out.data_ptr(),
                                     d_ds_id=ds_id.data_ptr() if ds_id is not None else 0,
                                     d_status=status.data_ptr() if status is not None else 0,
                                     d_ltot=ltot.data_ptr() if ltot is not None else 0, stream=stream)
        return out

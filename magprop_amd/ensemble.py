"""Device-resident ensemble sampler with the slice of emcee's API that the reference driver uses.

`code/synthetic_datasets/synth_mcmc.py:175-226` does

    sampler = em.EnsembleSampler(Nwalk, Npars, lnprob, args=(x, y, yerr, fbad), pool=pool)
    sampler.run_mcmc(pos, Nstep, progress=True)
    sampler.chain[i, j, k]; sampler.lnprobability[i, j]; sampler.acceptance_fraction; sampler.get_autocorr_time()

`EnsembleSampler` here keeps positions, log-posteriors, acceptance counters and the chain in HBM and runs every
stretch-move half-step as one fused kernel (propose -> lnprob -> accept -> store), so there is no host round
trip per half-step.  Several independent ensembles (e.g. one per GRB dataset) can be advanced together.
"""
import ctypes as C
import sys

import numpy as np

from . import _capi, engine, synth


class EnsembleSampler:
    def __init__(self, nwalkers, ndim=6, x=None, y=None, yerr=None, variant="synth", GRBtype=None, seed=0, a=2.0,
                 datasets=None, lower="default", upper="default", log_mask=None, device=-1, target="posterior",
                 fbad=None, sweep_tol=None, max_stride=None, whole_step=True):
        """One ensemble on dataset (x, y, yerr), or one ensemble per entry of `datasets` = [(x, y, yerr), ...].
        whole_step: small ensembles run a whole step per launch (include/magprop_amd.h mp_sampler_set_whole_step; same
        chain bit for bit as one launch per half-step, which False selects).
        fbad: file that receives the proposals whose model failed, like the reference's lnprob(…, fbad)
        (code/synthetic_datasets/mcmc_eqns.py:72-79); written after every run_mcmc call."""
        if nwalkers % 2 or nwalkers < 2:
            raise ValueError("nwalkers must be even")            # emcee requires an even number too
        self.nwalkers, self.ndim = int(nwalkers), int(ndim)
        self._L = _capi.lib()
        tol_kw = {} if sweep_tol is None else {"sweep_tol": float(sweep_tol)}   # None: _capi.DEFAULT_SWEEP_TOL
        if max_stride is not None:                                              # None: _capi.DEFAULT_MAX_STRIDE
            tol_kw["max_stride"] = int(max_stride)
        if variant == "synth":
            cfg, lo, hi, mask = _capi.cfg_synth(**tol_kw), synth.PRIOR_LOWER, synth.PRIOR_UPPER, synth.LOG_MASK
        elif variant == "lib":
            from . import mcmc_eqns
            cfg = _capi.cfg_lib(**tol_kw)
            lo, hi = mcmc_eqns._bounds(ndim)
            mask = mcmc_eqns.LIB_LOG_MASK
        else:
            raise ValueError("variant must be 'synth' or 'lib'")
        if not isinstance(lower, str):
            lo, hi = lower, upper
        if log_mask is not None:
            mask = log_mask
        self.handle = _capi.Handle(cfg, engine.grid(GRBtype), device)
        self.handle.set_prior(lo, hi, mask)
        self._target = {"posterior": 0, "gaussian": 1}[target]
        if datasets is None:
            datasets = [(x, y, yerr)] if x is not None else []
        if self._target == 0 and not datasets:
            raise ValueError("a dataset is required")
        for k, (dx, dy, de) in enumerate(datasets):
            self.handle.set_dataset(k, dx, dy, de)
        self.nensembles = max(1, len(datasets))
        ids = np.arange(self.nensembles, dtype=np.int32)
        self._s = self._L.mp_sampler_create(self.handle._h, self.nwalkers, self.nensembles, self.ndim,
                                            ids.ctypes.data_as(C.POINTER(C.c_int32)), C.c_uint64(int(seed)),
                                            C.c_double(a), self._target)
        if not self._s:
            raise _capi.MagpropAmdError("mp_sampler_create failed: " + _capi.last_error())
        _capi.check(self._L.mp_sampler_set_whole_step(self._s, int(bool(whole_step))), "mp_sampler_set_whole_step")
        self.seed = int(seed)
        self._chain = None
        self._lnp = None
        self.iteration = 0
        self.fbad = fbad
        self._bad_written = 0
        self._warned_bad = False

    def close(self):
        if getattr(self, "_s", None):
            self._L.mp_sampler_destroy(self._s)
            self._s = None
        if getattr(self, "handle", None):
            self.handle.close()

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    @property
    def ntotal(self):
        return self.nwalkers * self.nensembles

    def run_mcmc(self, pos, nsteps, store=True, progress=False):
        """pos: (nwalkers, ndim) [or (nensembles*nwalkers, ndim)], or None to continue.  Returns the final positions."""
        if pos is not None:
            self.set_positions(pos)
        chain = lnp = None
        cp = lp = None
        if store and nsteps > 0:
            chain = np.empty((nsteps, self.ntotal, self.ndim))
            lnp = np.empty((nsteps, self.ntotal))
            cp, lp = chain.ctypes.data_as(C.POINTER(C.c_double)), lnp.ctypes.data_as(C.POINTER(C.c_double))
        _capi.check(self._L.mp_sampler_run(self._s, int(nsteps), cp, lp), "mp_sampler_run")
        if store and nsteps > 0:
            self._chain = chain if self._chain is None else np.concatenate([self._chain, chain])
            self._lnp = lnp if self._lnp is None else np.concatenate([self._lnp, lnp])
        self.iteration += int(nsteps)
        self._flush_fbad()
        return self.get_last_sample()[0]

    # ---- failed proposals (the reference's fbad file)
    def get_bad(self, first_row=0, max_rows=None):
        """(n_bad, pars[rows, ndim]): how many proposals inside the prior failed in the model so far (exact), and rows
        [first_row, first_row + max_rows) of the library's log of them, in sampler coordinates.  The log misses rows only
        if more than MP_BAD_WINDOW (65 536) proposals failed between two drains; a warning says so."""
        n_bad, n_logged = C.c_int64(0), C.c_int64(0)
        rc = self._L.mp_sampler_get_bad(self._s, 0, None, 0, C.byref(n_bad), C.byref(n_logged))
        if rc < 0:
            _capi.check(rc, "mp_sampler_get_bad")
        want = max(0, n_logged.value - int(first_row))
        if max_rows is not None:
            want = min(want, int(max_rows))
        buf = np.empty((want, self.ndim))
        if want:
            rows = self._L.mp_sampler_get_bad(self._s, int(first_row), buf.ctypes.data_as(C.POINTER(C.c_double)), want,
                                              None, None)
            if rows < 0:
                _capi.check(rows, "mp_sampler_get_bad")
            buf = buf[:rows]
        if n_logged.value < n_bad.value and not self._warned_bad:
            import warnings
            warnings.warn(f"{n_bad.value - n_logged.value} of {n_bad.value} failed proposals are missing from the fbad log "
                          "(more than 65 536 failed between two drains)", RuntimeWarning)
            self._warned_bad = True
        return int(n_bad.value), buf

    def _flush_fbad(self):
        if self.fbad is None or self._target != 0:
            return
        _, new = self.get_bad(first_row=self._bad_written)
        if len(new):
            with open(self.fbad, "a") as f:
                for r in new:
                    f.write(", ".join(f"{v}" for v in r) + "\n")
            self._bad_written += len(new)

    # ---- walker-sharded driving (magprop_amd.distributed.DistributedEnsembleSampler); device pointers as ints
    @property
    def n_slots(self):
        return self._L.mp_sampler_n_slots(self._s)

    @property
    def row_doubles(self):
        return self._L.mp_sampler_row_doubles(self._s)

    def set_positions(self, pos):
        p = np.ascontiguousarray(pos, dtype=np.float64)
        if p.shape != (self.ntotal, self.ndim):
            raise ValueError(f"pos must have shape {(self.ntotal, self.ndim)}")
        _capi.check(self._L.mp_sampler_set_positions(self._s, p.ctypes.data_as(C.POINTER(C.c_double))),
                    "mp_sampler_set_positions")

    def halfstep_shard(self, half, lo, hi, d_rows, stream=0):
        _capi.check(self._L.mp_sampler_halfstep_shard(self._s, int(half), int(lo), int(hi), C.c_void_p(d_rows or None),
                                                      C.c_void_p(stream or None)), "mp_sampler_halfstep_shard")

    def halfstep_apply(self, half, d_rows, d_chain_row=0, d_chain_lnp_row=0, stream=0):
        _capi.check(self._L.mp_sampler_halfstep_apply(self._s, int(half), C.c_void_p(d_rows), C.c_void_p(d_chain_row or None),
                                                      C.c_void_p(d_chain_lnp_row or None), C.c_void_p(stream or None)),
                    "mp_sampler_halfstep_apply")
        if half == 1:
            self.iteration += 1

    # whole-step protocol (include/magprop_amd.h mp_sampler_step_shard / _apply)
    @property
    def step_blocks(self):
        return self._L.mp_sampler_step_blocks(self._s)

    @property
    def step_row_doubles(self):
        return self._L.mp_sampler_step_row_doubles(self._s)

    def step_shard(self, lo, hi, d_rows, stream=0):
        _capi.check(self._L.mp_sampler_step_shard(self._s, int(lo), int(hi), C.c_void_p(d_rows or None), C.c_void_p(stream or None)),
                    "mp_sampler_step_shard")

    def step_apply(self, d_rows, d_chain_row=0, d_chain_lnp_row=0, stream=0):
        _capi.check(self._L.mp_sampler_step_apply(self._s, C.c_void_p(d_rows), C.c_void_p(d_chain_row or None),
                                                  C.c_void_p(d_chain_lnp_row or None), C.c_void_p(stream or None)),
                    "mp_sampler_step_apply")
        self.iteration += 1

    def get_last_sample(self):
        pos = np.empty((self.ntotal, self.ndim))
        lnp = np.empty(self.ntotal)
        acc = np.empty(self.ntotal, dtype=np.int64)
        done = C.c_int64(0)
        _capi.check(self._L.mp_sampler_get_state(self._s, pos.ctypes.data_as(C.POINTER(C.c_double)),
                                                 lnp.ctypes.data_as(C.POINTER(C.c_double)),
                                                 acc.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(done)),
                    "mp_sampler_get_state")
        return pos, lnp, acc

    # ---- emcee-shaped views (synth_mcmc.py:188-226 indexes chain[i, j, k], lnprobability[i, j])
    def get_chain(self):
        """(nsteps, nwalkers_total, ndim)"""
        return self._chain

    def get_log_prob(self):
        return self._lnp

    @property
    def chain(self):
        return None if self._chain is None else np.swapaxes(self._chain, 0, 1)

    @property
    def lnprobability(self):
        return None if self._lnp is None else self._lnp.T

    @property
    def acceptance_fraction(self):
        return self.get_last_sample()[2] / max(self.iteration, 1)

    def get_autocorr_time(self, c=5.0, tol=50, quiet=False):
        """emcee's default (quiet=False) raises when the chain is shorter than tol autocorrelation times; the
        reference calls it bare (code/synthetic_datasets/synth_mcmc.py:220)."""
        from .mcmc_io import integrated_time
        return integrated_time(self._chain, c=c, tol=tol, quiet=quiet)

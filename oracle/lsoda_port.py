"""oracle/lsoda_port.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy/scipy restatement of the reference's *own* algorithm for the hot path, i.e. with the
third-party integrator it actually uses: ``scipy.integrate.odeint`` (ODEPACK LSODA; the reference
pins scipy==1.3.0 in requirements.txt:9 and passes no tolerances, so rtol = atol ~ 1.49e-8,
mxstep = 500) driving a Python right-hand side ~10^4 times per light curve, followed by the
vectorised luminosity stage, ``interp1d`` and the chi-square.  It has the reference's cost
structure and is therefore what bench.py times as ``cpu_baseline`` (kind "port"); it is also
the loose (LSODA-noise-limited, ~1e-7) checker next to the tight C checker in mp_oracle.c.

Restated from (paths relative to the reference checkout):
  constants               magnetar/funcs.py:7-13, code/synthetic_datasets/funcs.py:12-19
  init_conds              magnetar/funcs.py:17-29, code/synthetic_datasets/funcs.py:51-71
  RHS                     magnetar/funcs.py:33-101, code/synthetic_datasets/funcs.py:75-142
  model_lc / model_lum    magnetar/funcs.py:105-220, code/synthetic_datasets/funcs.py:146-236
  lnlike/lnprior/lnprob   code/synthetic_datasets/mcmc_eqns.py:5-81, magnetar/mcmc_eqns.py:6-119
Both physics variants are one code path parametrised by ``Variant``.
Pinned by tests/test_oracle.py against golden vectors captured from the real reference
(tests/golden/make_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
from dataclasses import dataclass, replace

import numpy as np
from scipy.integrate import odeint

G = 6.674e-8
c = 3.0e10
R = 1.0e6
Msol = 1.99e33
M = 1.4 * Msol
GM = G * M
_beta = GM / (R * c ** 2.0)
modW = 0.6 * M * c ** 2.0 * (_beta / (1.0 - 0.5 * _beta))  # magnetar/funcs.py:75-76


@dataclass(frozen=True)
class Variant:
    inertia_factor: float
    rm_massflow_factor: float
    n_ode: float
    n_lum: float
    alpha: float = 0.1
    cs7: float = 1.0
    k: float = 0.9
    dipeff: float = 1.0
    propeff: float = 1.0
    f_beam: float = 1.0
    nacc_lum_threshold: float = 0.27
    lprop_gm_term: bool = True

    @property
    def inertia(self):
        return self.inertia_factor * M * R ** 2.0


SYNTH = Variant(0.35, 3.0, 10.0, 10.0)
LIB = Variant(0.8, 1.0, 1.0, 1.0, dipeff=0.05, propeff=0.4, f_beam=1.0, nacc_lum_threshold=0.0,
              lprop_gm_term=False)

SYNTH_PRIOR_LOWER = np.array([1.0e-3, 0.69, -6.0, np.log10(50.0), -2.0, -1.0])
SYNTH_PRIOR_UPPER = np.array([10.0, 10.0, -2.0, np.log10(2000.0), 2.0, 3.0])


def grid(kind="L"):
    """magnetar/funcs.py:132-137; the synth variant always uses the "L" grid (funcs.py:19)."""
    if kind == "S":
        return np.logspace(-3.0, 6.0, num=10001, base=10.0)
    if kind in ("L", None):
        return np.logspace(0.0, 6.0, num=10001, base=10.0)
    raise ValueError("Please provide a valid value for GRBtype.\nOptions are: L, S, or None.")


def _walker_constants(v, B, MdiscI, RdiscI, epsilon, delta):
    tvisc = (RdiscI * 1.0e5) / (v.alpha * v.cs7 * 1.0e7)
    mu = 1.0e15 * B * R ** 3.0
    M0 = delta * MdiscI * Msol
    tfb = epsilon * tvisc
    return tvisc, mu, M0, tfb


def rhs(y, t, v, n, tvisc, mu, M0, tfb):
    """Scalar right-hand side called by LSODA (the reference's hot loop #2)."""
    Mdisc, omega = y
    inertia = v.inertia
    Rm = (mu ** (4.0 / 7.0)) * (GM ** (-1.0 / 7.0)) * ((v.rm_massflow_factor * Mdisc) / tvisc) ** (-2.0 / 7.0)
    Rc = (GM / omega ** 2.0) ** (1.0 / 3.0)
    Rlc = c / omega
    if Rm >= v.k * Rlc:
        Rm = v.k * Rlc
    w = (Rm / Rc) ** 1.5
    rot_param = 0.5 * inertia * omega ** 2.0 / modW
    Ndip = (-1.0 * mu ** 2.0 * omega ** 3.0) / (6.0 * c ** 3.0)
    eta2 = 0.5 * (1.0 + np.tanh(n * (w - 1.0)))
    eta1 = 1.0 - eta2
    Mdotprop = eta2 * (Mdisc / tvisc)
    Mdotacc = eta1 * (Mdisc / tvisc)
    Mdotfb = (M0 / tfb) * ((t + tfb) / tfb) ** (-5.0 / 3.0)
    if rot_param > 0.27:
        Nacc = 0.0
    elif Rm >= R:
        Nacc = (GM * Rm) ** 0.5 * (Mdotacc - Mdotprop)
    else:
        Nacc = (GM * R) ** 0.5 * (Mdotacc - Mdotprop)
    return Mdotfb - Mdotacc - Mdotprop, (Nacc + Ndip) / inertia


def rhs_per_call_constants(y, t, v, n, B, MdiscI, RdiscI, epsilon, delta):
    """The same right-hand side with the reference's COST structure: the reference derives the walker constants and the
    star's binding energy afresh inside every call (code/synthetic_datasets/funcs.py:98-118), ~9 800 times per light
    curve.  Same numbers as `rhs`; used by the cpu_baseline timing (reference_cost=True)."""
    tvisc, mu, M0, tfb = _walker_constants(v, B, MdiscI, RdiscI, epsilon, delta)
    beta = GM / (R * c ** 2.0)
    binding = 0.6 * M * c ** 2.0 * (beta / (1.0 - 0.5 * beta))
    Mdisc, omega = y
    inertia = v.inertia
    Rm = (mu ** (4.0 / 7.0)) * (GM ** (-1.0 / 7.0)) * ((v.rm_massflow_factor * Mdisc) / tvisc) ** (-2.0 / 7.0)
    Rc = (GM / omega ** 2.0) ** (1.0 / 3.0)
    Rlc = c / omega
    if Rm >= v.k * Rlc:
        Rm = v.k * Rlc
    w = (Rm / Rc) ** 1.5
    rot_param = (0.5 * inertia * omega ** 2.0) / binding
    Ndip = (-1.0 * mu ** 2.0 * omega ** 3.0) / (6.0 * c ** 3.0)
    eta2 = 0.5 * (1.0 + np.tanh(n * (w - 1.0)))
    eta1 = 1.0 - eta2
    Mdotprop = eta2 * (Mdisc / tvisc)
    Mdotacc = eta1 * (Mdisc / tvisc)
    Mdotfb = (M0 / tfb) * ((t + tfb) / tfb) ** (-5.0 / 3.0)
    if rot_param > 0.27:
        Nacc = 0.0
    elif Rm >= R:
        Nacc = (GM * Rm) ** 0.5 * (Mdotacc - Mdotprop)
    else:
        Nacc = (GM * R) ** 0.5 * (Mdotacc - Mdotprop)
    return Mdotfb - Mdotacc - Mdotprop, (Nacc + Ndip) / inertia


def integrate(pars, tarr, v=SYNTH, rtol=None, atol=None, mxstep=0, reference_cost=False):
    """(soln[n,2] or None if LSODA reports anything but success, info)."""
    B, P, MdiscI, RdiscI, epsilon, delta = pars[:6]
    y0 = (MdiscI * Msol, (2.0 * np.pi) / (1.0e-3 * P))
    wc = _walker_constants(v, B, MdiscI, RdiscI, epsilon, delta)
    fn, args = (rhs_per_call_constants, (v, v.n_ode, B, MdiscI, RdiscI, epsilon, delta)) if reference_cost else \
        (rhs, (v, v.n_ode) + wc)
    with np.errstate(all="ignore"):
        soln, info = odeint(fn, y0, tarr, args=args, full_output=True,
                            rtol=rtol, atol=atol, mxstep=mxstep, printmessg=False)
    if info["message"] != "Integration successful.":
        return None, info
    return soln, info


def luminosity(soln, pars, v=SYNTH, reference_cost=False):
    """Luminosity stage: (Ltot, Lprop, Ldip) in erg/s on the grid.  Vectorised; with reference_cost=True the accretion
    torque is filled element by element in the interpreter, as the reference does over its 10 001 grid points
    (code/synthetic_datasets/funcs.py:204-212, ~8 ms per light curve)."""
    B, P, MdiscI, RdiscI, epsilon, delta = pars[:6]
    tvisc, mu, M0, tfb = _walker_constants(v, B, MdiscI, RdiscI, epsilon, delta)
    Mdisc = soln[:, 0]
    omega = soln[:, 1]
    with np.errstate(all="ignore"):
        Rm = (mu ** (4.0 / 7.0)) * (GM ** (-1.0 / 7.0)) * ((v.rm_massflow_factor * Mdisc) / tvisc) ** (-2.0 / 7.0)
        Rc = (GM / omega ** 2.0) ** (1.0 / 3.0)
        Rlc = c / omega
        Rm = np.where(Rm >= v.k * Rlc, v.k * Rlc, Rm)
        w = (Rm / Rc) ** 1.5
        rot_param = 0.5 * v.inertia * omega ** 2.0 / modW
        eta2 = 0.5 * (1.0 + np.tanh(v.n_lum * (w - 1.0)))
        eta1 = 1.0 - eta2
        Mdotprop = eta2 * (Mdisc / tvisc)
        Mdotacc = eta1 * (Mdisc / tvisc)
        if reference_cost:
            Nacc = np.zeros_like(Mdisc)
            for j in range(Nacc.size):
                if rot_param[j] > v.nacc_lum_threshold:
                    Nacc[j] = 0.0
                elif Rm[j] >= R:
                    Nacc[j] = (GM * Rm[j]) ** 0.5 * (Mdotacc[j] - Mdotprop[j])
                else:
                    Nacc[j] = (GM * R) ** 0.5 * (Mdotacc[j] - Mdotprop[j])
        else:
            arm = np.where(Rm >= R, (GM * Rm) ** 0.5, (GM * R) ** 0.5)
            Nacc = np.where(rot_param > v.nacc_lum_threshold, 0.0, arm * (Mdotacc - Mdotprop))
        Ldip = v.dipeff * ((mu ** 2.0 * omega ** 4.0) / (6.0 * c ** 3.0))
        Ldip = np.where(Ldip <= 0.0, 0.0, Ldip)
        Ldip = np.where(np.isfinite(Ldip), Ldip, 0.0)
        if v.lprop_gm_term:
            Lprop = v.propeff * ((-1.0 * Nacc * omega) - ((GM / Rm) * eta2 * (Mdisc / tvisc)))
        else:
            Lprop = v.propeff * (-1.0 * Nacc * omega)
        Lprop = np.where(Lprop <= 0.0, 0.0, Lprop)
        Lprop = np.where(np.isfinite(Lprop), Lprop, 0.0)
    return v.f_beam * (Ldip + Lprop), Lprop, Ldip


def model(pars, tarr, xdata=None, v=SYNTH, reference_cost=False):
    """model_lc / model_lum: 'flag' | L(xdata)/1e50 | array([tarr, Ltot, Lprop, Ldip]/1e50)."""
    soln, _ = integrate(pars, tarr, v, reference_cost=reference_cost)
    if soln is None:
        return "flag"
    Ltot, Lprop, Ldip = luminosity(soln, pars, v, reference_cost=reference_cost)
    if xdata is None:
        return np.array([tarr, Ltot / 1.0e50, Lprop / 1.0e50, Ldip / 1.0e50])
    xdata = np.asarray(xdata, dtype=float)
    if np.any(xdata < tarr[0]) or np.any(xdata > tarr[-1]):
        raise ValueError("A value in x_new is outside the interpolation range.")
    return np.interp(xdata, tarr, Ltot) / 1.0e50


def with_extra_pars(v, pars):
    """7/8/9-parameter likelihoods, magnetar/mcmc_eqns.py:22-34."""
    n = len(pars)
    if n == 7:
        return replace(v, f_beam=pars[6])
    if n == 8:
        return replace(v, dipeff=pars[6], propeff=pars[7])
    if n == 9:
        return replace(v, dipeff=pars[6], propeff=pars[7], f_beam=pars[8])
    return v


def lnlike_physical(pars, tarr, x, y, yerr, v=SYNTH, reference_cost=False):
    """(-0.5*chi2 or -inf, status) for PHYSICAL parameters; status 1 = LSODA flag."""
    mod = model(pars[:6], tarr, xdata=x, v=with_extra_pars(v, pars), reference_cost=reference_cost)
    if isinstance(mod, str):
        return -np.inf, 1
    ll = -0.5 * np.sum(((y - mod) / yerr) ** 2.0)
    if not np.isfinite(ll):
        return -np.inf, 2
    return ll, 0


def lnprior(pars, lower, upper):
    pars = np.asarray(pars)
    if np.all(pars <= upper[:len(pars)]) and np.all(pars >= lower[:len(pars)]):
        return 0.0
    return -np.inf


def lnprob(pars, tarr, x, y, yerr, lower=SYNTH_PRIOR_LOWER, upper=SYNTH_PRIOR_UPPER, log_mask=0b111100,
           v=SYNTH, reference_cost=False):
    """(lnprob, status) in sampler coordinates; code/synthetic_datasets/mcmc_eqns.py:52-81."""
    if not np.isfinite(lnprior(pars, lower, upper)):
        return -np.inf, 3
    arr = np.array(pars, dtype=float)
    for i in range(arr.size):
        if (log_mask >> i) & 1:
            arr[i] = 10.0 ** arr[i]
    return lnlike_physical(arr, tarr, x, y, yerr, v, reference_cost=reference_cost)


# ---- multiprocessing helper for the cpu_baseline leg (mirrors synth_mcmc.py:178-185's Pool) ----
_POOL_STATE = {}


def _pool_init(tarr, x, y, yerr):
    _POOL_STATE.update(tarr=tarr, x=x, y=y, yerr=yerr)


def _pool_eval(p):
    """One evaluation with the reference's cost structure (what bench.py's cpu_baseline times)."""
    s = _POOL_STATE
    return lnprob(p, s["tarr"], s["x"], s["y"], s["yerr"], reference_cost=True)[0]

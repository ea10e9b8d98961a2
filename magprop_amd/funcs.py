"""Light-curve front end with the reference's signatures, evaluated by the gfx950 kernels.

Mirrors ``magnetar/funcs.py`` (``init_conds`` :17-29, ``model_lc`` :105-220) and
``code/synthetic_datasets/funcs.py`` (``model_lum`` :146-236).  Both physics variants are one kernel
parametrised by ``mp_model_cfg``.
"""
import numpy as np

from . import _capi, engine

Msol = 1.99e33


def init_conds(MdiscI, P):
    """magnetar/funcs.py:17-29: disc mass [g] and angular frequency [rad/s]."""
    return np.array([MdiscI * Msol, (2.0 * np.pi) / (1.0e-3 * P)])


def _curve(cfg, pars, xdata, GRBtype, device, dipeff, propeff, f_beam):
    pars = np.asarray(pars, dtype=np.float64)
    if pars.shape != (6,):
        raise ValueError("pars must hold the six parameters B, P, MdiscI, RdiscI, epsilon, delta")
    # the efficiencies and the beaming fraction travel as parameters 7-9 of the kernel (the 9-parameter form of
    # magnetar/mcmc_eqns.py:31-34), not in the handle's configuration: a scan over them reuses one cached handle
    pars = np.concatenate([pars, [float(dipeff), float(propeff), float(f_beam)]])
    with engine.use(cfg, GRBtype, device) as eng:
        status, out = eng.handle.model_lc(pars)
    if status != _capi.STATUS_OK:
        return "flag"  # magnetar/funcs.py:153-154
    if xdata is None:
        return out
    x = np.asarray(xdata, dtype=np.float64)
    t = eng.handle.tgrid
    if np.any(x < t[0]):
        raise ValueError("A value in x_new is below the interpolation range.")
    if np.any(x > t[-1]):
        raise ValueError("A value in x_new is above the interpolation range.")
    return np.interp(x, t, out[1])


def model_lc(pars, xdata=None, GRBtype=None, dipeff=0.05, propeff=0.4, f_beam=1.0, n=1.0, alpha=0.1, cs7=1.0,
             k=0.9, device=-1):
    """magnetar/funcs.py:105-220.  As in the reference, ``n``, ``alpha``, ``cs7`` and ``k`` reach only the luminosity
    stage: its ``odeint`` call passes five arguments (:150-151), so the ODE always runs with n=1, alpha=0.1, cs7=1,
    k=0.9.  In that stage they enter the radii, the fastness and the mass-flow split (:157-185), all of which feed
    only the accretion torque ``Nacc`` — and the reference's luminosity stage sets ``Nacc = 0`` at every grid point
    (``rot_param > 0.0`` always holds, :191-193), so ``Lprop`` is identically zero and ``Ltot = f_beam*dipeff*(-Ndip*omega)``
    does not depend on them.  They are therefore accepted and, exactly as in the reference, change nothing
    (reference outputs for several non-default values: tests/golden/golden_lib.npz ``lc_L_kw*``)."""
    engine.grid(GRBtype)  # ValueError for a bad GRBtype before anything else, as :138-141
    for name, v in (("n", n), ("alpha", alpha), ("cs7", cs7), ("k", k)):
        float(v)          # a non-numeric keyword fails here as it would in the reference's arithmetic
    return _curve(_capi.cfg_lib(), pars, xdata, GRBtype, device, dipeff, propeff, f_beam)


def model_lum(pars, xdata=None, n=10.0, alpha=0.1, cs7=1.0, k=0.9, dipeff=1.0, propeff=1.0, f_beam=1.0, device=-1):
    """code/synthetic_datasets/funcs.py:146-236 (fixed logspace(0,6,10001) grid, all keywords forwarded)."""
    cfg = _capi.cfg_synth(n_ode=n, n_lum=n, alpha=alpha, cs7=cs7, k=k)
    return _curve(cfg, pars, xdata, None, device, dipeff, propeff, f_beam)


def _rhs(cfg, y, t, B, MdiscI, RdiscI, epsilon, delta, device):
    y = np.asarray(y, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64)
    single = y.ndim == 1
    yy = np.atleast_2d(y)
    n = yy.shape[0]
    tt = np.broadcast_to(t, (n,))
    cols = [np.broadcast_to(np.asarray(v, dtype=np.float64), (n,)) for v in (B, 1.0, MdiscI, RdiscI, epsilon, delta)]
    pars = np.stack(cols, axis=1)                      # P (column 1) does not enter the right-hand side
    with engine.use(cfg, None, device) as eng:
        out = eng.handle.rhs_batch(pars, tt, yy)
    return out[0] if single else out


def odes(y, t, B, MdiscI, RdiscI, epsilon, delta, n=1.0, alpha=0.1, cs7=1.0, k=0.9, device=-1):
    """magnetar/funcs.py:33-101: time derivatives (dMdisc/dt, domega/dt) at state y = (Mdisc, omega) and time t,
    evaluated on the GPU by the device functions of the log-posterior kernels.  Usable as the right-hand-side callable
    of an ODE integrator exactly like the reference's (one launch per call), and batched: y (n, 2), t (n,) or scalar,
    parameters scalar or (n,)."""
    cfg = _capi.cfg_lib(n_ode=n, alpha=alpha, cs7=cs7, k=k)
    return _rhs(cfg, y, t, B, MdiscI, RdiscI, epsilon, delta, device)


def ODEs(y, t, B, MdiscI, RdiscI, epsilon, delta, n, alpha, cs7, k, device=-1):
    """code/synthetic_datasets/funcs.py:75-142 (I = 0.35 M R^2, factor 3 in the Alfven radius)."""
    cfg = _capi.cfg_synth(n_ode=n, alpha=alpha, cs7=cs7, k=k)
    return _rhs(cfg, y, t, B, MdiscI, RdiscI, epsilon, delta, device)

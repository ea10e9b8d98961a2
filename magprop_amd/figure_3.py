"""The two spin-down models the reference compares in ``code/figure_3.py``, evaluated by the gfx950 kernels.

``piroott(y, t, B, MdiscI, RdiscI, epsilon, delta)`` (:40-102) and ``bucciantini(...)`` (:105-165) are that script's
right-hand sides: the propeller model with ``I = 0.8 M R^2``, the factor 3 in the Alfven radius, ``n = 10``, ``k = 0.9``,
and either the packages' dipole torque ``-mu^2 omega^3 / (6 c^3)`` or the alternative
``-(2/3) (mu^2 omega^3 / c^3) (Rlc / Rm)^3`` (``mp_model_cfg.dipole_torque = 1``).  ``trajectory(...)`` integrates a
model over the script's grid, ``np.logspace(0, 6, 10001)`` (:20): what its two ``odeint`` calls (:194-201) return.
The script's plotting is out of scope.
"""
import numpy as np

from . import _capi, engine
from .funcs import _rhs, init_conds  # noqa: F401  (init_conds: code/figure_3.py:24-36, the same function)


def _cfg(torque):
    # code/figure_3.py:8-19: I = (4/5) M R^2, n = 10, alpha = 0.1, cs7 = 1, k = 0.9; (3 Mdisc / tvisc) in the Alfven radius (:64, :129)
    return _capi.cfg_lib(inertia_factor=0.8, rm_massflow_factor=3.0, n_ode=10.0, n_lum=10.0, dipole_torque=int(torque))


def piroott(y, t, B, MdiscI, RdiscI, epsilon, delta, device=-1):
    """code/figure_3.py:40-102: (dMdisc/dt, domega/dt) at y = (Mdisc, omega), time t.  Batched like ``magprop_amd.odes``."""
    return _rhs(_cfg(0), y, t, B, MdiscI, RdiscI, epsilon, delta, device)


def bucciantini(y, t, B, MdiscI, RdiscI, epsilon, delta, device=-1):
    """code/figure_3.py:105-165: the same with the dipole torque of Bucciantini et al. (2006)."""
    return _rhs(_cfg(1), y, t, B, MdiscI, RdiscI, epsilon, delta, device)


def trajectory(model, B, P, MdiscI, RdiscI, epsilon, delta, device=-1):
    """(tarr, Mdisc, omega) as the script's two integrations return them (code/figure_3.py:194-201: `model` from
    ``init_conds(MdiscI, P)`` over ``tarr`` with ``args=(B, MdiscI, RdiscI, epsilon, delta)``) for model "piroott" or "bucciantini"; the string "flag" where the reference's integrator
    gives up (break-up limit), as the packages' light-curve functions do."""
    if model not in ("piroott", "bucciantini"):
        raise ValueError("model must be 'piroott' or 'bucciantini'")
    pars = np.array([B, P, MdiscI, RdiscI, epsilon, delta], dtype=np.float64)
    with engine.use(_cfg(model == "bucciantini"), None, device) as eng:
        status, out, traj = eng.handle.model_lc(pars, want_traj=True)
    if status != _capi.STATUS_OK:
        return "flag"
    return out[0], traj[0], traj[1]

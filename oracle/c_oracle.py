"""ctypes loader for oracle/libmp_oracle.so — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It wraps the serial C restatement in oracle/mp_oracle.c (see that file's header
for the reference file:line each function follows).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmp_oracle.so")

STATUS_OK, STATUS_FLAG, STATUS_NONFINITE, STATUS_PRIOR = 0, 1, 2, 3


class Cfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "inertia_factor", "rm_massflow_factor", "n_ode", "n_lum", "alpha", "cs7", "k",
        "dipeff", "propeff", "f_beam", "nacc_lum_threshold")] + [
        ("lprop_gm_term", C.c_int32), ("dipole_torque", C.c_int32)]


def cfg_synth(**kw):
    """code/synthetic_datasets/funcs.py:17,105,146-147,206,222-223."""
    c = Cfg(0.35, 3.0, 10.0, 10.0, 0.1, 1.0, 0.9, 1.0, 1.0, 1.0, 0.27, 1, 0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def cfg_lib(**kw):
    """magnetar/funcs.py:12,33-34,64,105-106,150-151,193,206 (the ODE always runs with n=1)."""
    c = Cfg(0.8, 1.0, 1.0, 1.0, 0.1, 1.0, 0.9, 0.05, 0.4, 1.0, 0.0, 0, 0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def build(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(
            os.path.join(_HERE, "mp_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        L.mpo_trajectory.restype = C.c_int
        L.mpo_trajectory.argtypes = [C.POINTER(Cfg), dp, C.c_int, dp, C.c_int, C.c_int, dp, dp]
        L.mpo_trajectory_etd4rk.restype = C.c_int
        L.mpo_trajectory_etd4rk.argtypes = L.mpo_trajectory.argtypes
        L.mpo_model_lc.restype = C.c_int
        L.mpo_model_lc.argtypes = [C.POINTER(Cfg), dp, C.c_int, dp, C.c_int, C.c_int, dp, dp]
        L.mpo_rhs.restype = None
        L.mpo_rhs.argtypes = [C.POINTER(Cfg), dp, C.c_int, C.c_double, C.c_double, C.c_double, dp, dp]
        L.mpo_lnlike.restype = C.c_double
        L.mpo_lnlike.argtypes = [C.POINTER(Cfg), dp, C.c_int, dp, C.c_int, dp, dp, dp, C.c_int,
                                 C.POINTER(C.c_int)]
        L.mpo_lnprob_batch_mode.restype = None
        L.mpo_lnprob_batch_mode.argtypes = [C.POINTER(Cfg), dp, C.c_int, C.c_int, dp, dp, C.c_int, C.c_uint32,
                                            dp, C.c_int, dp, dp, dp, C.c_int, dp, ip, C.c_int, C.c_int, ip]
        L.mpo_lnprob_batch.restype = None
        L.mpo_lnprob_batch.argtypes = [C.POINTER(Cfg), dp, C.c_int, C.c_int, dp, dp, C.c_int, C.c_uint32,
                                       dp, C.c_int, dp, dp, dp, C.c_int, dp, ip]
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def trajectory(cfg, pars, tgrid, nsub=1, scheme="eam4"):
    """(status, Mdisc[n], omega[n]) for PHYSICAL parameters.  scheme "eam4" is the production scheme
    (exponential Adams-Moulton, what the HIP kernel implements); "etd4rk" the independent one-step
    cross-check (Krogstad exponential RK4 with an exponential Mdisc step)."""
    p, pp = _d(pars)
    t, tp = _d(tgrid)
    M = np.empty(t.size)
    W = np.empty(t.size)
    fn = lib().mpo_trajectory if scheme == "eam4" else lib().mpo_trajectory_etd4rk
    st = fn(C.byref(cfg), pp, p.size, tp, t.size, nsub,
                              M.ctypes.data_as(C.POINTER(C.c_double)), W.ctypes.data_as(C.POINTER(C.c_double)))
    return st, M, W


def rhs(cfg, pars, t, Mdisc, omega):
    """((dMdisc/dt, domega/dt), d(omega_dot)/d(omega)) at one state; PHYSICAL parameters."""
    p, pp = _d(pars)
    out = np.empty(2)
    lam = C.c_double(0.0)
    lib().mpo_rhs(C.byref(cfg), pp, p.size, float(t), float(Mdisc), float(omega),
                  out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(lam))
    return out, lam.value


def model_lc(cfg, pars, tgrid, nsub=1, want_traj=False):
    """(status, out[4][n]) like model_lc/model_lum(xdata=None); PHYSICAL parameters."""
    p, pp = _d(pars)
    t, tp = _d(tgrid)
    out = np.empty((4, t.size))
    traj = np.empty((2, t.size))
    st = lib().mpo_model_lc(C.byref(cfg), pp, p.size, tp, t.size, nsub,
                            out.ctypes.data_as(C.POINTER(C.c_double)),
                            traj.ctypes.data_as(C.POINTER(C.c_double)))
    return (st, out, traj) if want_traj else (st, out)


def lnlike(cfg, pars, tgrid, x, y, yerr):
    """(lnlike, status) for PHYSICAL parameters."""
    p, pp = _d(pars)
    t, tp = _d(tgrid)
    x, xp = _d(x)
    y, yp = _d(y)
    e, ep = _d(yerr)
    st = C.c_int(0)
    ll = lib().mpo_lnlike(C.byref(cfg), pp, p.size, tp, t.size, xp, yp, ep, x.size, C.byref(st))
    return ll, st.value


def lnprob_batch(cfg, pars, tgrid, x, y, yerr, lower=None, upper=None, log_mask=0, mode="fixed", spl=4, want_tiles=False):
    """(lnprob[n], status[n]) in sampler coordinates (box prior + un-logging per log_mask).  mode "fixed": every grid
    interval is a step (after the sub-stepped first 32 intervals); "adaptive": tiles of 64*spl steps over 1, 2, 4 or 8
    intervals, the product default (mp_oracle.c mpo_trajectory_mode).  want_tiles: also [n][3] = tile solves, tiles cut
    short or redone, steps kept."""
    pars = np.atleast_2d(np.ascontiguousarray(pars, dtype=np.float64))
    nw, nd = pars.shape
    t, tp = _d(tgrid)
    x, xp = _d(x)
    y, yp = _d(y)
    e, ep = _d(yerr)
    if lower is None:
        lo, lop = _d(np.zeros(1))
        hi, hip = _d(np.zeros(1))
        npr = 0
    else:
        lo, lop = _d(lower)
        hi, hip = _d(upper)
        npr = lo.size
    out = np.empty(nw)
    st = np.empty(nw, dtype=np.int32)
    tiles = np.zeros((nw, 3), dtype=np.int32)
    lib().mpo_lnprob_batch_mode(C.byref(cfg), pars.ctypes.data_as(C.POINTER(C.c_double)), nw, nd, lop, hip, npr,
                                C.c_uint32(log_mask), tp, t.size, xp, yp, ep, x.size,
                                out.ctypes.data_as(C.POINTER(C.c_double)), st.ctypes.data_as(C.POINTER(C.c_int32)),
                                {"fixed": 0, "adaptive": 1}[mode], int(spl), tiles.ctypes.data_as(C.POINTER(C.c_int32)))
    return (out, st, tiles) if want_tiles else (out, st)

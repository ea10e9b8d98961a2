"""ctypes binding of include/magprop_amd.h (libmagprop_amd.so: C ABI + gfx950 kernels).

This is the stub a magprop maintainer would add next to ``magnetar/funcs.py`` to call the HIP path
(see INTEGRATION.md).  There is no CPU fallback here: if the shared library is missing or no HIP
device is visible, importing/creating raises ``MagpropAmdError`` loudly.
"""
import ctypes as C
import sys
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MAGPROP_AMD_LIB") or os.path.join(_PKG, "libmagprop_amd.so")  # override: A/B builds

MP_OK, MP_EINVAL, MP_EHIP, MP_ERANGE, MP_ENODEV, MP_ESTATE = 0, -1, -2, -3, -4, -5
STATUS_OK, STATUS_FLAG, STATUS_NONFINITE, STATUS_PRIOR, STATUS_BADDATASET = 0, 1, 2, 3, 4
MAX_NDIM = 9
MAX_DATASETS = 64

EXPORTS = (
    "mp_abi_version", "mp_last_error", "mp_cfg_synth", "mp_cfg_lib", "mp_create", "mp_destroy",
    "mp_set_dataset", "mp_set_prior", "mp_lnprob_batch", "mp_lnprob_batch_dev", "mp_model_lc", "mp_rhs_batch",
    "mp_synchronize", "mp_device", "mp_stream", "mp_n_grid", "mp_last_mean_sweeps", "mp_last_mean_tiles",
    "mp_sampler_create", "mp_sampler_destroy", "mp_sampler_set_positions", "mp_sampler_run", "mp_sampler_set_whole_step", "mp_sampler_get_state",
    "mp_sampler_get_bad", "mp_sampler_n_slots", "mp_sampler_row_doubles", "mp_sampler_halfstep_shard",
    "mp_sampler_halfstep_apply", "mp_sampler_step_blocks", "mp_sampler_step_row_doubles", "mp_sampler_step_shard",
    "mp_sampler_step_apply", "mp_sampler_state_ptrs", "mp_sweep_tol", "mp_n_simd", "mp_last_sweeps", "mp_last_tiles", "mp_tile_log", "mp_last_tile_log",
    "mp_get_policy", "mp_create_multi", "mp_n_devices",
)
ABI_VERSION = 5
# order of mp_get_policy()'s vector (include/magprop_amd.h MP_POLICY_*)
POLICY_FIELDS = ("max_stride", "stride_tol", "sweep_tol", "early_hold_seconds", "k4_tol_factor", "coarse_tol_factor",
                 "coarse_max_sweeps", "fine_max_sweeps", "trouble_limit", "stop_factor", "forced_steps_per_lane", "experiments_build",
                 "light_tol", "cut_by_ratio", "abort_skip_ratio", "logpred_min_kind", "pre_early_end_factor")


class MagpropAmdError(RuntimeError):
    pass


class ModelCfg(C.Structure):
    """mp_model_cfg (include/magprop_amd.h)."""
    _fields_ = [(n, C.c_double) for n in (
        "inertia_factor", "rm_massflow_factor", "n_ode", "n_lum", "alpha", "cs7", "k",
        "dipeff", "propeff", "f_beam", "nacc_lum_threshold")] + [
        ("lprop_gm_term", C.c_int32), ("max_stride", C.c_int32), ("sweep_tol", C.c_double), ("stride_tol", C.c_double),
        ("dipole_torque", C.c_int32), ("reserved", C.c_int32)]


def build(force=False, verbose=False):
    """Compile the HIP library in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    src = os.path.join(_PKG, "csrc")
    args = ["make", "-C", src, "-s"] + (["-B"] if force else [])
    subprocess.check_call(args, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as the
    system one, different file name), and two HIP/HSA runtimes in one process do not work: whichever
    initialises second sees no GPU.  If torch is loaded first, the dynamic linker already resolves our
    NEEDED libamdhip64.so.7 to torch's copy.  For the other order (magprop_amd first, torch later) torch's
    copy is preloaded here, found without importing torch.  No torch, or MAGPROP_AMD_SYSTEM_HIP=1: the
    runtime under /opt/rocm is used."""
    import sys
    if "torch" in sys.modules or os.environ.get("MAGPROP_AMD_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        try:
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
        except OSError:
            pass  # fall back to the system runtime (fine as long as torch is not used in this process)


def lib():
    """Load libmagprop_amd.so (building it if hipcc is available and it is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        try:
            build()
        except Exception as exc:  # noqa: BLE001
            raise MagpropAmdError(
                f"{LIB_PATH} is missing and could not be built ({exc}); run `python -c 'import "
                "__graft_entry__ as g; g.build()'` on a machine with hipcc") from exc
    _share_hip_runtime_with_torch()
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as exc:
        raise MagpropAmdError(f"cannot load {LIB_PATH}: {exc}") from exc
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
    # the ABI check comes before any other symbol is touched: a stale build must fail with the rebuild hint, not with an
    # AttributeError on a symbol it does not have yet
    try:
        L.mp_abi_version.restype = C.c_int
        abi = L.mp_abi_version()
    except AttributeError:
        abi = None
    if abi != ABI_VERSION:
        raise MagpropAmdError(f"{LIB_PATH} has ABI version {abi}, this binding expects {ABI_VERSION}: "
                              "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
    L.mp_last_error.restype = C.c_char_p
    L.mp_cfg_synth.argtypes = [C.POINTER(ModelCfg)]
    L.mp_cfg_synth.restype = None
    L.mp_cfg_lib.argtypes = [C.POINTER(ModelCfg)]
    L.mp_cfg_lib.restype = None
    L.mp_create.restype = vp
    L.mp_create.argtypes = [C.POINTER(ModelCfg), dp, C.c_int, C.c_int]
    L.mp_create_multi.restype = vp
    L.mp_create_multi.argtypes = [C.POINTER(ModelCfg), dp, C.c_int, ip, C.c_int]
    L.mp_n_devices.argtypes = [vp]
    L.mp_n_devices.restype = C.c_int
    L.mp_destroy.argtypes = [vp]
    L.mp_set_dataset.argtypes = [vp, C.c_int, dp, dp, dp, C.c_int]
    L.mp_set_prior.argtypes = [vp, dp, dp, C.c_int, C.c_uint32]
    L.mp_lnprob_batch.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]   # (host pointers as integers: the hot entry)
    L.mp_lnprob_batch_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    L.mp_model_lc.argtypes = [vp, dp, C.c_int, dp, dp, ip]
    L.mp_rhs_batch.argtypes = [vp, dp, C.c_int, dp, dp, C.c_int, dp, dp]
    L.mp_synchronize.argtypes = [vp]
    L.mp_device.argtypes = [vp]
    L.mp_stream.argtypes = [vp]
    L.mp_stream.restype = vp
    L.mp_n_grid.argtypes = [vp]
    i64p = C.POINTER(C.c_int64)
    L.mp_sampler_create.restype = vp
    L.mp_sampler_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, ip, C.c_uint64, C.c_double, C.c_int]
    L.mp_sampler_destroy.argtypes = [vp]
    L.mp_sampler_set_positions.argtypes = [vp, dp]
    L.mp_sampler_run.argtypes = [vp, C.c_int, dp, dp]
    L.mp_sampler_set_whole_step.argtypes = [vp, C.c_int]
    L.mp_sampler_get_state.argtypes = [vp, dp, dp, i64p, i64p]
    L.mp_last_mean_sweeps.argtypes = [vp]
    L.mp_last_mean_sweeps.restype = C.c_double
    L.mp_last_mean_tiles.argtypes = [vp]
    L.mp_last_mean_tiles.restype = C.c_double
    L.mp_sweep_tol.argtypes = [vp]
    L.mp_sweep_tol.restype = C.c_double
    L.mp_n_simd.argtypes = [vp]
    L.mp_get_policy.argtypes = [vp, dp, C.c_int]
    L.mp_get_policy.restype = C.c_int
    L.mp_last_sweeps.argtypes = [vp, ip, C.c_int]
    L.mp_last_sweeps.restype = C.c_int
    L.mp_last_tiles.argtypes = [vp, ip, C.c_int]
    L.mp_last_tiles.restype = C.c_int
    L.mp_tile_log.argtypes = [vp, C.c_int]
    L.mp_tile_log.restype = C.c_int
    L.mp_last_tile_log.argtypes = [vp, C.c_int, ip, C.c_int]
    L.mp_last_tile_log.restype = C.c_int
    L.mp_sampler_get_bad.argtypes = [vp, C.c_int64, dp, C.c_int, i64p, i64p]
    L.mp_sampler_n_slots.argtypes = [vp]
    L.mp_sampler_row_doubles.argtypes = [vp]
    L.mp_sampler_halfstep_shard.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.mp_sampler_halfstep_apply.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.mp_sampler_state_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.mp_sampler_step_blocks.argtypes = [vp]
    L.mp_sampler_step_row_doubles.argtypes = [vp]
    L.mp_sampler_step_shard.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.mp_sampler_step_apply.argtypes = [vp, vp, vp, vp, vp]
    for name in ("mp_destroy", "mp_set_dataset", "mp_set_prior", "mp_lnprob_batch", "mp_lnprob_batch_dev",
                 "mp_model_lc", "mp_rhs_batch", "mp_synchronize", "mp_device", "mp_n_grid", "mp_sampler_destroy",
                 "mp_sampler_set_positions", "mp_sampler_run", "mp_sampler_set_whole_step", "mp_sampler_get_state", "mp_sampler_get_bad",
                 "mp_sampler_n_slots", "mp_sampler_row_doubles", "mp_sampler_halfstep_shard",
                 "mp_sampler_halfstep_apply", "mp_sampler_step_blocks", "mp_sampler_step_row_doubles",
                 "mp_sampler_step_shard", "mp_sampler_step_apply", "mp_sampler_state_ptrs", "mp_n_simd"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


def last_error():
    return lib().mp_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc == MP_OK:
        return
    msg = last_error()
    if rc == MP_ERANGE:
        raise ValueError(msg)  # scipy's interp1d raises ValueError there (magnetar/funcs.py:214-215)
    if rc == MP_EINVAL:
        raise ValueError(f"{what}: {msg}")
    raise MagpropAmdError(f"{what} failed (rc={rc}): {msg}")


SWEEP_TOL_DEFAULT, SWEEP_TOL_STRICT = 1.0e-7, 1.0e-11   # include/magprop_amd.h MP_SWEEP_TOL_*
DEFAULT_SWEEP_TOL = 0.0   # what cfg_synth()/cfg_lib() put into mp_model_cfg.sweep_tol (0 = the library default); the test
                          # suite sets SWEEP_TOL_STRICT here for its kernel-vs-serial-restatement comparisons
DEFAULT_MAX_STRIDE = 0    # likewise mp_model_cfg.max_stride (0 = the library default, adaptive up to 8 grid intervals per
                          # step); the strict test mode sets 1: every grid interval a step, the scheme of the serial restatement


def cfg_synth(**kw):
    c = ModelCfg()
    lib().mp_cfg_synth(C.byref(c))
    c.sweep_tol = DEFAULT_SWEEP_TOL
    c.max_stride = DEFAULT_MAX_STRIDE
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def cfg_lib(**kw):
    c = ModelCfg()
    lib().mp_cfg_lib(C.byref(c))
    c.sweep_tol = DEFAULT_SWEEP_TOL
    c.max_stride = DEFAULT_MAX_STRIDE
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def curve_steps_per_lane(n, n_simd):
    """Steps per lane of the curve kernel (mode B: light curves to HBM) a batch of n walkers runs on: the rule of
    magprop_amd/csrc/mp_device.h kernel_spl_curves, restated for labels (bench.py) and tests (tests/test_capi_cpu.py holds the
    two together)."""
    if n <= n_simd:
        return 4
    x2 = n / (7.0 * (n_simd // 4))
    r2 = 1.62 * ((1.0 if x2 <= 1.0 else 2.0) if x2 <= 2.0 else x2 + 0.2)
    return 4 if -(-n // n_simd) <= r2 else 2


def whole_step_fits(whole_step_blocks, n_simd):
    """mp_sampler_run evaluates a whole step per launch up to this many evaluations (3/2 x the walkers of all ensembles): the
    rule of magprop_amd/csrc/mp_device.h stretch_whole_step_fits, restated for labels (tests/test_capi_cpu.py holds the two
    together)."""
    return 8 * int(whole_step_blocks) <= 19 * int(n_simd)


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Handle:
    """Owns one mp_handle: a model configuration + time grid bound to one GPU."""

    def __init__(self, cfg, tgrid, device=-1):
        """device: a HIP device index (-1: the current one), or a sequence of indices for a multi-device handle
        (mp_create_multi: host-buffer batches are dealt out over the listed devices inside the library)."""
        self._L = lib()
        self.tgrid = np.ascontiguousarray(tgrid, dtype=np.float64)
        self.cfg = cfg
        if np.ndim(device) > 0:
            devs = np.ascontiguousarray(device, dtype=np.int32)
            self._h = self._L.mp_create_multi(C.byref(cfg), _dptr(self.tgrid), int(self.tgrid.size), _iptr(devs), int(devs.size))
        else:
            self._h = self._L.mp_create(C.byref(cfg), _dptr(self.tgrid), int(self.tgrid.size), int(device))
        if not self._h:
            raise MagpropAmdError("mp_create failed: " + last_error())

    @property
    def n_devices(self):
        return self._L.mp_n_devices(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.mp_destroy(self._h)
            self._h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():      # interpreter shutdown: the HIP runtime may already be gone; the OS reclaims the rest
            return
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    @property
    def device(self):
        return self._L.mp_device(self._h)

    def set_dataset(self, ds_id, x, y, yerr):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        yerr = np.ascontiguousarray(yerr, dtype=np.float64)
        if not (x.ndim == y.ndim == yerr.ndim == 1 and x.size == y.size == yerr.size):
            raise ValueError("x, y, yerr must be 1-D arrays of equal length")
        check(self._L.mp_set_dataset(self._h, int(ds_id), _dptr(x), _dptr(y), _dptr(yerr), int(x.size)),
              "mp_set_dataset")

    def set_prior(self, lower, upper, log_mask=0):
        if lower is None:
            check(self._L.mp_set_prior(self._h, None, None, 0, C.c_uint32(log_mask)), "mp_set_prior")
            return
        lo = np.ascontiguousarray(lower, dtype=np.float64)
        hi = np.ascontiguousarray(upper, dtype=np.float64)
        if lo.shape != hi.shape or lo.ndim != 1:
            raise ValueError("lower/upper must be 1-D arrays of equal length")
        check(self._L.mp_set_prior(self._h, _dptr(lo), _dptr(hi), int(lo.size), C.c_uint32(log_mask)),
              "mp_set_prior")

    def lnprob_batch(self, pars, ds_id=None, want_status=False, want_ltot=False):
        p = np.ascontiguousarray(pars, dtype=np.float64)
        if p.ndim != 2:
            raise ValueError("pars must be 2-D (n_walkers, ndim)")
        n, nd = p.shape
        out = np.empty(n, dtype=np.float64)
        st = np.empty(n, dtype=np.int32)
        lt = np.empty((n, self.tgrid.size), dtype=np.float64) if want_ltot else None
        ids = None
        if ds_id is not None:
            ids = (np.full(n, ds_id, dtype=np.int32) if np.ndim(ds_id) == 0
                   else np.ascontiguousarray(np.broadcast_to(np.asarray(ds_id, dtype=np.int32), (n,))))
        rc = self._L.mp_lnprob_batch(self._h, p.ctypes.data, ids.ctypes.data if ids is not None else None, n, nd,
                                     out.ctypes.data, st.ctypes.data, lt.ctypes.data if lt is not None else None)
        if rc:
            check(rc, "mp_lnprob_batch")
        res = (out,)
        if want_status:
            res += (st,)
        if want_ltot:
            res += (lt,)
        return res if len(res) > 1 else out

    def lnprob_batch_dev(self, d_pars, n, ndim, d_lnprob, d_ds_id=0, d_status=0, d_ltot=0, stream=0):
        """Device-pointer entry (ints from e.g. torch.Tensor.data_ptr()); asynchronous on `stream`
        (a hipStream_t as int; 0 = HIP's default stream, which is also torch's default stream)."""
        check(self._L.mp_lnprob_batch_dev(self._h, C.c_void_p(d_pars), C.c_void_p(d_ds_id or None), int(n),
                                          int(ndim), C.c_void_p(d_lnprob), C.c_void_p(d_status or None),
                                          C.c_void_p(d_ltot or None), C.c_void_p(stream or None)),
              "mp_lnprob_batch_dev")

    def model_lc(self, pars, want_traj=False):
        p = np.ascontiguousarray(pars, dtype=np.float64).ravel()
        out = np.empty((4, self.tgrid.size), dtype=np.float64)
        traj = np.empty((2, self.tgrid.size), dtype=np.float64)
        st = C.c_int32(0)
        check(self._L.mp_model_lc(self._h, _dptr(p), int(p.size), _dptr(out), _dptr(traj), C.byref(st)),
              "mp_model_lc")
        return (st.value, out, traj) if want_traj else (st.value, out)

    def rhs_batch(self, pars, t, y, want_lam=False):
        """(dMdisc/dt, domega/dt) at n states: pars (n, ndim) physical, t (n,), y (n, 2) = (Mdisc, omega)."""
        p = np.ascontiguousarray(pars, dtype=np.float64)
        tt = np.ascontiguousarray(t, dtype=np.float64).ravel()
        yy = np.ascontiguousarray(y, dtype=np.float64)
        if p.ndim != 2 or yy.shape != (p.shape[0], 2) or tt.shape != (p.shape[0],):
            raise ValueError("pars (n, ndim), t (n,), y (n, 2) expected")
        n = p.shape[0]
        out = np.empty((n, 2), dtype=np.float64)
        lam = np.empty(n, dtype=np.float64)
        check(self._L.mp_rhs_batch(self._h, _dptr(p), int(p.shape[1]), _dptr(tt), _dptr(yy), n, _dptr(out), _dptr(lam)),
              "mp_rhs_batch")
        return (out, lam) if want_lam else out

    def synchronize(self):
        check(self._L.mp_synchronize(self._h), "mp_synchronize")

    @property
    def last_mean_sweeps(self):
        return self._L.mp_last_mean_sweeps(self._h)

    @property
    def last_mean_tiles(self):
        """Tiles solved per walker in the most recent host-buffer batch (kept or redone)."""
        return self._L.mp_last_mean_tiles(self._h)

    def last_sweeps(self, n):
        """Total Newton sweeps (over all tiles) of each of the first n walkers of the most recent host-buffer batch."""
        out = np.zeros(n, dtype=np.int32)
        m = self._L.mp_last_sweeps(self._h, _iptr(out), int(n))
        return out[:max(m, 0)]

    def last_tiles(self, n):
        """Tiles solved (kept or redone) by each of the first n walkers of the most recent host-buffer batch."""
        out = np.zeros(n, dtype=np.int32)
        m = self._L.mp_last_tiles(self._h, _iptr(out), int(n))
        return out[:max(m, 0)]

    def tile_log(self, enable=True):
        """Record, in every later host-buffer batch, what each walker's tiles were (diagnostics)."""
        check(self._L.mp_tile_log(self._h, int(bool(enable))), "mp_tile_log")

    def last_tile_log(self, walker):
        """[(kind, sweeps, lanes kept, why), ...] of `walker` in the most recent host-buffer batch (tile_log() on): kind 0 =
        1/8-interval sub-steps, 1 .. 4 = steps over 1 / 2 / 4 / 8 grid intervals; 0 lanes kept = the tile was redone; why =
        bits: 1 a branch of the right-hand side changed inside the tile, 2 / 4 / 8 / 32 the smoothness indicator exceeded
        stride_tol / its 64th / its 2048th / its 65536th somewhere, 16 lanes had not converged when the sweeps were
        stopped, 64 the tile was given up after its second sweep."""
        buf = np.zeros(96, dtype=np.int32)
        m = self._L.mp_last_tile_log(self._h, int(walker), _iptr(buf), 96)
        return [(int(w) & 15, (int(w) >> 4) & 0xFFF, (int(w) >> 16) & 0xFF, (int(w) >> 24) & 0xFF) for w in buf[:max(m, 0)]]

    @property
    def sweep_tol(self):
        return self._L.mp_sweep_tol(self._h)

    @property
    def policy(self):
        """The solver settings in force (mp_get_policy): a dict keyed by POLICY_FIELDS.  experiments_build = 1.0 marks the
        developer build that honours MAGPROP_AMD_* environment overrides; the shipped library reports 0."""
        buf = np.zeros(len(POLICY_FIELDS))
        m = self._L.mp_get_policy(self._h, _dptr(buf), int(buf.size))
        if m < 0:
            check(m, "mp_get_policy")
        return dict(zip(POLICY_FIELDS[:m], (float(v) for v in buf[:m])))

    @property
    def n_simd(self):
        """SIMDs of the device: batches up to n_simd walkers run the 4-steps-per-lane kernel, larger ones the
        2-steps-per-lane kernel (mp_device.h)."""
        return self._L.mp_n_simd(self._h)

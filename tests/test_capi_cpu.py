"""CPU-side checks of the C ABI library and the host logic (no GPU compute)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from magprop_amd import _capi, engine, mcmc_eqns, synth


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "magprop_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mp_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    L = _capi.lib()
    names = _declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/magprop_amd.h but not exported"
    assert set(names) == set(_capi.EXPORTS)
    assert L.mp_abi_version() == _capi.ABI_VERSION == 5


def test_python_mirror_follows_the_header_constants():
    """The constants and the order of mp_get_policy()'s vector in magprop_amd/_capi.py are those of include/magprop_amd.h."""
    hdr = open(os.path.join(ROOT, "include", "magprop_amd.h")).read()

    def define(name):
        return float(re.search(r"#define\s+%s\s+([0-9.eE+-]+)" % name, hdr).group(1))

    assert define("MP_SWEEP_TOL_DEFAULT") == _capi.SWEEP_TOL_DEFAULT == 1.0e-7
    assert define("MP_SWEEP_TOL_STRICT") == _capi.SWEEP_TOL_STRICT == 1.0e-11
    assert define("MP_STOP_FACTOR") == 0.01
    body = re.search(r"enum\s*\{\s*MP_POLICY_MAX_STRIDE.*?MP_POLICY_COUNT", hdr, flags=re.S).group(0)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"MP_POLICY_([A-Z0-9_]+)", body)[:-1]
    alias = {"FORCED_STEPS_PER_LANE": "forced_steps_per_lane", "EXPERIMENTS": "experiments_build"}
    assert [alias.get(n, n.lower()) for n in names] == list(_capi.POLICY_FIELDS)
    # the compile-time policy constants: one definition (mp_device.h), mirrored by name in the serial restatement
    dev = open(os.path.join(ROOT, "magprop_amd", "csrc", "mp_device.h")).read()
    orc = open(os.path.join(ROOT, "oracle", "mp_oracle.c")).read()
    pre = float(re.search(r"#define\s+MP_PRE_EARLY_END_FACTOR\s+([0-9.eE+-]+)", dev).group(1))
    assert pre == float(re.search(r"#define\s+MPO_PRE_EARLY_END_FACTOR\s+([0-9.eE+-]+)", orc).group(1)) == 65536.0 * 100.0
    assert "6553600.0 *" not in orc                           # (the literal is gone from the oracle's code)


def test_curve_kernel_rule_mirror(tmp_path):
    """_capi.curve_steps_per_lane (labels of bench.py, expectations of the GPU tests) IS kernel_spl_curves of
    magprop_amd/csrc/mp_device.h: the header's inline function compiled for the host against the Python restatement, every batch
    size up to 20 000 on two device sizes."""
    import subprocess
    src = tmp_path / "rule.cpp"
    src.write_text('#include <cstdio>\n#include <initializer_list>\n#include "magprop_amd/csrc/mp_device.h"\n'
                   'int main() { mp::DevShared sh{}; for (int simd : {1024, 416}) { sh.n_simd = simd; '
                   'for (int n = 1; n <= 20000; ++n) std::printf("%d", mp::kernel_spl_curves(sh, n)); std::printf("\\n"); '
                   'for (int n = 1; n <= 20000; ++n) std::printf("%d", (int)mp::stretch_whole_step_fits(sh, n)); std::printf("\\n"); } }\n')
    exe = tmp_path / "rule"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", ROOT, str(src), "-o", str(exe)], check=True)
    rows = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    for k, simd in enumerate((1024, 416)):
        assert rows[2 * k] == "".join(str(_capi.curve_steps_per_lane(n, simd)) for n in range(1, 20001)), simd
        # (likewise the size up to which the sampler evaluates a whole step per launch)
        assert rows[2 * k + 1] == "".join(str(int(_capi.whole_step_fits(n, simd))) for n in range(1, 20001)), simd
    assert _capi.whole_step_fits(3 * 810, 1024) and not _capi.whole_step_fits(3 * 811, 1024)
    # the measured points of profiles/r05_ab_curve_spl.log (an MI355X: 1 024 SIMDs)
    assert [_capi.curve_steps_per_lane(n, 1024) for n in (1024, 1536, 2048, 3072, 3584, 4096, 5120, 8192)] == [4, 2, 4, 4, 2, 4, 2, 2]


def test_cfg_struct_layout_and_presets():
    assert ctypes.sizeof(_capi.ModelCfg) == 11 * 8 + 2 * 4 + 2 * 8 + 2 * 4
    c0 = _capi.cfg_synth()
    assert c0.sweep_tol == 0.0 and c0.stride_tol == 0.0 and c0.max_stride == 0   # 0 = the library defaults (MP_*_DEFAULT)
    s, l = _capi.cfg_synth(), _capi.cfg_lib()
    assert (s.inertia_factor, s.rm_massflow_factor, s.n_ode, s.n_lum, s.nacc_lum_threshold, s.lprop_gm_term) == \
        (0.35, 3.0, 10.0, 10.0, 0.27, 1)
    assert (l.inertia_factor, l.rm_massflow_factor, l.n_ode, l.dipeff, l.propeff, l.nacc_lum_threshold,
            l.lprop_gm_term) == (0.8, 1.0, 1.0, 0.05, 0.4, 0.0, 0)


def test_no_silent_cpu_fallback():
    """Without a GPU the product path must fail loudly (there is no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(_capi.MagpropAmdError, match="no HIP device"):
        _capi.Handle(_capi.cfg_synth(), engine.grid(None))


def test_create_argument_validation():
    L = _capi.lib()
    t = np.array([1.0, 1.0, 2.0])
    assert not L.mp_create(ctypes.byref(_capi.cfg_synth()), t.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 3, 0)
    assert "strictly increasing" in _capi.last_error()
    assert not L.mp_create(None, None, 0, 0)


def test_only_log_spaced_grids_are_accepted():
    """The integrator is built on np.logspace grids (the only ones the reference uses); anything else is refused."""
    L = _capi.lib()
    dp = ctypes.POINTER(ctypes.c_double)
    t = np.linspace(1.0, 100.0, 200)
    assert not L.mp_create(ctypes.byref(_capi.cfg_synth()), t.ctypes.data_as(dp), t.size, 0)
    assert "log-spaced" in _capi.last_error()
    t = np.logspace(0.0, 6.0, 10001)
    t[5000] *= 1.0 + 1e-6
    assert not L.mp_create(ctypes.byref(_capi.cfg_synth()), t.ctypes.data_as(dp), t.size, 0)
    assert "log-spaced" in _capi.last_error()


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ in any way."""
    pkg = os.path.join(ROOT, "magprop_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libmp_oracle" not in src, f
                assert "scipy.integrate" not in src and "odeint(" not in src, f


def test_grid_selection_matches_reference():
    assert np.array_equal(engine.grid(None), np.logspace(0.0, 6.0, num=10001, base=10.0))
    assert np.array_equal(engine.grid("L"), engine.grid(None))
    assert np.array_equal(engine.grid("S"), np.logspace(-3.0, 6.0, num=10001, base=10.0))
    with pytest.raises(ValueError, match="valid value for GRBtype"):
        engine.grid("X")


def test_synth_lnprior(gsynth):
    assert np.array_equal(synth.PRIOR_LOWER, gsynth["prior_lower"])
    assert np.array_equal(synth.PRIOR_UPPER, gsynth["prior_upper"])
    assert synth.lnprior([1, 5, -3, 2, -1, 0]) == 0.0
    assert synth.lnprior([1, 5, -3, 2, -1, 3.5]) == -np.inf
    assert synth.lnprior(synth.PRIOR_LOWER) == 0.0 and synth.lnprior(synth.PRIOR_UPPER) == 0.0  # inclusive
    out = synth.lnprior(np.array([[1, 5, -3, 2, -1, 0], [11, 5, -3, 2, -1, 0]]))
    assert out.tolist() == [0.0, -np.inf]


def test_lib_lnprior_matches_reference(glib):
    """magnetar/mcmc_eqns.py:40-84 incl. the 7-parameter special case, on golden cases."""
    assert np.array_equal(mcmc_eqns.DEFAULT_LIMITS_LOWER, glib["limits_lower"])
    assert np.array_equal(mcmc_eqns.DEFAULT_LIMITS_UPPER, glib["limits_upper"])
    for row, ref in zip(glib["lnprior_pars"], glib["lnprior"]):
        p = row[~np.isnan(row)]
        assert mcmc_eqns.lnprior(p) == ref


def test_lib_lnprior_custom_limits(tmp_path):
    p = tmp_path / "lims.csv"
    p.write_text("pars,lower,upper\nB,0,1\nP,0,1\na,0,1\nb,0,1\nc,0,1\nd,0,1\ne,0,1\nf,0,1\ng,5,6\n")
    assert mcmc_eqns.lnprior([0.5] * 6, custom_lims=str(p)) == 0.0
    assert mcmc_eqns.lnprior([0.5] * 5 + [2.0], custom_lims=str(p)) == -np.inf
    assert mcmc_eqns.lnprior([0.5] * 6 + [5.5], custom_lims=str(p)) == 0.0      # 7th -> last row
    assert mcmc_eqns.lnprior([0.5] * 6 + [0.5], custom_lims=str(p)) == -np.inf
    with pytest.raises(ValueError, match="valid file path"):
        mcmc_eqns.lnprior([0.5] * 6, custom_lims=str(tmp_path / "missing.csv"))


def test_fit_stats_match_reference(glib):
    """magnetar/fit_stats.py on the reference's own noisy_gaussian fixture (tests/test_funcs.py:153-182)."""
    import magprop_amd as mpa
    yd, ye, ym = glib["fit_ydata"], glib["fit_yerr"], glib["fit_ymod"]
    assert mpa.redchisq(yd, ym, sd=ye) == float(glib["fit_redchisq_sd"])
    assert mpa.redchisq(yd, ym, deg=6, sd=ye) == float(glib["fit_redchisq_sd_deg6"])
    assert mpa.redchisq(yd, ym) == float(glib["fit_redchisq_plain"])
    assert mpa.aicc(yd, ym, ye, 2) == float(glib["fit_aicc_2"])
    assert mpa.aicc(yd, ym, ye, 6) == float(glib["fit_aicc_6"])
    with pytest.raises(ValueError, match="same length"):
        mpa.aicc(yd, ym[:-1], ye, 2)


def test_model_lc_keyword_goldens_are_keyword_independent():
    """model_lc(alpha, cs7, k, n != defaults) in the REFERENCE (tests/golden/golden_libkw.npz, made by make_golden.py
    --only libkw): those keywords reach only its luminosity stage, whose accretion torque is identically zero
    (magnetar/funcs.py:150-151,191-193), so the reference's curves equal its default-keyword curves.  This is the fact
    magprop_amd.model_lc relies on when it accepts them without effect (GPU comparison: tests/test_gpu_parity.py)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_libkw.npz"))
    lib = np.load(os.path.join(ROOT, "tests", "golden", "golden_lib.npz"))
    for i in range(int(g["n_cases"])):
        names = [str(n) for n in g[f"kw{i}_names"]]
        if not set(names) & {"dipeff", "propeff", "f_beam"}:
            for kind in ("L", "S"):
                assert np.array_equal(g[f"kw{i}_{kind}_humped"], lib[f"lc_{kind}"]), (i, kind)
        assert np.all(g[f"kw{i}_L_humped"][2] == 0.0)           # Lprop == 0 whatever the keywords


def test_engine_cache_is_bounded(monkeypatch):
    """A keyword sweep (one model configuration per value) must not pile up GPU handles: least recently used out."""
    made, closed = [], []

    class FakeHandle:
        def __init__(self, cfg, tgrid, device):
            made.append(self)

        def close(self):
            closed.append(self)

    monkeypatch.setattr(_capi, "Handle", FakeHandle)
    engine.clear()
    first = engine.engine(_capi.cfg_synth(k=0.5))
    for i in range(engine.MAX_ENGINES + 3):
        engine.engine(_capi.cfg_synth(alpha=0.01 * (i + 1)))
        assert engine.engine(_capi.cfg_synth(k=0.5)) is first     # kept alive by use
    assert len(made) == engine.MAX_ENGINES + 4 and len(closed) == 4 and first.handle not in closed
    engine.clear()
    assert len(closed) == len(made)


def test_engine_in_use_is_never_evicted(monkeypatch):
    """An engine a caller is inside (engine.use) survives any number of look-ups for other configurations, from other
    threads too; once released it is evicted like any other."""
    import threading
    closed = []

    class FakeHandle:
        _h = 1

        def __init__(self, cfg, tgrid, device):
            pass

        def close(self):
            closed.append(self)
            self._h = None

    monkeypatch.setattr(_capi, "Handle", FakeHandle)
    engine.clear()
    with engine.use(_capi.cfg_synth(k=0.5)) as held:
        def sweep():
            for i in range(engine.MAX_ENGINES + 5):
                engine.engine(_capi.cfg_synth(alpha=0.01 * (i + 1)))
        t = threading.Thread(target=sweep)
        t.start()
        t.join()
        assert held.handle not in closed and held.pins == 1 and len(closed) >= 5
    assert held.pins == 0
    for i in range(engine.MAX_ENGINES + 1):
        engine.engine(_capi.cfg_synth(cs7=0.1 * (i + 1)))
    assert held.handle in closed
    engine.clear()


def test_only_the_checkers_touch_the_oracle():
    """oracle/ is test infrastructure: besides tests/, only bench.py's CPU-baseline legs (cpu_baseline, config1_cpu) and __graft_entry__.smoke() (and
    build(), which compiles it) may refer to it — no tool, no product module."""
    import re
    allowed = {"bench.py": "def cpu_baseline", "__graft_entry__.py": None}
    for dirpath, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out", "tests", "oracle", "__pycache__", ".pytest_cache")]
        for f in files:
            if not f.endswith((".py", ".sh", ".cpp", ".hip", ".hpp", ".h")):
                continue
            rel = os.path.relpath(os.path.join(dirpath, f), ROOT)
            src = open(os.path.join(dirpath, f)).read()
            # imports, links or executions (comments that cite oracle/mp_oracle.c as the serial restatement are fine)
            uses = re.search(r"^\s*(from|import)\s+oracle|libmp_oracle|#include\s+\"[^\"]*oracle|python[^\n]*oracle/", src, re.M) is not None
            if rel in allowed:
                if rel == "bench.py":      # every import of the oracle sits inside the CPU legs cpu_baseline() / config1_cpu()
                    body = src[src.index("def cpu_baseline"):src.index("def free_port")]
                    assert src.count("from oracle") == body.count("from oracle") == 3
                continue
            assert not uses, rel

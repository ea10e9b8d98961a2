"""Where a walker's time goes, section by section (GPU box; developer tool).
    make -C magprop_amd/csrc phase-profile && MAGPROP_AMD_LIB=$PWD/magprop_amd/libmagprop_amd_phase.so python tools/phase_profile.py
The phase-profile build reads the shader clock between the sections of every tile and leaves the sums in the walker's
tile-log row (mp_eval.hpp MP_PHASE_PROFILE)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magprop_amd import LogProb, _capi

NAMES = ["walker setup (prior, constants, initial state, observations into registers)", "history lookup", "Mdisc step + scan + disc quantities",
         "predictor", "sweep: entry (wild-guess check), loop control", "acceptance policy", "image to LDS", "observations / curves",
         "tile control", "luminosity at the observations + chi^2", "step times + fallback rate", "failure detection", "carry to the next tile",
         "sweep: omega_dot (+ Jacobian)", "sweep: flags, neighbour fetch, node values", "sweep: phi functions + weights",
         "sweep: increment, scan, update", "sweep: convergence tests, early give-up"]
NPH = len(NAMES)
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
lo, hi = gs["prior_lower"], gs["prior_upper"]
rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sets = {"near truth": np.array([1.0, 5.0, -3.0, 2.0, -1.0, 0.0]) + 1e-4 * rng.standard_normal((n, 6)),
        "prior-wide": lo + (hi - lo) * rng.random((n, 6))}
lp = LogProb(gs["Humped_x"], gs["Humped_y"], gs["Humped_yerr"])
lp.handle.tile_log(True)
for name, P in sets.items():
    out, st = lp.handle.lnprob_batch(P, want_status=True)
    tiles, sweeps = lp.handle.last_tiles(n), lp.handle.last_sweeps(n)
    acc = np.zeros((n, NPH))
    buf = np.zeros(96, dtype=np.int32)
    for i in range(n):
        lp.handle._L.mp_last_tile_log(lp.handle._h, i, _capi._iptr(buf), 96)
        acc[i] = buf[:NPH]
    ok = st == 0
    tot = acc[ok].sum(axis=1)
    print(f"{name}: {ok.sum()} walkers, {tiles[ok].mean():.1f} tiles, {sweeps[ok].mean():.1f} sweeps per walker; clock ticks per walker mean {tot.mean():.0f} max {tot.max():.0f}")
    for k in range(NPH):
        print(f"   {NAMES[k]:80s} {100 * acc[ok, k].sum() / tot.sum():5.1f} %   per tile {acc[ok, k].sum() / tiles[ok].sum():8.1f} ticks")
    sw = acc[ok][:, [4, 13, 14, 15, 16, 17]].sum()
    print(f"   all of the sweeps: {100 * sw / tot.sum():.1f} %, per sweep {sw / sweeps[ok].sum():.1f} ticks")

#!/usr/bin/env python3
"""Developer diagnostic (GPU box): the Newton sweeps of every tile of a few walkers, sweep by sweep.
    make -C magprop_amd/csrc corr-trace && MAGPROP_AMD_LIB=$PWD/magprop_amd/libmagprop_amd_corr.so python tools/corr_trace.py [n]
Per sweep: F (full: omega_dot, Jacobian, phi functions, weights) or L (light: omega_dot only), in lower case when the tile
ended with it on the contraction estimate (no verification pass); log10 of the largest relative correction over the
wavefront, lanes still pending.  Then, over a near-truth and a prior-wide batch, the first two corrections per tile kind."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magprop_amd import LogProb, _capi
gs = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
lo, hi = gs["prior_lower"], gs["prior_upper"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(1)
near = np.array([1.0, 5.0, -3.0, 2.0, -1.0, 0.0]) + 1e-4 * rng.standard_normal((n, 6))
wide = lo + (hi - lo) * rng.random((n, 6))
lp = LogProb(gs["Humped_x"], gs["Humped_y"], gs["Humped_yerr"])
lp.handle.tile_log(True)
buf = np.zeros(96, dtype=np.int32)


def words(i):
    m = lp.handle._L.mp_last_tile_log(lp.handle._h, int(i), _capi._iptr(buf), 96)
    return [int(w) for w in buf[:max(m, 0)] if w != -1]


def show(i):
    tiles = {}
    for w in words(i):
        tiles.setdefault(w & 0xFF, []).append(("P1248"[(w >> 8) & 7], ("F", "L", "f", "l")[(w >> 11) & 3], (w >> 13) & 0x7F, -0.1 * ((w >> 20) & 0xFF)))
    return " | ".join(f"{v[0][0]}: " + " ".join(f"{m}{q:.1f}/{p}" for _, m, p, q in v) for _, v in sorted(tiles.items()))


for name, X in (("near truth", near), ("prior-wide", wide)):
    out, st = lp.handle.lnprob_batch(X, want_status=True)
    sw = lp.handle.last_sweeps(n)
    print(f"== {name}: {n} walkers")
    order = np.argsort(sw)[::-1]
    for i in ([0, 1] if name == "near truth" else list(order[:4]) + list(order[n // 2:n // 2 + 2])):
        print(f"walker {i} status {st[i]} sweeps {sw[i]}: {show(i)}")
    modes = np.zeros(4)
    first_corr, second_corr = {}, {}
    for i in range(min(n, 512)):
        if st[i] != 0:
            continue
        tiles = {}
        for w in words(i):
            tiles.setdefault(w & 0xFF, []).append(w)
            modes[(w >> 11) & 3] += 1
        for v in tiles.values():
            k = "P1248"[(v[0] >> 8) & 7]
            first_corr.setdefault(k, []).append(-0.1 * ((v[0] >> 20) & 0xFF))
            if len(v) > 1:
                second_corr.setdefault(k, []).append(-0.1 * ((v[1] >> 20) & 0xFF))
    print("   sweeps by mode: full %d, light %d; of these the last of a tile that ended on the contraction estimate "
          "(lower case above): full %d, light %d" % (modes[0] + modes[2], modes[1] + modes[3], modes[2], modes[3]))
    for k in "P1248":
        if k in first_corr:
            a, b = np.array(first_corr[k]), np.array(second_corr.get(k, [0.0]))
            print(f"   kind {k}: {len(a)} tiles; log10 correction of sweep 1: median {np.median(a):.1f}, 10 % above {np.percentile(a, 90):.1f}, below -2: {np.mean(a < -2):.2f}, "
                  f"below -3: {np.mean(a < -3):.2f}; of sweep 2: median {np.median(b):.1f}, 90th percentile {np.percentile(b, 90):.1f}")

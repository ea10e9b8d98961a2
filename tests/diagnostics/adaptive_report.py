"""Report (GPU box): the product's default stride-adaptive solver and the fixed-step mode against the reference's
tight-integrator values and the serial C restatement, over every golden point; tiles per walker and sweeps per tile.
    python tests/diagnostics/adaptive_report.py [--quick]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import TYPES  # noqa: E402


def main():
    from magprop_amd import LogProb, _capi
    from oracle import c_oracle as co
    G = os.path.join(ROOT, "tests", "golden")
    gs, gf, gf2 = (np.load(os.path.join(G, f)) for f in ("golden_synth.npz", "golden_flagscan.npz", "golden_flagscan2.npz"))
    tarr = np.logspace(0.0, 6.0, 10001)
    lo, hi = gs["prior_lower"], gs["prior_upper"]
    quick = "--quick" in sys.argv
    sets = [(n + "_cloud", gs[n + "_pars"], gs[n + "_lnprob_tight"], gs[n + "_status"], k) for k, n in enumerate(TYPES)]
    sets.append(("flagscan", gf["pars"], gf["lnprob_tight"], gf["status"], 0))
    for d, n in enumerate(TYPES[1:]):
        m = gf2["ds"] == d
        sets.append(("flagscan2_" + n, gf2["pars"][m], gf2["lnprob_tight"][m], gf2["status"][m], d + 1))
    data = [(gs[n + "_x"], gs[n + "_y"], gs[n + "_yerr"]) for n in TYPES]
    rep = {}
    for label, kw in (("adaptive_default", {}), ("fixed_strict", {"sweep_tol": _capi.SWEEP_TOL_STRICT, "max_stride": 1}),
                      ("fixed_default_tol", {"max_stride": 1})):
        lp = LogProb(*data[0], **kw)
        for s in data[1:]:
            lp.add_dataset(*s)
        rr, mism, tiles, sweeps, vs_c = [], 0, [], [], []
        for name, P, tight, rst, ds in sets:
            for batch in ((len(P),) if quick else (len(P), 1500)):     # 4-steps-per-lane kernel; split so both kernels run
                out = np.empty(len(P)); st = np.empty(len(P), dtype=np.int32)
                for a in range(0, len(P), batch):
                    o, s_ = lp.handle.lnprob_batch(P[a:a + batch], ds_id=ds, want_status=True)
                    out[a:a + batch], st[a:a + batch] = o, s_
                    tiles.append(lp.handle.last_mean_tiles); sweeps.append(lp.handle.last_mean_sweeps)
                ok = (rst == 0) & np.isfinite(tight) & (st == 0)
                mism += int(np.sum(st != rst))
                rr.append(np.abs(out[ok] - tight[ok]) / (1e-7 + 1e-7 * np.abs(tight[ok])))
            if label == "fixed_strict" and not quick:
                ref, cst = co.lnprob_batch(co.cfg_synth(), P, tarr, *data[ds], lo, hi, 0b111100)
                both = (cst == 0) & (st == 0)
                vs_c.append(np.abs(out[both] - ref[both]) / np.abs(ref[both]))
        r = np.concatenate(rr)
        rep[label] = {"max_over_tight_bound": float(r.max()), "p999": float(np.percentile(r, 99.9)), "p99": float(np.percentile(r, 99)),
                      "median": float(np.median(r)), "status_mismatches": mism, "tiles_per_walker": float(np.mean(tiles)),
                      "sweeps_per_tile": float(np.mean(sweeps))}
        if vs_c:
            v = np.concatenate(vs_c)
            rep[label]["vs_c_oracle_fixed_max_rel"] = float(v.max())
            rep[label]["vs_c_oracle_fixed_p999_rel"] = float(np.percentile(v, 99.9))
        print(label, json.dumps(rep[label]), flush=True)
    # timing of the three modes at the headline size
    import torch
    rng = np.random.default_rng(1)
    P = np.array([1.0, 5.0, -3.0, 2.0, -1.0, 0.0]) + 1e-4 * rng.standard_normal((1024, 6))
    wide = lo + (hi - lo) * rng.random((1024, 6))
    for label, kw in (("adaptive_default", {}), ("max_stride_2", {"max_stride": 2}), ("fixed", {"max_stride": 1})):
        lp = LogProb(*data[0], **kw)
        for nm, X in (("near_truth", P), ("prior_wide", wide)):
            dP = torch.from_numpy(X).cuda()
            out = torch.empty(1024, dtype=torch.float64, device="cuda")
            for _ in range(20):
                lp.lnprob_device(dP, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                lp.lnprob_device(dP, out=out)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 200
            lp(X)
            print(f"{label} {nm}: {1e3 * dt:.4f} ms per 1024 walkers = {1024 / dt / 1e6:.2f} M evals/s; tiles/walker "
                  f"{lp.handle.last_mean_tiles:.2f}, sweeps/tile {lp.handle.last_mean_sweeps:.2f}", flush=True)


if __name__ == "__main__":
    main()

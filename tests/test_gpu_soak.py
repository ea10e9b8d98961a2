"""Soak parity: every kernel variant against the serial C oracle on thousands of walkers spread over the whole prior
box, its faces and the neighbourhood of the four truths (flag agreement as a confusion matrix)."""
import json
import multiprocessing as mp
import os

import numpy as np
import pytest

from conftest import TRUTHS, TYPES

pytestmark = pytest.mark.gpu
LOG_MASK = 0b111100
N_SOAK = int(os.environ.get("MAGPROP_SOAK_N", "32768"))
STRICT = 1.0e-11   # MP_SWEEP_TOL_STRICT: the kernels against the serial restatement of the scheme
# Two sets of walkers: seed 20261003 is the one every constant of the stride policy was tuned on in round 3 (kept: a
# regression guard); the second seed was drawn after the round-4 constants were frozen (MAGPROP_SOAK_SEED overrides it, so
# the points can be changed without touching the file).
TUNING_SEED = 20261003
FRESH_SEED = int(os.environ.get("MAGPROP_SOAK_SEED", "77120264"))
# Product defaults (adaptive stride, sweep tolerance 1e-7) against the fixed-step scheme (C oracle mode 0).  Round 4, both
# seeds, 32 768 walkers: max 4.5e-8 / 4.8e-8, 99.9 % below 8e-9 (profiles/r04_soak_parity.json) -- asserted at 1e-7, the
# bound the API documents, at EVERY sample size (round 4's 262 144-walker runs reach 7.2e-8; until round 5 runs beyond 65 536
# walkers were held to 2e-7 only, which would have let a regression of nearly 3 x the observed maximum pass in exactly the
# runs that probe the tail), and the 99.99th percentile at 5e-8 so that tail growth shows before the maximum does.
DEFAULTS_VS_FIXED_MAX, DEFAULTS_VS_FIXED_P999, DEFAULTS_VS_FIXED_P9999 = 1.0e-7, 2.0e-8, 5.0e-8


def _walkers(rng, n, lo, hi):
    P = lo + (hi - lo) * rng.random((n, 6))                                   # uniform over the prior box
    k = n // 4
    for i, name in enumerate(TYPES):                                          # a quarter near the four truths
        t = np.array(TRUTHS[name])
        sl = slice(i * (k // 4), (i + 1) * (k // 4))
        scale = 10.0 ** rng.uniform(-4, -1, (k // 4, 1))
        P[sl] = np.clip(t + scale * np.maximum(np.abs(t), 0.5) * rng.standard_normal((k // 4, 6)), lo, hi)
    f = slice(k, k + n // 10)                                                 # a tenth on the faces of the box
    m = P[f]
    j = rng.integers(0, 6, len(m))
    side = rng.random(len(m)) < 0.5
    m[np.arange(len(m)), j] = np.where(side, lo[j], hi[j])
    P[f] = m
    o = slice(k + n // 10, k + n // 10 + n // 50)                             # 2 %: one ulp outside a face (prior: -inf)
    m = P[o]
    j = rng.integers(0, 6, len(m))
    side = rng.random(len(m)) < 0.5
    m[np.arange(len(m)), j] = np.where(side, np.nextafter(lo[j], -np.inf), np.nextafter(hi[j], np.inf))
    P[o] = m
    return P


@pytest.mark.parametrize("seed", [TUNING_SEED, FRESH_SEED], ids=["tuning-seed", "fresh-seed"])
def test_every_kernel_variant_against_the_c_oracle(gsynth, tarr, seed):
    from magprop_amd import LogProb
    rng = np.random.default_rng(seed)
    lo, hi = gsynth["prior_lower"], gsynth["prior_upper"]
    P = _walkers(rng, N_SOAK, lo, hi)
    ids = rng.integers(0, 4, N_SOAK).astype(np.int32)
    sets = [(gsynth[n + "_x"], gsynth[n + "_y"], gsynth[n + "_yerr"]) for n in TYPES]

    # oracle: spawned CPU workers (clean interpreters: no fork of a process that has initialised the GPU)
    from _soak_worker import oracle_slice
    jobs, where = [], []
    for d in range(4):
        idx = np.nonzero(ids == d)[0]
        for part in np.array_split(idx, 8):
            jobs.append(("synth", P[part], sets[d], tarr, lo, hi, LOG_MASK))
            where.append(part)
    ncpu = max(1, min(16, len(os.sched_getaffinity(0))))
    with mp.get_context("spawn").Pool(ncpu) as pool:
        res = pool.map(oracle_slice, jobs, chunksize=1)
    ref = np.empty(N_SOAK)
    rst = np.empty(N_SOAK, dtype=np.int32)
    for part, (v, s) in zip(where, res):
        ref[part], rst[part] = v, s

    summary = {"n": N_SOAK, "seed": int(seed), "oracle_status_counts": np.bincount(rst, minlength=4).tolist(), "variants": {}}
    # strict: sweep tolerance 1e-9 and every grid interval a step, i.e. the scheme the serial C restatement integrates
    # (mode 0); product defaults: sweep tolerance 1e-7, steps over 1, 2, 4 or 8 grid intervals (include/magprop_amd.h)
    # (batches of <= n_simd / 2 walkers run the team kernels, four wavefronts per walker: <= n_simd / 4 one wavefront per
    # SIMD, above two: mp_kernels.hip launch_lnprob; n_simd = 1 024 on an MI355X)
    for batch, label, env in ((256, "team of 4 wavefronts, one per SIMD", {}),
                              (512, "team of 4 wavefronts, two per SIMD", {}),
                              (1024, "4 steps per lane", {}),
                              (4096, "2 steps per lane", {}),
                              (4096, "2 steps per lane, product defaults (adaptive stride)", {"defaults": True}),
                              (1024, "4 steps per lane, product defaults (adaptive stride)", {"defaults": True}),
                              (512, "team of 4 wavefronts, two per SIMD, product defaults (adaptive stride)", {"defaults": True}),
                              (256, "team of 4 wavefronts, one per SIMD, product defaults (adaptive stride)", {"defaults": True})):
        loose = bool(env.get("defaults"))
        lp_ = LogProb(*sets[0]) if loose else LogProb(*sets[0], sweep_tol=STRICT, max_stride=1)
        for s in sets[1:]:
            lp_.add_dataset(*s)
        out = np.empty(N_SOAK)
        st = np.empty(N_SOAK, dtype=np.int32)
        for a in range(0, N_SOAK, batch):
            o, s = lp_.handle.lnprob_batch(P[a:a + batch], ds_id=ids[a:a + batch], want_status=True)
            out[a:a + batch], st[a:a + batch] = o, s
        conf = np.zeros((4, 4), dtype=int)
        np.add.at(conf, (rst, st), 1)
        both = (rst == 0) & (st == 0)
        rel = np.abs(out[both] - ref[both]) / np.maximum(np.abs(ref[both]), 1.0)
        worst = np.nonzero(both)[0][np.argsort(rel)[::-1][:5]]
        summary["variants"][label] = {"batch": batch, "confusion_oracle_rows_kernel_cols": conf.tolist(),
                                      "worst": [{"i": int(i), "ds": int(ids[i]), "pars": P[i].tolist(), "hip": float(out[i]),
                                                 "oracle_fixed": float(ref[i])} for i in worst],
                                      "status_mismatches": int(np.sum(rst != st)), "max_rel_diff": float(rel.max()),
                                      "median_rel_diff": float(np.median(rel)),
                                      "p999_rel_diff": float(np.quantile(rel, 0.999))}
        assert not np.any(np.isnan(out))
        assert np.all(out[st != 0] == -np.inf)
        assert np.sum(rst != st) == 0, summary["variants"][label]      # ok / flag / prior verdicts identical to the oracle's
        # product defaults against the fixed-step restatement: the adaptive steps add up to ~5e-8 (same size as the
        # scheme's own deviation from the reference's tight-integrator values), 99.9 % of the walkers below 2e-8
        assert rel.max() <= (DEFAULTS_VS_FIXED_MAX if loose else 1e-9), summary["variants"][label]
        assert np.quantile(rel, 0.999) <= (DEFAULTS_VS_FIXED_P999 if loose else 1e-10), summary["variants"][label]
        if loose:
            assert np.quantile(rel, 0.9999) <= DEFAULTS_VS_FIXED_P9999, summary["variants"][label]
    out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"soak_parity_seed{seed}.json"), "w") as f:
            json.dump(summary, f, indent=1)
    print(json.dumps(summary))


@pytest.mark.parametrize("grid,seed", [("L", 77), ("S", FRESH_SEED + 1)], ids=["L-grid", "S-grid-fresh-seed"])
def test_library_variant_against_the_c_oracle(glib, grid, seed):
    """Same soak for the `magnetar` package variant (I = 0.8 M R^2, n = 1 in the ODE, Lprop == 0; 7-parameter
    likelihood with f_beam) over its own prior box (magnetar/mcmc_limits.csv), on both grids of the reference
    (magnetar/funcs.py:132-137; the "S" grid starts at 1 ms: the early hold of the stride policy is a physical time)."""
    from magprop_amd import LogProb, engine, mcmc_eqns
    from _soak_worker import oracle_slice
    tarr = engine.grid(grid)
    n = N_SOAK // 4
    rng = np.random.default_rng(seed)
    lo, hi = mcmc_eqns._bounds(7)
    P = lo + (hi - lo) * rng.random((n, 7))
    P[: n // 50, 6] = np.nextafter(hi[6], np.inf)                               # f_beam one ulp above its bound
    ds = tuple(glib["ds_" + grid])
    parts = np.array_split(np.arange(n), 16)
    ncpu = max(1, min(16, len(os.sched_getaffinity(0))))
    with mp.get_context("spawn").Pool(ncpu) as pool:
        res = pool.map(oracle_slice, [("lib", P[p], ds, tarr, lo, hi, mcmc_eqns.LIB_LOG_MASK) for p in parts], chunksize=1)
    ref = np.concatenate([r[0] for r in res])
    rst = np.concatenate([r[1] for r in res])
    lp_ = LogProb(*ds, variant="lib", GRBtype=grid, lower=lo, upper=hi, sweep_tol=STRICT, max_stride=1)
    for batch in (256, 512, 1024, 4096):
        out = np.empty(n)
        st = np.empty(n, dtype=np.int32)
        for a in range(0, n, batch):
            out[a:a + batch], st[a:a + batch] = lp_.handle.lnprob_batch(P[a:a + batch], want_status=True)
        assert np.array_equal(st, rst), (batch, np.nonzero(st != rst)[0][:5])
        both = rst == 0
        rel = np.abs(out[both] - ref[both]) / np.maximum(np.abs(ref[both]), 1.0)
        assert both.sum() > 0.9 * n and rel.max() <= 1e-9 and np.quantile(rel, 0.999) <= 1e-10, (batch, rel.max())
        assert np.all(out[~both] == -np.inf)
    # the product defaults (adaptive stride) on the same walkers: identical verdicts, values within the documented bound
    lpd = LogProb(*ds, variant="lib", GRBtype=grid, lower=lo, upper=hi)
    worst = {}
    for batch in (256, 1024, 4096):
        out = np.empty(n)
        st = np.empty(n, dtype=np.int32)
        for a in range(0, n, batch):
            out[a:a + batch], st[a:a + batch] = lpd.handle.lnprob_batch(P[a:a + batch], want_status=True)
        assert np.array_equal(st, rst), (batch, np.nonzero(st != rst)[0][:5])
        rel = np.abs(out[both] - ref[both]) / np.maximum(np.abs(ref[both]), 1.0)
        worst[batch] = {"max_rel_diff": float(rel.max()), "p999_rel_diff": float(np.quantile(rel, 0.999))}
        assert rel.max() <= DEFAULTS_VS_FIXED_MAX and np.quantile(rel, 0.999) <= DEFAULTS_VS_FIXED_P999, (batch, rel.max())
    out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"soak_parity_lib_{grid}_seed{seed}.json"), "w") as f:
            json.dump({"n": n, "grid": grid, "seed": int(seed), "status_counts": np.bincount(rst, minlength=4).tolist(),
                       "product_defaults_vs_fixed_steps": worst}, f, indent=1)

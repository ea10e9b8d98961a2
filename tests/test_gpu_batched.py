"""GPU parity tests of the two batched workloads BASELINE.json names beyond the plain ensemble pass (run with -m gpu):

  * mode B — `mp_lnprob_batch(..., ltot_out)`: every walker's model light curve written to HBM next to its lnprob
    (north star: "coalesced HBM writes of the model light-curve"; reference: code/synthetic_datasets/funcs.py:229-231);
  * config 5 — the four GRB types, 1 024 walkers each, light curves of mixed lengths, ONE launch, plain
    and through the fused ensemble sampler.
"""
import time

import numpy as np
import pytest

from conftest import CANON, REF_ATOL, REF_RTOL, TIGHT_ATOL, TIGHT_RTOL, TRUTHS, TYPES

pytestmark = pytest.mark.gpu
LOG_MASK = 0b111100


@pytest.fixture(scope="module")
def co():
    from oracle import c_oracle
    return c_oracle


def _long_set(rng, tarr, base_lc, n):
    """A light curve of n points scattered over the grid (first / last exactly on the end knots) around a model curve."""
    x = np.sort(10.0 ** rng.uniform(0.0, 6.0, n))
    x[0], x[-1] = tarr[0], tarr[-1]
    y0 = np.interp(x, tarr, base_lc)
    yerr = 0.2 * y0
    return x, y0 + rng.normal(0, yerr), yerr


# ---------------------------------------------------------------- mode B
@pytest.mark.parametrize("n", [64, 1024, 1300, 2048])
def test_mode_b_batched_light_curves(gsynth, gflag, tarr, n):
    """lnprob_batch(P, want_ltot=True): 64 walkers run the 4-steps-per-lane curve kernel like 1 024 do, 1 300 the
    2-steps-per-lane one, 2 048 two rounds of the 4-steps-per-lane one again (mp_device.h kernel_spl_curves).  Physical parameters on a prior-free handle, so that every row can be compared with
    mp_model_lc of the same walker bit for bit (same variant) — then the same walkers in sampler coordinates with the
    prior: flagged and out-of-prior rows are NaN, the rest agree."""
    import magprop_amd as mpa
    from magprop_amd import _capi, synth
    rng = np.random.default_rng(100 + n)
    base = mpa.model_lum(CANON["Classic"])
    sets = [(gsynth[t + "_x"], gsynth[t + "_y"], gsynth[t + "_yerr"]) for t in TYPES] + [_long_set(rng, tarr, base[1], 410)]
    hp = _capi.Handle(_capi.cfg_synth(), tarr)          # physical parameters, no prior
    hs = _capi.Handle(_capi.cfg_synth(), tarr)          # sampler coordinates + prior box
    hs.set_prior(synth.PRIOR_LOWER, synth.PRIOR_UPPER, synth.LOG_MASK)
    for k, s in enumerate(sets):
        hp.set_dataset(k, *s)
        hs.set_dataset(k, *s)
    hp.set_prior(None, None, 0)

    # walkers: the four canonical parameter sets first, clouds around the four truths, a slice of the prior-wide scan
    # (contains 'flag' points), and a few outside the prior box
    S = np.empty((n, 6))
    S[:4] = [TRUTHS[t] for t in TYPES]
    k_cloud = (n - 4) * 5 // 8
    which = rng.integers(0, 4, k_cloud)
    S[4:4 + k_cloud] = np.array([TRUTHS[TYPES[w]] for w in which]) + 0.02 * rng.standard_normal((k_cloud, 6))
    rest = n - 4 - k_cloud
    flag_rows = np.nonzero(gflag["status"] == 1)[0]
    pick = np.concatenate([flag_rows[:3], rng.choice(len(gflag["pars"]), rest - 3, replace=False)])
    S[4 + k_cloud:] = gflag["pars"][pick]
    S[5, 5] = 3.5                                         # outside the prior (delta)
    S[6, 0] = 1.0e-4                                      # outside the prior (B)
    ids = rng.integers(0, len(sets), n).astype(np.int32)
    ids[:4] = np.arange(4)
    ids[7] = 4                                            # at least one walker on the 410-point light curve
    Pphys = S.copy()
    Pphys[:, 2:] = 10.0 ** S[:, 2:]

    lnp_b, st_b, lt = hp.lnprob_batch(Pphys, ds_id=ids, want_status=True, want_ltot=True)
    if n > hp.n_simd and _capi.curve_steps_per_lane(n, hp.n_simd) == 4:
        # (mode B runs rounds of the 4-steps-per-lane kernel here, a mode A launch of this size the 2-steps-per-lane one: mode A
        # in launches of the same variant, for the comparison to the last digits below)
        parts = [hp.lnprob_batch(Pphys[i:i + 1024], ds_id=ids[i:i + 1024], want_status=True) for i in range(0, n, 1024)]
        lnp_a, st_a = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
    else:
        lnp_a, st_a = hp.lnprob_batch(Pphys, ds_id=ids, want_status=True)
    assert lt.shape == (n, 10001)
    assert np.array_equal(st_a, st_b) and np.sum(st_b == 1) >= 3
    ok = st_b == 0
    assert ok.sum() > 0.8 * n
    # chi^2 from the staged light curve (mode B) and from the two bracketing luminosities (mode A): same numbers
    assert np.allclose(lnp_b[ok], lnp_a[ok], rtol=1e-12, atol=1e-12) and np.all(lnp_b[~ok] == -np.inf)
    assert np.all(np.isnan(lt[~ok])) and np.all(np.isfinite(lt[ok])) and np.all(lt[ok] >= 0.0)
    # every row against mp_model_lc of the same walker
    same_variant = _capi.curve_steps_per_lane(n, hp.n_simd) == 4
    assert same_variant == (n != 1300) or hp.n_simd != 1024
    worst = 0.0
    for i in range(n):
        st_i, out_i = hp.model_lc(Pphys[i])
        assert st_i == st_b[i], (i, st_i, st_b[i])
        if st_i == 0:
            if same_variant:
                assert np.array_equal(lt[i], out_i[1]), i
            else:
                # the other kernel variant (tiles of 128 instead of 256 steps: the adaptive tiles fall differently) at the
                # product's defaults: the light curve of a prior-wide walker agrees to ~1.4e-6 where the propeller switches
                # on or off (a 5e-8 difference in omega, amplified by the n = 10 switch; the reference's own LSODA noise on
                # these curves is 1e-6 ... 5e-6)
                worst = max(worst, float(np.max(np.abs(lt[i] - out_i[1]) / (np.abs(out_i[1]) + 1e-3 * out_i[1].max()))))
    assert worst <= 3e-6, worst
    # the canonical rows against the reference's model_lum (golden_synth.npz, decimated; LSODA noise ~1e-6)
    d = int(gsynth["decim"])
    for k, t in enumerate(TYPES):
        ref = gsynth[t + "_lc"][1]
        assert np.all(np.abs(lt[k, ::d] - ref) <= 1e-12 + 5e-6 * np.abs(ref)), t

    # sampler coordinates + prior: same light curves (10**p on the device vs on the host: a last-bit difference in the
    # parameters), NaN rows for the two walkers outside the box as well
    lnp_s, st_s, lt_s = hs.lnprob_batch(S, ds_id=ids, want_status=True, want_ltot=True)
    assert st_s[5] == 3 and st_s[6] == 3 and np.all(np.isnan(lt_s[[5, 6]])) and lnp_s[5] == -np.inf
    inside = np.ones(n, bool)
    inside[[5, 6]] = False
    assert np.array_equal(st_s[inside], st_b[inside])
    both = ok & inside
    # (a change of a parameter in its last bits moves a prior-wide curve by up to ~1e-8 where the propeller switches)
    scale = lt[both].max(axis=1, keepdims=True)
    assert np.all(np.abs(lt_s[both] - lt[both]) <= 1e-7 * np.abs(lt[both]) + 1e-9 * scale)
    assert np.allclose(lnp_s[both], lnp_b[both], rtol=1e-8, atol=1e-9)
    assert np.all(np.isnan(lt_s[~ok]))
    hp.close()
    hs.close()


def test_mode_b_device_pointers_and_long_light_curve(gsynth, tarr):
    """mp_lnprob_batch_dev with d_ltot on torch tensors: rows identical to the host-buffer form, NaN rows written by the
    kernel itself (the output tensor starts as garbage), chi^2 of a 1 944-point light curve through the curve kernel."""
    import torch
    import magprop_amd as mpa
    from magprop_amd import LogProb
    rng = np.random.default_rng(5)
    base = mpa.model_lum(CANON["Humped"])
    lp_ = LogProb(gsynth["Humped_x"], gsynth["Humped_y"], gsynth["Humped_yerr"])
    lp_.add_dataset(*_long_set(rng, tarr, base[1], 1944))
    n = 200
    P = np.array(TRUTHS["Humped"]) + 0.02 * rng.standard_normal((n, 6))
    P[3] = [1.8171068, 3.68147895, -2.61786801, 1.99840102, -0.33083576, 2.95613803]      # flags
    P[9, 1] = 0.1                                                                         # outside the prior
    ids = (np.arange(n) % 2).astype(np.int32)
    lnp_h, st_h, lt_h = lp_.lnprob_and_curves(P, ds_id=ids)
    lnp_a = lp_(P, ds_id=ids)
    assert st_h[3] == 1 and st_h[9] == 3
    ok = st_h == 0
    assert np.allclose(lnp_h[ok], lnp_a[ok], rtol=1e-12, atol=1e-12)
    dev = torch.device("cuda", lp_.handle.device)
    tp = torch.from_numpy(P).to(dev)
    tid = torch.from_numpy(ids).to(dev)
    lt_d = torch.full((n, 10001), 123.0, dtype=torch.float64, device=dev)
    st_d = torch.full((n,), -7, dtype=torch.int32, device=dev)
    out = lp_.lnprob_device(tp, ds_id=tid, status=st_d, ltot=lt_d)
    torch.cuda.synchronize(dev)
    assert np.array_equal(out.cpu().numpy(), lnp_h) and np.array_equal(st_d.cpu().numpy(), st_h)
    got = lt_d.cpu().numpy()
    assert np.array_equal(got[ok], lt_h[ok]) and np.all(np.isnan(got[~ok])) and np.all(np.isnan(lt_h[~ok]))


# ---------------------------------------------------------------- config 5
def _config5_sets(gsynth, glonglc, tarr):
    """Eleven light curves of 8 ... 1 944 points: the four seeded synthetic sets (50 each), the reference-evaluated long
    sets of golden_longlc.npz (112 / 410 / 1 944 points; real SGRB lengths) and four short ones around
    the model curves of the four types (8 / 63 / 64 / 65 points: either side of the 64 register-resident observations)."""
    import magprop_amd as mpa
    rng = np.random.default_rng(55)
    sets = [(gsynth[t + "_x"], gsynth[t + "_y"], gsynth[t + "_yerr"]) for t in TYPES]
    sets += [tuple(glonglc[f"synth{m}_ds"]) for m in (112, 410, 1944)]
    for t, m in zip(TYPES, (8, 63, 64, 65)):
        sets.append(_long_set(rng, tarr, mpa.model_lum(CANON[t])[1], m))
    return sets


def test_config5_four_types_mixed_lengths_one_launch(gsynth, glonglc, tarr, co):
    """BASELINE config 5 as stated: 4 GRB types x 1 024 walkers, per-walker ds_id over eleven light curves of mixed
    lengths, one launch of 4 096.  Embedded in the batch: the reference-evaluated walkers of golden_synth.npz (4 x 64)
    and golden_longlc.npz (3 x 12), checked against the reference's values; a 320-walker sample against the C oracle."""
    from magprop_amd import LogProb
    sets = _config5_sets(gsynth, glonglc, tarr)
    assert sorted(len(s[0]) for s in sets) == [8, 50, 50, 50, 50, 63, 64, 65, 112, 410, 1944]
    lp_ = LogProb(*sets[0])                                    # product default: against the reference's values
    from magprop_amd import _capi as _c
    lps = LogProb(*sets[0], sweep_tol=_c.SWEEP_TOL_STRICT, max_stride=1)   # strict, every grid interval a step: against the serial restatement
    for s in sets[1:]:
        lp_.add_dataset(*s)
        lps.add_dataset(*s)
    rng = np.random.default_rng(2026)
    nw = 1024
    P = np.empty((4 * nw, 6))
    ids = np.empty(4 * nw, dtype=np.int32)
    for k, t in enumerate(TYPES):
        blk = slice(k * nw, (k + 1) * nw)
        P[blk] = np.array(TRUTHS[t]) + 0.01 * rng.standard_normal((nw, 6))
        ids[blk] = np.where(np.arange(nw) < nw // 2, k, rng.integers(4, len(sets), nw))   # half on the type's own set
        P[blk][:64] = gsynth[t + "_pars"]                      # the reference-evaluated cloud of this type, on its set
        ids[blk][:64] = k
    gold = []                                                  # (rows of the batch, reference values)
    for j, m in enumerate((112, 410, 1944)):                   # Humped-type walkers on the long reference sets
        rows = np.arange(64 + 12 * j, 64 + 12 * (j + 1))
        P[rows] = glonglc[f"synth{m}_pars"]
        ids[rows] = 4 + j
        gold.append((rows, glonglc[f"synth{m}_lnprob"], glonglc[f"synth{m}_lnprob_tight"], glonglc[f"synth{m}_status"]))
    for k, t in enumerate(TYPES):
        gold.append((np.arange(k * nw, k * nw + 64), gsynth[t + "_lnprob"], gsynth[t + "_lnprob_tight"], gsynth[t + "_status"]))
    P[5 * 64] = gsynth["prior_upper"] + 1.0                    # one walker outside the prior

    out, st = lp_.handle.lnprob_batch(P, ds_id=ids, want_status=True)
    assert out.shape == (4096,) and not np.any(np.isnan(out)) and st[5 * 64] == 3
    for rows, ref, tight, rst in gold:
        assert np.array_equal(st[rows], rst)
        ok = rst == 0
        o = out[rows]
        assert np.all(np.abs(o[ok] - ref[ok]) <= REF_ATOL + REF_RTOL * np.abs(ref[ok]))
        assert np.all(np.abs(o[ok] - tight[ok]) <= TIGHT_ATOL + TIGHT_RTOL * np.abs(tight[ok]))
        assert np.all(o[~ok] == -np.inf)
    sample = np.concatenate([rng.choice(4096, 300, replace=False), [0, 1023, 1024, 2047, 2048, 3071, 3072, 4095],
                             np.nonzero(ids == 6)[0][:6], np.nonzero(ids == 7)[0][:6]])
    out_s, st_s = lps.handle.lnprob_batch(P, ds_id=ids, want_status=True)
    assert np.array_equal(st_s, st)
    fin = st == 0
    assert np.all(np.abs(out[fin] - out_s[fin]) <= 1e-7 * np.abs(out_s[fin]) + 1e-9)       # product defaults (adaptive stride) vs strict fixed steps
    for i in sample:
        x, y, yerr = sets[ids[i]]
        ref, rs = co.lnprob_batch(co.cfg_synth(), P[i], tarr, x, y, yerr, gsynth["prior_lower"], gsynth["prior_upper"], LOG_MASK)
        assert st[i] == rs[0], (i, st[i], rs[0])
        if rs[0] == 0:
            assert abs(out_s[i] - ref[0]) <= 1e-9 * abs(ref[0]) + 1e-9, (i, ids[i], out_s[i], ref[0])
    # size-independent properties at full size: a permutation of the batch permutes the result bit for bit, and the
    # four types evaluated separately (1 024 each: another kernel variant) agree to rounding
    perm = rng.permutation(4096)
    out_p = lp_.handle.lnprob_batch(P[perm], ds_id=ids[perm])
    assert np.array_equal(out_p, out[perm])
    for k in range(4):
        blk = slice(k * nw, (k + 1) * nw)
        o_k = lp_.handle.lnprob_batch(P[blk], ds_id=ids[blk])
        fin = np.isfinite(out[blk])
        assert np.array_equal(np.isfinite(o_k), fin)
        assert np.allclose(o_k[fin], out[blk][fin], rtol=1e-8, atol=1e-9)


def test_config5_through_the_ensemble_sampler(gsynth, glonglc, tarr):
    """Four ensembles of 1 024 walkers, one per GRB type, each on a light curve of a different length (50 / 410 / 8 /
    1 944 points), advanced together by the fused stretch-move kernel: half-steps of 4 x 512 proposals in one launch."""
    import magprop_amd as mpa
    from magprop_amd import EnsembleSampler, LogProb
    chosen = [(gsynth["Humped_x"], gsynth["Humped_y"], gsynth["Humped_yerr"])]
    for t, m, seed in (("Classic", 410, 1), ("Sloped", 8, 3), ("Stuttering", 1944, 2)):
        chosen.append(_long_set(np.random.default_rng(seed), tarr, mpa.model_lum(CANON[t])[1], m))
    rng = np.random.default_rng(9)
    nw, nsteps = 1024, 6
    pos = np.concatenate([np.array(TRUTHS[t]) + 1.0e-4 * rng.standard_normal((nw, 6)) for t in TYPES])
    s = EnsembleSampler(nw, 6, datasets=chosen, seed=5)
    assert s.nensembles == 4 and s.ntotal == 4096
    s.run_mcmc(pos, nsteps)
    chain, lnp = s.get_chain(), s.get_log_prob()
    assert chain.shape == (nsteps, 4096, 6) and np.all(np.isfinite(lnp))
    lp_ = LogProb(*chosen[0])
    for c in chosen[1:]:
        lp_.add_dataset(*c)
    ids = np.repeat(np.arange(4, dtype=np.int32), nw)
    for row in (0, nsteps - 1):
        ref = lp_(chain[row], ds_id=ids)
        assert np.allclose(ref, lnp[row], rtol=1e-8, atol=1e-9)
    af = s.acceptance_fraction.reshape(4, nw).mean(axis=1)
    assert np.all(af > 0.2) and np.all(af < 0.95), af
    moved = np.any(chain[-1] != pos, axis=1).reshape(4, nw).mean(axis=1)
    assert np.all(moved > 0.9)
    # walkers never leave their ensemble: every ensemble stays near its own truth
    for k, t in enumerate(TYPES):
        assert np.all(np.abs(chain[-1, k * nw:(k + 1) * nw].mean(axis=0) - np.array(TRUTHS[t])) < 0.05)
    # same seed, same chain
    s2 = EnsembleSampler(nw, 6, datasets=chosen, seed=5)
    s2.run_mcmc(pos, nsteps)
    assert np.array_equal(s2.get_chain(), chain)
    s.close()
    s2.close()


# ---------------------------------------------------------------- threads and streams (include/magprop_amd.h contract)
def test_one_handle_from_several_threads_and_streams(gsynth, tarr):
    """A handle that holds a long light curve used to own per-walker scratch rows (rounds 1-3; none since round 4): host calls from several Python threads
    (ctypes drops the GIL) and device calls on two torch streams must give the results of the same calls made one
    after the other."""
    import threading
    import torch
    import magprop_amd as mpa
    from magprop_amd import LogProb
    rng = np.random.default_rng(77)
    base = mpa.model_lum(CANON["Humped"])
    lp_ = LogProb(gsynth["Humped_x"], gsynth["Humped_y"], gsynth["Humped_yerr"])
    lp_.add_dataset(*_long_set(rng, tarr, base[1], 900))
    batches = [np.array(TRUTHS["Humped"]) + 0.01 * rng.standard_normal((n, 6)) for n in (700, 1500, 300, 1100)]
    ids = [np.ones(len(b), np.int32) for b in batches]                    # everyone on the 900-point light curve
    want = [lp_(b, ds_id=i) for b, i in zip(batches, ids)]
    got = [None] * len(batches)

    def work(k):
        for _ in range(5):
            got[k] = lp_(batches[k], ds_id=ids[k])

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(batches))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for w, g_ in zip(want, got):
        assert np.array_equal(w, g_)
    # device entry on two streams, interleaved: the launches share nothing writable and may overlap
    dev = torch.device("cuda", lp_.handle.device)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    tb = [torch.from_numpy(b).to(dev) for b in batches]
    ti = [torch.from_numpy(i).to(dev) for i in ids]
    outs = [torch.empty(len(b), dtype=torch.float64, device=dev) for b in batches]
    torch.cuda.synchronize(dev)
    for rep in range(4):
        for k in range(len(batches)):
            with torch.cuda.stream(s1 if k % 2 == 0 else s2):
                lp_.lnprob_device(tb[k], out=outs[k], ds_id=ti[k])
    torch.cuda.synchronize(dev)
    for w, o in zip(want, outs):
        assert np.array_equal(w, o.cpu().numpy())
    # ... and they DO overlap (round 4: no scratch rows, no event between launches of one handle): two 300-walker launches of
    # the long-light-curve kernel fit the device side by side, so alternating two streams must finish clearly sooner than
    # the same launches on one stream
    k = 2                                                      # the 300-walker batch
    o2 = torch.empty_like(outs[k])

    def timed(streams, reps=40):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for r in range(reps):
            with torch.cuda.stream(streams[r % len(streams)]):
                lp_.lnprob_device(tb[k], out=outs[k] if r % 2 == 0 else o2, ds_id=ti[k])
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0
    timed([s1, s2], 10)
    one, two = min(timed([s1]) for _ in range(3)), min(timed([s1, s2]) for _ in range(3))
    assert two < 0.85 * one, (one, two)


def test_longest_light_curves_first_launch_order(gsynth, tarr):
    """A batch that mixes long and short light curves and needs more than one round of the device's wave slots is evaluated
    longest light curves first (mp_kernels.hip order_kernel; index buffers from a ring of 8 per handle, reuse ordered by
    an event).  The order must not show in the results: every walker equals its value from a small batch (no ordering),
    bit for bit; 20 launches alternating two streams (more than the ring holds) with different id patterns all come out
    right; a bad id in an ordered batch is still a bad id."""
    import torch
    import magprop_amd as mpa
    from magprop_amd import LogProb
    rng = np.random.default_rng(404)
    base = mpa.model_lum(CANON["Humped"])
    lp_ = LogProb(gsynth["Humped_x"], gsynth["Humped_y"], gsynth["Humped_yerr"])
    for m in (1500, 300, 8, 700, 100):
        lp_.add_dataset(*_long_set(rng, tarr, base[1], m))
    n = 2 * lp_.handle.n_simd + 1000                                      # more than two waves per SIMD
    P = np.array(TRUTHS["Humped"]) + 0.01 * rng.standard_normal((n, 6))
    patterns = [rng.integers(0, 6, n).astype(np.int32) for _ in range(5)]
    patterns.append(np.zeros(n, np.int32))                                # all short: one class
    want = []
    half = n // 2                                                         # two chunks: the same kernel variant (2 steps per lane),
    assert lp_.handle.n_simd < half <= 2 * lp_.handle.n_simd              # one round of wave slots each: launched in index order
    for ids in patterns:
        want.append(np.concatenate([lp_(P[a:a + half], ds_id=ids[a:a + half]) for a in (0, half)]))
    for ids, w in zip(patterns, want):
        got = lp_(P, ds_id=ids)
        assert np.array_equal(got, w)
        assert np.array_equal(lp_(P[::-1].copy(), ds_id=ids[::-1].copy())[::-1], got)
    dev = torch.device("cuda", lp_.handle.device)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    tP = torch.from_numpy(P).to(dev)
    tids = [torch.from_numpy(i).to(dev) for i in patterns]
    outs = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(20)]
    first = [lp_(P, ds_id=ids) for ids in patterns]
    torch.cuda.synchronize(dev)
    for r in range(20):
        with torch.cuda.stream(s1 if r % 2 == 0 else s2):
            lp_.lnprob_device(tP, out=outs[r], ds_id=tids[r % len(tids)])
    torch.cuda.synchronize(dev)
    for r in range(20):
        assert np.array_equal(outs[r].cpu().numpy(), first[r % len(tids)])
    bad = patterns[0].copy()
    bad[7], bad[n - 3] = 40, -1                                           # out of range / never set
    st = torch.empty(n, dtype=torch.int32, device=dev)
    o = torch.empty(n, dtype=torch.float64, device=dev)
    lp_.handle.lnprob_batch_dev(tP.data_ptr(), n, 6, o.data_ptr(), d_ds_id=torch.from_numpy(bad).to(dev).data_ptr(),
                                d_status=st.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    st, o = st.cpu().numpy(), o.cpu().numpy()
    assert st[7] == 4 and st[n - 3] == 4 and o[7] == -np.inf and o[n - 3] == -np.inf
    keep = np.ones(n, bool)
    keep[[7, n - 3]] = False
    assert np.array_equal(o[keep], first[0][keep])


# ---------------------------------------------------------------- datasets: bad ids, incremental registration
def test_bad_dataset_id_is_never_a_perfect_fit(gsynth):
    """Device-pointer entry (the ids live in HBM, the host cannot validate them): a walker whose ds_id is out of range or
    names a slot that was never set gets lnprob = -inf and MP_STATUS_BADDATASET — not chi^2 = 0 — on both kernel
    families, and its neighbours are unaffected.  The host-buffer entry rejects the same batch with ValueError."""
    import torch
    from magprop_amd import LogProb, _capi
    x, y, yerr = gsynth["Humped_x"], gsynth["Humped_y"], gsynth["Humped_yerr"]
    lp = LogProb(x, y, yerr)
    lp.handle.set_dataset(5, gsynth["Classic_x"], gsynth["Classic_y"], gsynth["Classic_yerr"])   # slots 1-4 stay unset
    rng = np.random.default_rng(5)
    for n in (96, 1200):                                        # 4-steps-per-lane kernel (one wave per SIMD); 2-steps-per-lane kernel
        P = np.array(TRUTHS["Humped"]) + 1.0e-3 * rng.standard_normal((n, 6))
        ids = np.where(np.arange(n) % 2 == 0, 0, 5).astype(np.int32)
        bad = {3: -1, 10: 40, 11: 64, 20: 1000000, 21: 2, 50: 4, n - 1: -2147483648}
        want = lp(P, ds_id=ids)
        for i, v in bad.items():
            ids[i] = v
        dP, dI = torch.from_numpy(P).cuda(), torch.from_numpy(ids).cuda()
        st = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        out = lp.lnprob_device(dP, ds_id=dI, status=st).cpu().numpy()
        st = st.cpu().numpy()
        idx = np.array(sorted(bad))
        assert np.all(out[idx] == -np.inf) and np.all(st[idx] == _capi.STATUS_BADDATASET)
        ok = np.ones(n, bool)
        ok[idx] = False
        assert np.array_equal(out[ok], want[ok]) and np.all(st[ok] == 0) and np.all(np.isfinite(out[ok]))
        with pytest.raises(ValueError):
            lp(P, ds_id=ids)
    # out-of-prior walkers with a bad id still report the dataset problem (a usage error outranks the prior)
    P = np.tile(TRUTHS["Humped"], (64, 1))
    P[:, 5] = 3.5
    st = torch.zeros(64, dtype=torch.int32, device="cuda")
    out = lp.lnprob_device(torch.from_numpy(P).cuda(), ds_id=torch.full((64,), 9, dtype=torch.int32, device="cuda"), status=st)
    assert np.all(out.cpu().numpy() == -np.inf) and np.all(st.cpu().numpy() == _capi.STATUS_BADDATASET)


def test_registering_64_datasets_is_incremental(gsynth, tarr):
    """mp_set_dataset for a NEW slot appends (only the new light curve travels, no device-wide wait unless the arena
    grows): 64 registrations take milliseconds and every slot evaluates as a handle holding that set alone does;
    replacing a slot and growing past the arena's capacity keep every other slot intact."""
    import time
    import magprop_amd as mpa
    from magprop_amd import LogProb
    rng = np.random.default_rng(64)
    base = mpa.model_lum(CANON["Humped"])[1]
    sets = [_long_set(rng, tarr, base, int(m)) for m in rng.integers(8, 400, 64)]
    sets[40] = _long_set(rng, tarr, base, 1944)                 # forces the arena to grow midway
    lp = LogProb(*sets[0])
    t0 = time.perf_counter()
    for s in sets[1:]:
        lp.add_dataset(*s)
    dt = time.perf_counter() - t0
    print(f"63 mp_set_dataset calls: {1e3 * dt:.1f} ms")
    assert dt < 1.0
    P = np.array(TRUTHS["Humped"]) + 1.0e-3 * rng.standard_normal((64, 6))
    got = lp(P, ds_id=np.arange(64, dtype=np.int32))
    for k in (0, 1, 17, 39, 40, 41, 63):
        alone = LogProb(*sets[k])
        assert np.isclose(alone(P[k:k + 1])[0], got[k], rtol=1e-9, atol=0.0), k     # (kernel builds with / without the long-light-curve path)
    # replace slot 17 by another light curve: slot 17 changes, the others do not
    lp.handle.set_dataset(17, *sets[3])
    again = lp(P, ds_id=np.arange(64, dtype=np.int32))
    keep = np.arange(64) != 17
    assert np.array_equal(again[keep], got[keep])
    assert np.isclose(again[17], LogProb(*sets[3])(P[17:18])[0], rtol=1e-9, atol=0.0) and again[17] != got[17]
    with pytest.raises(ValueError):
        lp.add_dataset(*sets[0])                                # a 65th set: MP_MAX_DATASETS


def test_multi_device_handle_deals_a_batch_out_inside_the_library(gsynth):
    """mp_create_multi (ABI 5): ONE handle over several devices, one process, no torch: a host-buffer batch goes out in
    contiguous blocks of ceil(n / G) rows, every device's kernel is enqueued before the first is waited for, one vector comes
    back.  The box has one GPU: the same device listed twice (and three times) -- results and statuses equal the single-device
    handle's bit for bit where the blocks select the same kernel variant, to rounding (same tiles, same policy) where a block
    falls into the team kernels' range; datasets and the prior reach every evaluator; diagnostics are per walker of the batch."""
    from magprop_amd import LogProb, _capi
    sets = [(gsynth[t + "_x"], gsynth[t + "_y"], gsynth[t + "_yerr"]) for t in TYPES]
    one = LogProb(*sets[0])
    two = LogProb(*sets[0], device=[0, 0])
    three = LogProb(*sets[0], device=[0, 0, 0])
    for lp_ in (one, two, three):
        for s_ in sets[1:]:
            lp_.add_dataset(*s_)
    assert one.handle.n_devices == 1 and two.handle.n_devices == 2 and three.handle.n_devices == 3
    rng = np.random.default_rng(11)
    lo, hi = gsynth["prior_lower"], gsynth["prior_upper"]
    for n in (4096, 2400, 1, 7):
        P = lo + (hi - lo) * rng.random((n, 6))
        P[: n // 2] = np.array(TRUTHS["Humped"]) + 1.0e-3 * rng.standard_normal((n // 2, 6))
        if n > 5:
            P[5] = hi + 0.5                                                         # outside the prior
        ids = rng.integers(0, 4, n).astype(np.int32)
        ref, st_ref = one.handle.lnprob_batch(P, ds_id=ids, want_status=True)
        for lp_ in (two, three):
            out, st = lp_.handle.lnprob_batch(P, ds_id=ids, want_status=True)
            assert np.array_equal(st, st_ref)
            ok = st == 0
            assert np.all(out[~ok] == -np.inf)
            per = -(-n // lp_.handle.n_devices)
            if n > 1024 and per > 1024:                                             # 2-steps-per-lane kernel on both sides
                assert np.array_equal(out, ref)
            else:
                assert np.allclose(out[ok], ref[ok], rtol=1e-7, atol=0.0)
            assert len(lp_.handle.last_tiles(n)) == n and np.array_equal(lp_.handle.last_tiles(n) > 0, one.handle.last_tiles(n) > 0)
    # mode B rows land at the right offsets too
    P = np.array(TRUTHS["Classic"]) + 1.0e-3 * rng.standard_normal((6, 6))
    a, sa, la = one.handle.lnprob_batch(P, ds_id=1, want_status=True, want_ltot=True)
    b, sb, lb = two.handle.lnprob_batch(P, ds_id=1, want_status=True, want_ltot=True)
    assert np.array_equal(a, b) and np.array_equal(sa, sb) and np.array_equal(la, lb)
    # one-device entries say so instead of guessing a device
    import torch
    d = torch.zeros(4, 6, dtype=torch.float64, device="cuda")
    with pytest.raises(_capi.MagpropAmdError, match="ONE device"):
        two.lnprob_device(d)
    # the model light curve of one walker: the first device
    assert np.array_equal(two.handle.model_lc(np.array([1.0, 5.0, 1e-3, 100.0, 0.1, 1.0]))[1], one.handle.model_lc(np.array([1.0, 5.0, 1e-3, 100.0, 0.1, 1.0]))[1])

"""GPU tests of the fused stretch-move ensemble sampler through the C ABI."""
import os

import numpy as np
import pytest

from conftest import ROOT, TRUTHS, TYPES

pytestmark = pytest.mark.gpu


def test_gaussian_target_chain_matches_the_oracle_bit_for_bit():
    """Same Philox streams, same unfused arithmetic: the device chain equals the numpy restatement exactly."""
    from magprop_amd import EnsembleSampler
    from oracle import stretch_oracle as so
    rng = np.random.default_rng(5)
    pos = rng.normal(size=(32, 3)) * 2.0 + 1.0
    s = EnsembleSampler(32, 3, target="gaussian", seed=20261003)
    s.run_mcmc(pos, 120)
    chain, lnp, acc = so.run(pos, 120, seed=20261003)
    assert np.array_equal(s.get_chain(), chain)
    assert np.array_equal(s.get_log_prob(), lnp)
    assert np.array_equal(s.get_last_sample()[2], acc)
    # continuing is the same as running in one go; chain views have emcee's (walker, step, par) order
    s2 = EnsembleSampler(32, 3, target="gaussian", seed=20261003)
    s2.run_mcmc(pos, 50)
    s2.run_mcmc(None, 70)
    assert np.array_equal(s2.get_chain(), chain)
    assert s2.chain.shape == (32, 120, 3) and s2.lnprobability.shape == (32, 120)
    s3 = EnsembleSampler(32, 3, target="gaussian", seed=7)
    s3.run_mcmc(pos, 120)
    assert not np.array_equal(s3.get_chain(), chain)


def test_two_ensembles_match_the_oracle():
    from magprop_amd import EnsembleSampler
    from oracle import stretch_oracle as so
    rng = np.random.default_rng(6)
    pos = rng.normal(size=(2 * 16, 2))
    # two Gaussian ensembles advanced together: built through the C ABI directly (no datasets needed for target 1)
    import ctypes as C
    from magprop_amd import _capi, engine
    L = _capi.lib()
    h = _capi.Handle(_capi.cfg_synth(), engine.grid(None))
    sp = L.mp_sampler_create(h._h, 16, 2, 2, None, C.c_uint64(11), C.c_double(2.0), 1)
    assert sp
    dp = C.POINTER(C.c_double)
    p = np.ascontiguousarray(pos)
    chain = np.empty((40, 32, 2)); lnp = np.empty((40, 32))
    assert L.mp_sampler_set_positions(sp, p.ctypes.data_as(dp)) == 0
    assert L.mp_sampler_run(sp, 40, chain.ctypes.data_as(dp), lnp.ctypes.data_as(dp)) == 0
    ref_chain, ref_lnp, _ = so.run(pos, 40, seed=11, n_ensembles=2)
    assert np.array_equal(chain, ref_chain) and np.array_equal(lnp, ref_lnp)
    L.mp_sampler_destroy(sp)
    h.close()


def test_gaussian_target_statistics():
    from magprop_amd import EnsembleSampler
    rng = np.random.default_rng(8)
    s = EnsembleSampler(256, 6, target="gaussian", seed=3)
    s.run_mcmc(rng.normal(size=(256, 6)) * 0.1 + 3.0, 1500)
    tail = s.get_chain()[500:].reshape(-1, 6)
    assert np.all(np.abs(tail.mean(axis=0)) < 0.05)
    assert np.all(np.abs(tail.var(axis=0) - 1.0) < 0.06)
    af = s.acceptance_fraction
    assert 0.3 < af.mean() < 0.7
    tau = s.get_autocorr_time(quiet=True)
    assert tau.shape == (6,) and np.all(tau > 1.0) and np.all(tau < 200.0)


def test_humped_posterior_run(gsynth):
    """config 1 in miniature: synth_mcmc.py:175-185 on the seeded Humped dataset."""
    from magprop_amd import EnsembleSampler, LogProb
    x, y, yerr = gsynth["Humped_x"], gsynth["Humped_y"], gsynth["Humped_yerr"]
    rng = np.random.default_rng(20261003)
    nwalk, nstep = 64, 250
    pos = np.array(TRUTHS["Humped"]) + 1.0e-4 * rng.standard_normal((nwalk, 6))
    s = EnsembleSampler(nwalk, 6, x, y, yerr, seed=42)
    s.run_mcmc(pos, nstep)
    chain, lnp = s.get_chain(), s.get_log_prob()
    assert chain.shape == (nstep, nwalk, 6) and np.all(np.isfinite(lnp))
    # stored log-posteriors are the kernel's values at the stored positions (the sampler's whole-step launch runs one wavefront
    # per walker, a 64-row call of the log-posterior a team of four: same tiles, same policy, rounding apart)
    lp = LogProb(x, y, yerr)
    assert np.allclose(lp(chain[-1]), lnp[-1], rtol=1e-11, atol=0.0)
    assert np.allclose(lp(chain[100]), lnp[100], rtol=1e-11, atol=0.0)
    af = s.acceptance_fraction
    assert 0.15 < af.mean() < 0.7
    # the ensemble has spread out from the 1e-4 ball and stays where the likelihood is high
    assert np.std(chain[-1][:, 0]) > 1e-3
    assert np.median(lnp[-1]) > np.median(lnp[0]) - 10.0
    med = np.median(chain[150:].reshape(-1, 6), axis=0)
    assert np.all(np.abs(med - np.array(TRUTHS["Humped"])) < np.array([0.5, 1.0, 0.5, 0.5, 1.0, 1.0]))


def test_whole_step_launch_equals_half_step_launches(gsynth):
    """Small ensembles run a whole step per launch (the proposals of the first half plus both candidate proposals of every
    walker of the second half, include/magprop_amd.h mp_sampler_set_whole_step): the chain, the acceptance counts and the
    failed proposals are those of one launch per half-step, bit for bit."""
    from magprop_amd import EnsembleSampler
    rng = np.random.default_rng(31)
    p0 = rng.normal(size=(256, 6)) * 0.1 + 3.0
    runs = [EnsembleSampler(256, 6, target="gaussian", seed=9, whole_step=w) for w in (True, False)]
    for s in runs:
        s.run_mcmc(p0, 60)
    assert np.array_equal(runs[0].get_chain(), runs[1].get_chain())
    assert np.array_equal(runs[0].acceptance_fraction, runs[1].acceptance_fraction)
    # the smallest ensembles (one walker per half: its partner is the only other walker) against the numpy restatement
    from oracle import stretch_oracle as so
    for nwalk in (2, 4):
        q0 = rng.normal(size=(nwalk, 3))
        for w in (True, False):
            s = EnsembleSampler(nwalk, 3, target="gaussian", seed=23, whole_step=w)
            s.run_mcmc(q0, 50)
            ref_chain, ref_lnp, ref_acc = so.run(q0, 50, 23)
            assert np.array_equal(s.get_chain(), ref_chain) and np.array_equal(s.acceptance_fraction, ref_acc / 50), (nwalk, w)
    # posterior: one ensemble near the truth; four ensembles, each on its own dataset; walkers all over the prior box
    # (failed proposals); a 600-point light curve (the LONG kernel builds)
    sets = [(gsynth[n + "_x"], gsynth[n + "_y"], gsynth[n + "_yerr"]) for n in TYPES]
    lo, hi = gsynth["prior_lower"], gsynth["prior_upper"]
    xl = np.sort(10.0 ** rng.uniform(0.0, 6.0, 600))
    yl = np.interp(xl, sets[1][0], sets[1][1])
    cases = [("near truth", dict(x=sets[0][0], y=sets[0][1], yerr=sets[0][2]), 64,
              np.array(TRUTHS["Humped"]) + 1.0e-4 * rng.standard_normal((64, 6))),
             ("four ensembles", dict(datasets=sets), 32,
              np.concatenate([np.array(TRUTHS[n]) + 1.0e-4 * rng.standard_normal((32, 6)) for n in TYPES])),
             ("prior-wide", dict(x=sets[0][0], y=sets[0][1], yerr=sets[0][2]), 128, lo + (hi - lo) * rng.random((128, 6))),
             ("long light curve", dict(x=xl, y=yl, yerr=0.2 * yl), 32,
              np.array(TRUTHS["Classic"]) + 1.0e-4 * rng.standard_normal((32, 6)))]
    for label, kw, nwalk, pos in cases:
        out = []
        for w in (True, False):
            s = EnsembleSampler(nwalk, 6, seed=17, whole_step=w, **kw)
            s.run_mcmc(pos, 40)
            nbad, bad = s.get_bad()
            out.append((s.get_chain(), s.get_log_prob(), s.acceptance_fraction, nbad, bad[np.lexsort(bad.T)] if len(bad) else bad))
        for a_, b_ in zip(out[0], out[1]):
            assert np.array_equal(a_, b_), label
        if label == "prior-wide":
            assert out[0][3] > 0                                   # failed proposals occurred and were logged alike


def test_four_grb_ensembles_in_one_launch(gsynth):
    """config 5 shape: one ensemble per GRB type advanced together, each against its own dataset."""
    from magprop_amd import EnsembleSampler, LogProb
    sets = [(gsynth[n + "_x"], gsynth[n + "_y"], gsynth[n + "_yerr"]) for n in TYPES]
    rng = np.random.default_rng(1)
    nwalk = 32
    pos = np.concatenate([np.array(TRUTHS[n]) + 1.0e-4 * rng.standard_normal((nwalk, 6)) for n in TYPES])
    s = EnsembleSampler(nwalk, 6, datasets=sets, seed=5)
    s.run_mcmc(pos, 60)
    chain, lnp = s.get_chain(), s.get_log_prob()
    assert chain.shape == (60, 4 * nwalk, 6) and np.all(np.isfinite(lnp))
    for e, n in enumerate(TYPES):
        lp = LogProb(*sets[e])
        sl = slice(e * nwalk, (e + 1) * nwalk)
        ref = lp(chain[-1][sl])
        assert np.allclose(ref, lnp[-1][sl], rtol=1e-8, atol=1e-9)   # another kernel variant, default sweep tolerance
    assert 0.1 < s.acceptance_fraction.mean() < 0.8


def test_sampler_on_a_long_light_curve(gsynth):
    """A 600-point light curve (longer than the register-resident 64 / 256 observations) through the fused move."""
    from magprop_amd import EnsembleSampler, LogProb, model_lum
    from magprop_amd import engine
    rng = np.random.default_rng(8)
    tarr = engine.grid(None)
    truth = np.array(TRUTHS["Classic"])
    phys = truth.copy()
    phys[2:] = 10.0 ** phys[2:]
    x = np.sort(10.0 ** rng.uniform(0.0, 6.0, 600))
    y0 = model_lum(phys, xdata=x)
    yerr = 0.2 * y0
    y = y0 + rng.normal(0, yerr)
    import os
    for nwalk, env in ((32, {}), (640, {}), (2600, {})):
        # small half-steps (16 and 320 proposals: 4 steps per lane) and a large one (1 300: 2 steps per lane); one wavefront per walker throughout
        pos = truth + 1.0e-4 * rng.standard_normal((nwalk, 6))
        os.environ.update(env)
        try:
            s = EnsembleSampler(nwalk, 6, x, y, yerr, seed=77)
        finally:
            for k_ in env:
                del os.environ[k_]
        s.run_mcmc(pos, 20)
        chain, lnp = s.get_chain(), s.get_log_prob()
        assert np.all(np.isfinite(lnp))
        ref = LogProb(x, y, yerr)(chain[-1])
        assert np.allclose(ref, lnp[-1], rtol=1e-8, atol=1e-9)   # another kernel variant, default sweep tolerance
        assert 0.05 < s.acceptance_fraction.mean() < 0.9
    assert tarr[0] <= x[0] and x[-1] <= tarr[-1]


def test_sampler_library_variant_seven_parameters_short_grb_grid(glib):
    """The `magnetar` package variant through the fused sampler: 7 parameters (f_beam sampled, magnetar/mcmc_eqns.py:22-25,
    64-75), the "S" grid (magnetar/funcs.py:132-133): the stored log-posteriors are those of magprop_amd.lnprob — the
    front end checked against the reference's values in tests/test_gpu_parity.py — at the stored positions."""
    import pandas as pd
    import magprop_amd as mpa
    from magprop_amd import EnsembleSampler
    x, y, yerr = glib["ds_S"]
    data = pd.DataFrame({"t": x, "Lum50": y, "Lum50err": yerr})
    rng = np.random.default_rng(12)
    start = np.array([1.0, 5.0, -2.0, 2.0, 0.5, 0.0, 5.0])                  # B, P, lg MdiscI, lg RdiscI, lg eps, lg delta, f_beam (inside magnetar/mcmc_limits.csv)
    pos = start + 1.0e-3 * rng.standard_normal((56, 7))
    s = EnsembleSampler(56, 7, x, y, yerr, variant="lib", GRBtype="S", seed=21)
    s.run_mcmc(pos, 25)
    chain, lnp = s.get_chain(), s.get_log_prob()
    assert chain.shape == (25, 56, 7) and np.all(np.isfinite(lnp))
    ref = mpa.lnprob(chain[-1], data, "S")
    assert np.allclose(ref, lnp[-1], rtol=1e-8, atol=1e-9)
    assert 0.1 < s.acceptance_fraction.mean() < 0.95 and np.any(chain[-1] != pos)
    lo, hi = mpa.mcmc_eqns._bounds(7)
    assert np.all(chain >= lo) and np.all(chain <= hi)                      # accepted positions never leave the prior box
    s.close()


def test_synth_mcmc_tool_rerun_reproduces_the_chain(tmp_path, capsys):
    """synth_mcmc.py:139-149 (--re-run): the run recorded in <grb>_info.json is repeated bit for bit."""
    import filecmp
    import importlib.util
    import os
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_synth_mcmc", os.path.join(root, "tools", "run_synth_mcmc.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    a, b = tmp_path / "a", tmp_path / "b"
    tool.main(["--grb", "Sloped", "-w", "24", "-s", "12", "--seed", "99", "--out", str(a)])
    os.makedirs(b)
    shutil.copy(a / "Sloped_info.json", b / "Sloped_info.json")
    tool.main(["--grb", "Sloped", "--re-run", "--out", str(b)])
    assert "Mean acceptance fraction" in capsys.readouterr().out
    for name in ("Sloped_chain.csv", "Sloped_lnp.csv", "Sloped_0.csv", "Sloped_5.csv"):
        assert filecmp.cmp(a / name, b / name, shallow=False), name


def test_sampler_argument_validation(gsynth):
    from magprop_amd import EnsembleSampler
    x, y, yerr = gsynth["Humped_x"], gsynth["Humped_y"], gsynth["Humped_yerr"]
    with pytest.raises(ValueError):
        EnsembleSampler(7, 6, x, y, yerr)
    s = EnsembleSampler(16, 6, x, y, yerr)
    with pytest.raises(ValueError):
        s.run_mcmc(np.zeros((15, 6)), 1)
    from magprop_amd import _capi
    with pytest.raises(_capi.MagpropAmdError):
        s.run_mcmc(None, 1)              # no state yet
    # the sharded entry points refuse to run without a state, outside their ranges, or without a row buffer
    assert s.n_slots == 8 and s.step_blocks == 24 and s.step_row_doubles == 6 + 6 and s.row_doubles == 6 + 3
    with pytest.raises(_capi.MagpropAmdError):
        s.step_shard(0, 24, 0)
    s.set_positions(np.array(TRUTHS["Humped"]) + 1.0e-4 * np.random.default_rng(0).standard_normal((16, 6)))
    for lo, hi in ((-1, 4), (0, 25), (5, 4)):
        with pytest.raises(ValueError):
            s.step_shard(lo, hi, 1)
        with pytest.raises(ValueError):
            s.halfstep_shard(0, lo, min(hi, 9) if hi != 25 else 9, 1)
    with pytest.raises(ValueError):
        s.step_shard(0, 24, 0)           # NULL rows for a non-empty range
    with pytest.raises(ValueError):
        s.step_apply(0)
    s.step_shard(3, 3, 0)                # an empty share is legal (more ranks than blocks)
    assert _capi.lib().mp_sampler_set_whole_step(None, 1) != 0


def test_sharded_lnprob_pipelined_over_rccl(gsynth):
    """ShardedLnprob.start/finish with the real collective (RCCL, a process group of one on this 1-GPU box): the
    asynchronous all_gather_into_tensor of batch i overlaps the kernel of batch i+1; results equal the direct call."""
    import os
    import torch
    import torch.distributed as dist
    from magprop_amd import LogProb
    from magprop_amd.distributed import ShardedLnprob
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29800 + os.getpid() % 100))
    x, y, yerr = gsynth["Humped_x"], gsynth["Humped_y"], gsynth["Humped_yerr"]
    lp = LogProb(x, y, yerr, device=0)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sh = ShardedLnprob(lambda p, out: lp.lnprob_device(p, out=out), writes_out=True, always_gather=True)
        gen = torch.Generator(device=dev).manual_seed(3)
        truth = torch.tensor(TRUTHS["Humped"], dtype=torch.float64, device=dev)
        batches = [truth + 1.0e-3 * torch.randn(1024, 6, dtype=torch.float64, device=dev, generator=gen) for _ in range(5)]
        want = [lp.lnprob_device(b).clone() for b in batches]
        got, pending = [], None
        for b in batches:
            t = sh.start(b)
            if pending is not None:
                got.append(sh.finish(pending).clone())
            pending = t
        got.append(sh.finish(pending).clone())
        torch.cuda.synchronize()
        for w, g_ in zip(want, got):
            assert torch.equal(w, g_)
        assert torch.equal(sh(batches[0]), want[0])              # blocking form
        # round 4: the wait for the gathered vector on a CONSUMER stream (what bench.py's independent passes do, so that the
        # compute stream never waits for a collective): caller-owned receive buffers, results read on that stream
        consumer = torch.cuda.Stream(dev)
        recv = torch.empty(len(batches), 1024, dtype=torch.float64, device=dev)
        tickets = [sh.start(b, recv=recv[i]) for i, b in enumerate(batches)]
        outs = [sh.finish(t, stream=consumer) for t in tickets]
        with torch.cuda.stream(consumer):
            got2 = [o.clone() for o in outs]
        torch.cuda.synchronize()
        for w, g_ in zip(want, got2):
            assert torch.equal(w, g_)
        # the walker-sharded sampler with its real collective (all_gather_into_tensor of the outcome rows over RCCL)
        from magprop_amd import EnsembleSampler
        from magprop_amd.distributed import DistributedEnsembleSampler, HipShardEngine
        rng = np.random.default_rng(17)
        pos = np.array(TRUTHS["Humped"]) + 1.0e-4 * rng.standard_normal((128, 6))
        a = EnsembleSampler(128, 6, x, y, yerr, seed=8)
        a.run_mcmc(pos, 15)
        b = EnsembleSampler(128, 6, x, y, yerr, seed=8)
        d = DistributedEnsembleSampler(HipShardEngine(b, dev), always_gather=True)
        chain, lnp = d.run_mcmc(pos, 15)
        torch.cuda.synchronize()
        assert d.rows is not d.send
        assert np.array_equal(chain.cpu().numpy(), a.get_chain()) and np.array_equal(lnp.cpu().numpy(), a.get_log_prob())
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------- walker-sharded ensembles
def _sharded_case(case, world=2):
    import torch.multiprocessing as tmp
    from _shard_worker import run
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() + case["seed"]) % 300
    procs = [ctx.Process(target=run, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_sharded_sampler_gaussian_two_ranks_equal_the_fused_chain_and_the_oracle():
    """Two processes, each evaluating its half of every half-step's proposals with the fused kernel, one gather of the
    outcome rows, commit on both: the chain of mp_sampler_run (single GPU) bit for bit — and of the numpy restatement."""
    from magprop_amd import EnsembleSampler
    from oracle import stretch_oracle as so
    rng = np.random.default_rng(21)
    case = {"nwalkers": 66, "ndim": 3, "seed": 4242, "target": "gaussian", "nsteps": 40, "datasets": None,
            "pos": rng.normal(size=(66, 3)) * 1.5}
    s = EnsembleSampler(66, 3, target="gaussian", seed=4242)
    s.run_mcmc(case["pos"], 40)
    ref_chain, ref_lnp, ref_acc = so.run(case["pos"], 40, 4242)
    assert np.array_equal(s.get_chain(), ref_chain)
    # one ensemble of 66 walkers: the 33 slots of a half-step split 17 / 16; the 99 blocks of a whole step (one gather per
    # step: the first half's proposals and both candidates of every walker of the second half) split 50 / 49
    for whole, ranges in ((False, [(0, 17), (17, 33)]), (True, [(0, 50), (50, 99)])):
        res = _sharded_case(dict(case, whole_step=whole))
        for rank, chain, lnp, af, (lo, hi), _ in res:
            assert np.array_equal(chain, ref_chain) and np.array_equal(lnp, ref_lnp), (whole, rank)
            assert np.array_equal(af, ref_acc / 40)
        assert [r[4] for r in res] == ranges


def test_sharded_sampler_posterior_two_ranks_equal_the_fused_chain(gsynth):
    """The magnetar posterior, two ensembles (a 50-point and a 300-point light curve: the LONG kernel builds in play), 2 x 48
    walkers over two ranks: bit-identical to the single-GPU fused sampler, failed proposals logged on both ranks."""
    from magprop_amd import EnsembleSampler, engine, model_lum
    rng = np.random.default_rng(31)
    tarr = engine.grid(None)
    phys = np.array(TRUTHS["Classic"])
    phys[2:] = 10.0 ** phys[2:]
    x = np.sort(10.0 ** rng.uniform(0.0, 6.0, 300))
    y0 = model_lum(phys, xdata=x)
    sets = [(gsynth["Humped_x"], gsynth["Humped_y"], gsynth["Humped_yerr"]), (x, y0 + rng.normal(0, 0.2 * y0), 0.2 * y0)]
    pos = np.concatenate([np.array(TRUTHS["Humped"]) + 0.3 * rng.standard_normal((48, 6)),      # wide: some proposals fail
                          np.array(TRUTHS["Classic"]) + 1.0e-3 * rng.standard_normal((48, 6))])
    pos = np.clip(pos, gsynth["prior_lower"] + 1e-6, gsynth["prior_upper"] - 1e-6)
    case = {"nwalkers": 48, "ndim": 6, "seed": 99, "target": "posterior", "nsteps": 30, "datasets": sets, "pos": pos}
    s = EnsembleSampler(48, 6, datasets=sets, seed=99)
    s.run_mcmc(pos, 30)
    n_bad = s.get_bad()[0]
    for whole in (False, True):                                # one gather per half-step / per step
        res = _sharded_case(dict(case, whole_step=whole))
        for rank, chain, lnp, af, (lo, hi), nb in res:
            assert np.array_equal(chain, s.get_chain()) and np.array_equal(lnp, s.get_log_prob()), (whole, rank)
            assert np.array_equal(af, s.acceptance_fraction) and nb == n_bad
    assert tarr[0] <= x[0]


def test_sharded_entry_points_in_one_process_and_fbad(gsynth, tmp_path):
    """World size one through the shard / apply entry points equals mp_sampler_run; the failed proposals reach the fbad
    file like the reference's lnprob(…, fbad) (code/synthetic_datasets/mcmc_eqns.py:72-79)."""
    import torch
    from magprop_amd import EnsembleSampler, LogProb
    from magprop_amd.distributed import DistributedEnsembleSampler, HipShardEngine
    x, y, yerr = gsynth["Humped_x"], gsynth["Humped_y"], gsynth["Humped_yerr"]
    rng = np.random.default_rng(3)
    lo, hi = gsynth["prior_lower"], gsynth["prior_upper"]
    # half of the walkers start around a parameter set whose integration fails (SURVEY.md 8c known answer): proposals
    # drawn among them keep landing in the break-up region
    bad = np.array([1.8171068, 3.68147895, -2.61786801, 1.99840102, -0.33083576, 2.95613803])
    pos = np.concatenate([np.clip(bad + 0.02 * rng.standard_normal((32, 6)), lo, hi),
                          np.array(TRUTHS["Humped"]) + 1.0e-2 * rng.standard_normal((32, 6))])
    fbad = tmp_path / "bad.csv"
    a = EnsembleSampler(64, 6, x, y, yerr, seed=12, fbad=str(fbad))
    a.run_mcmc(pos, 25)
    b = EnsembleSampler(64, 6, x, y, yerr, seed=12)
    d = DistributedEnsembleSampler(HipShardEngine(b, f"cuda:{b.handle.device}"))
    chain, lnp = d.run_mcmc(pos, 25)
    torch.cuda.synchronize()
    assert np.array_equal(chain.cpu().numpy(), a.get_chain()) and np.array_equal(lnp.cpu().numpy(), a.get_log_prob())
    # the two drivers can take turns on one sampler: ten more steps, fused on both, continue the same chain
    a.run_mcmc(None, 10)
    b.run_mcmc(None, 10)
    assert np.array_equal(b.get_chain(), a.get_chain()[25:])
    n_bad, rows = a.get_bad()
    assert n_bad == b.get_bad()[0] and n_bad > 0 and len(rows) == n_bad
    logged = np.loadtxt(fbad, delimiter=",", ndmin=2)
    assert logged.shape == (len(rows), 6) and np.allclose(logged, rows, rtol=1e-15)
    # every logged row is inside the prior and fails in the model (status flag / non-finite), as in the reference
    lp_ = LogProb(x, y, yerr)
    out, st = lp_.handle.lnprob_batch(rows, want_status=True)
    assert np.all((st == 1) | (st == 2)) and np.all(out == -np.inf)


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2 --backend gloo` with no launcher around it: bench.py starts its own two rank processes
    (they share this box's one card), and rank 0 prints ONE valid line for n_gpus = 2 that carries the sharded-sampler
    leg and the strong-scaling legs of configs 4 and 5."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6",
                        "--warmup", "2", "--spin-up", "10", "--mcmc-steps", "6", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 6 and d["scaling"] == "weak"
    assert d["config"]["n_walk_total"] == 2048 and d["config"]["n_walk_per_gpu"] == 1024
    assert d["value"] > 0 and d["check"]["n_not_ok"] == 0
    assert d["ensemble_sampler"]["walkers"] == 2048 and 0.2 < d["ensemble_sampler"]["acceptance_fraction"] < 0.8
    assert d["configs"]["4_strong"]["n_walk_total"] == 8192 and d["configs"]["4_strong"]["n_walk_per_gpu"] == 4096
    assert d["configs"]["5_strong"]["n_walk_total"] == 4096 and d["configs"]["5_strong"]["n_not_ok"] == 0
    assert "gloo" in d["config"]["parallelism"]

"""world_size-2 gloo test of the walker sharding + all-gather used by the multi-GPU path."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from magprop_amd.distributed import ShardedLnprob, shard_range


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 512, 1024, 4097):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi, per = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n and hi - lo <= per
                got += list(range(lo, hi))
            assert got == list(range(n))


def _fake_lnprob(p):  # deterministic stand-in for the kernel: any per-walker function will do
    return -0.5 * (p ** 2).sum(dim=1) + torch.sin(p[:, 0])


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(123)
    pars = torch.randn(n, 6, dtype=torch.float64, generator=g)   # replicated proposals
    calls = []

    def eval_local(p):
        calls.append(p.shape[0])
        return _fake_lnprob(p)

    sh = ShardedLnprob(eval_local)
    full = sh(pars).clone()
    # pipelined form: three independent batches, two in flight, results identical to the blocking form
    batches = [pars, pars * 0.5, pars + 1.0]
    tickets, outs = [], []
    for b in batches:
        tickets.append(sh.start(b))
        if len(tickets) == 2:
            outs.append(sh.finish(tickets.pop(0)).clone())
    outs.append(sh.finish(tickets.pop(0)).clone())
    for b, o in zip(batches, outs):
        assert torch.equal(o, _fake_lnprob(b)), rank
    # eval_local writing straight into the send buffer
    def eval_out(p, out):
        out.copy_(_fake_lnprob(p))
    assert torch.equal(ShardedLnprob(eval_out, writes_out=True)(pars), full)
    q.put((rank, full.numpy().copy(), calls[:1]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1024, 1023, 3])
def test_sharded_lnprob_gloo_world2(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 500
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(123)
    pars = torch.randn(n, 6, dtype=torch.float64, generator=g)
    want = _fake_lnprob(pars).numpy()
    for rank, full, calls in res:
        assert np.array_equal(full, want)            # every rank sees the full ensemble, bit-identical
        assert sum(calls) == shard_range(n, rank, 2)[1] - shard_range(n, rank, 2)[0]


def _skew_worker(rank, world, port, q):
    """Five independent passes started back to back, finished late, with rank 1 arriving LATE at every collective: rank 0
    runs ahead and takes send slot k again (pass i + 2) while all-gather i is still waiting for the slow peer."""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 64
    passes = [torch.full((n, 6), float(i + 1), dtype=torch.float64) for i in range(5)]

    def eval_out(p, out):
        if rank == 1:
            time.sleep(0.15)                  # the slow peer
        out.copy_(p[:, 0] * 10.0 + rank)      # pass number and rank readable from every value

    sh = ShardedLnprob(eval_out, writes_out=True)
    recv = [torch.zeros(n, dtype=torch.float64) for _ in passes]
    tickets = [sh.start(p, recv=r) for p, r in zip(passes, recv)]    # all five in flight before the first finish
    got = [sh.finish(t).clone() for t in tickets]
    per = n // world
    ok = all(torch.equal(g[r * per:(r + 1) * per], torch.full((per,), (i + 1) * 10.0 + r, dtype=torch.float64))
             for i, g in enumerate(got) for r in range(world))
    q.put((rank, ok, [g[::per].tolist() for g in got]))
    dist.barrier()
    dist.destroy_process_group()


def test_slot_reuse_waits_for_a_collective_still_in_flight():
    """with finish() waiting on a consumer stream nothing ordered kernel i+2 (which rewrites send slot k)
    behind all-gather i (which reads it).  A rank that runs ahead of a slow peer must still deliver pass i's values."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + os.getpid() % 150
    procs = [ctx.Process(target=_skew_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, seen in res:
        assert ok, (rank, seen)


# ---------------------------------------------------------------- sharded stretch-move sampler
# The product engine is the HIP sampler (tests/test_gpu_sampler.py runs it across two processes on the GPU box); here the
# SAME driver class runs the numpy restatement of the shard / gather / commit protocol over gloo.
N_W, N_DIM, N_ENS, N_STEPS, SEED = 18, 3, 2, 25, 77


def _start_positions():
    return np.random.default_rng(5).normal(size=(N_W * N_ENS, N_DIM)) * 2.0 + 1.0


def _sampler_worker(rank, world, port, q, whole):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from magprop_amd.distributed import DistributedEnsembleSampler
    from oracle.stretch_oracle import NumpyShardEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = NumpyShardEngine(N_W, N_DIM, SEED, n_ensembles=N_ENS)
    s = DistributedEnsembleSampler(eng, whole_step=whole)
    assert s.whole_step == whole
    chain, lnp = s.run_mcmc(_start_positions(), N_STEPS)
    q.put((rank, chain.numpy().copy(), lnp.numpy().copy(), s.acceptance_fraction.copy(), eng.calls, (s.lo, s.hi, s.per)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("whole", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_distributed_ensemble_sampler_equals_the_single_process_chain(world, whole):
    """Sharding the half-step's proposals over the ranks changes nothing: every rank ends with the chain of the
    unsharded restatement of the move (oracle/stretch_oracle.run), bit for bit, having evaluated only its own block.
    whole: the whole-step protocol (one gather per step; the blocks of a step are the first half's proposals and both
    candidate proposals of every walker of the second half: 3/2 x walkers evaluations, each exactly once across the group)."""
    from oracle import stretch_oracle as so
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() + world + 7 * whole) % 200
    procs = [ctx.Process(target=_sampler_worker, args=(r, world, port, q, whole)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    chain, lnp, acc = so.run(_start_positions(), N_STEPS, SEED, n_ensembles=N_ENS)
    n_slots = (N_W // 2) * N_ENS
    units, launches = (3 * n_slots, 1) if whole else (n_slots, 2)   # what a launch shards, launches per step
    total_calls = 0
    for rank, c, l, af, calls, (lo, hi, per) in res:
        assert np.array_equal(c, chain) and np.array_equal(l, lnp)
        assert np.array_equal(af, acc / N_STEPS)
        assert calls == (hi - lo) * launches * N_STEPS and lo == min(rank * per, units)
        total_calls += calls
    assert total_calls == units * launches * N_STEPS                # every evaluation exactly once across the group


@pytest.mark.parametrize("whole", [False, True])
def test_distributed_ensemble_sampler_single_process_and_continuation(whole):
    from magprop_amd.distributed import DistributedEnsembleSampler
    from oracle import stretch_oracle as so
    eng = so.NumpyShardEngine(N_W, N_DIM, SEED, n_ensembles=N_ENS)
    s = DistributedEnsembleSampler(eng, whole_step=whole)
    c1, l1 = s.run_mcmc(_start_positions(), 10)
    c2, l2 = s.run_mcmc(None, N_STEPS - 10)                          # continue from the resident state
    chain, lnp, acc = so.run(_start_positions(), N_STEPS, SEED, n_ensembles=N_ENS)
    assert np.array_equal(np.concatenate([c1.numpy(), c2.numpy()]), chain)
    assert np.array_equal(np.concatenate([l1.numpy(), l2.numpy()]), lnp)
    assert s.run_mcmc(None, 3, store=False) == (None, None) and s.iteration == N_STEPS + 3

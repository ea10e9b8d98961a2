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


# ---------------------------------------------------------------- sharded stretch-move sampler
def _gauss(p):
    return -0.5 * (p ** 2).sum(dim=1)


def _sampler_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from magprop_amd.distributed import DistributedStretchSampler
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    pos = torch.randn(48, 4, dtype=torch.float64, generator=g) * 2.0 + 1.0
    s = DistributedStretchSampler(_gauss, 48, 4, seed=77)
    chain, lnp = s.run_mcmc(pos, 60)
    q.put((rank, chain.numpy().copy(), lnp.numpy().copy(), s.acceptance_fraction.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_distributed_stretch_sampler_world2_equals_world1():
    """Sharding the evaluations over two ranks changes nothing: both ranks hold the same chain, equal to the
    single-process chain with the same seed."""
    from magprop_amd.distributed import DistributedStretchSampler
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=_sampler_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(5)
    pos = torch.randn(48, 4, dtype=torch.float64, generator=g) * 2.0 + 1.0
    s = DistributedStretchSampler(_gauss, 48, 4, seed=77)
    chain, lnp = s.run_mcmc(pos, 60)
    for rank, c, l, af in res:
        assert np.array_equal(c, chain.numpy()) and np.array_equal(l, lnp.numpy())
        assert np.array_equal(af, s.acceptance_fraction.numpy())


def test_distributed_stretch_sampler_samples_the_target():
    from magprop_amd.distributed import DistributedStretchSampler
    g = torch.Generator().manual_seed(1)
    s = DistributedStretchSampler(_gauss, 64, 3, seed=3)
    chain, lnp = s.run_mcmc(torch.randn(64, 3, dtype=torch.float64, generator=g) * 0.1 + 3.0, 1200)
    tail = chain[400:].reshape(-1, 3)
    assert torch.all(tail.mean(dim=0).abs() < 0.1)
    assert torch.all((tail.var(dim=0) - 1.0).abs() < 0.12)
    assert 0.3 < float(s.acceptance_fraction.mean()) < 0.8
    assert torch.allclose(lnp[-1], _gauss(chain[-1]))
    with pytest.raises(ValueError):
        DistributedStretchSampler(_gauss, 7, 3)

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

    full = ShardedLnprob(eval_local)(pars)
    q.put((rank, full.numpy().copy(), calls))
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

"""Worker of tests/test_gpu_sampler.py::test_sharded_sampler_*: one rank of a walker-sharded ensemble on the GPU box.
Both ranks use the one card of the box (the pool allows six processes on it) and exchange their outcome rows over gloo
through host memory — the product path apart from the transport, which on a multi-GPU node is RCCL."""
import os
import sys


def run(rank, world, port, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import numpy as np
    import torch
    import torch.distributed as dist
    from magprop_amd import EnsembleSampler
    from magprop_amd.distributed import DistributedEnsembleSampler, HipShardEngine
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = EnsembleSampler(case["nwalkers"], case["ndim"], datasets=case.get("datasets"), seed=case["seed"],
                            target=case["target"], device=0)
        ds = DistributedEnsembleSampler(HipShardEngine(s, "cuda:0"), via_host=True, whole_step=case.get("whole_step"))
        assert ds.whole_step == bool(case.get("whole_step", True))
        c1, l1 = ds.run_mcmc(case["pos"], case["nsteps"] // 2)
        c2, l2 = ds.run_mcmc(None, case["nsteps"] - case["nsteps"] // 2)
        torch.cuda.synchronize()
        chain = torch.cat([c1, c2]).cpu().numpy()
        lnp = torch.cat([l1, l2]).cpu().numpy()
        q.put((rank, chain, lnp, np.asarray(ds.acceptance_fraction), (ds.lo, ds.hi), s.get_bad()[0]))
        dist.barrier()
    finally:
        dist.destroy_process_group()

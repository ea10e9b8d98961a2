#!/usr/bin/env python3
"""Driver overhead of the walker-sharded sampler on ONE GPU: fused mp_sampler_run against the shard -> all-gather (RCCL,
group of one) -> commit loop of magprop_amd.distributed.DistributedEnsembleSampler.   python tools/sharded_overhead.py"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from magprop_amd import EnsembleSampler  # noqa: E402
from magprop_amd.distributed import DistributedEnsembleSampler, HipShardEngine  # noqa: E402

g = np.load(os.path.join(ROOT, "tests/golden/golden_synth.npz"))
x, y, yerr = g["Humped_x"], g["Humped_y"], g["Humped_yerr"]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
rng = np.random.default_rng(1)
for nwalk in (1024, 2048, 8192):
    pos = np.array([1, 5, -3, 2, -1, 0.0]) + 1e-4 * rng.standard_normal((nwalk, 6))
    steps = 200
    a = EnsembleSampler(nwalk, 6, x, y, yerr, seed=3)   # fused: a whole step per launch where it fits
    a.run_mcmc(pos, 5, store=False)
    t0 = time.perf_counter()
    a.run_mcmc(None, steps, store=False)
    t_fused = (time.perf_counter() - t0) / steps
    res = {}
    for label, gather, whole in (("half-steps, commit only", False, False), ("half-steps, RCCL all-gather + commit", True, False),
                                 ("whole steps, commit only", False, True), ("whole steps, RCCL all-gather + commit", True, True)):
        b = EnsembleSampler(nwalk, 6, x, y, yerr, seed=3)
        if whole and not HipShardEngine(b, dev).whole_step_ok(1):
            b.close()
            continue
        d = DistributedEnsembleSampler(HipShardEngine(b, dev), always_gather=gather, whole_step=whole)
        d.run_mcmc(pos, 5, store=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d.run_mcmc(None, steps, store=False)
        t_host = (time.perf_counter() - t0) / steps
        torch.cuda.synchronize()
        res[label] = ((time.perf_counter() - t0) / steps, t_host)
        b.close()
    print(f"{nwalk} walkers: fused {t_fused * 1e3:.3f} ms/step; " + "; ".join(
        f"sharded driver ({k}) {v[0] * 1e3:.3f} ms/step (host enqueue {v[1] * 1e3:.3f})" for k, v in res.items()), flush=True)
    a.close()
dist.destroy_process_group()

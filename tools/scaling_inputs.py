#!/usr/bin/env python3
"""GPU box: kernel time against launch size on ONE MI355X -- the inputs of bench.py's `predicted` object
(profiles/scaling_inputs.json; copy gpurun_out/scaling_inputs.json there).  Humped walkers at truth + 1e-4 randn (BASELINE
configs 2 / 4) and the config-5 mixture (four GRB types, thirteen light curves of 8 ... 1 944 points)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from magprop_amd import LogProb  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
rng = np.random.default_rng(20261005)


def timed(lp, P, ids=None, reps=100):
    dP = torch.from_numpy(np.ascontiguousarray(P)).cuda()
    dI = torch.from_numpy(ids).cuda() if ids is not None else None
    out = torch.empty(len(P), dtype=torch.float64, device="cuda")
    for _ in range(10):
        lp.lnprob_device(dP, out=out, ds_id=dI)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        lp.lnprob_device(dP, out=out, ds_id=dI)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


lp = LogProb(g["Humped_x"], g["Humped_y"], g["Humped_yerr"])
near = {}
for n in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
    near[str(n)] = timed(lp, np.array(bench.TRUTH["Humped"]) + 1e-4 * rng.standard_normal((n, 6)))
sets = bench.config5_datasets(g)
lp5 = LogProb(*sets[0])
for s_ in sets[1:]:
    lp5.add_dataset(*s_)
c5 = {}
for n in (128, 256, 512, 1024, 2048, 4096):
    nw = n // 4
    P = np.concatenate([np.array(bench.TRUTH[t]) + 1e-4 * rng.standard_normal((nw, 6)) for t in bench.TYPES])
    ids = np.empty(n, dtype=np.int32)
    for k in range(4):
        ids[k * nw:(k + 1) * nw] = np.where(np.arange(nw) < nw // 2, k, rng.integers(4, len(sets), nw))
    c5[str(n)] = timed(lp5, P, ids)
out = {"build": bench.csrc_hash(), "device": torch.cuda.get_device_name(0), "lnprob_near_truth_ms": near, "config5_ms": c5,
       "collective_overhead_frac": 0.085,
       "what": "HIP-event time per launch over 100 back-to-back launches (tools/scaling_inputs.py); collective_overhead_frac: "
               "one RCCL all-gather per pass kept in flight behind the next kernel, measured in a group of one "
               "(profiles/r04_rccl_overlap.md)"}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "scaling_inputs.json"), "w"), indent=1)
print(json.dumps(out))

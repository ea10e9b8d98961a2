"""Worker of tests/test_gpu_soak.py: the serial C oracle on a slice of walkers (spawned CPU processes; never touches the GPU)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def oracle_slice(job):
    from oracle import c_oracle as co
    variant, pars, ds, tarr, lower, upper, log_mask = job
    x, y, yerr = ds
    cfg = co.cfg_synth() if variant == "synth" else co.cfg_lib()
    return co.lnprob_batch(cfg, pars, tarr, x, y, yerr, lower, upper, log_mask)

#!/usr/bin/env python3
"""bench.py — walker-lnprob evaluations/sec of the HIP hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]

With N > 1 and no WORLD_SIZE in the environment bench.py launches its own N rank processes (one per GPU, before the
parent makes any GPU call), waits for them and exits with the first non-zero child code; rank 0 prints the line.  Under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` it is one rank of that job.

A "step" is one pass of the hot path over one ensemble of proposals: every walker's lnprior + ODE integration over
the 10 001-point grid + luminosity + interpolation + chi^2.  Workloads (BASELINE.json configs; --config):

  2 (default)  Humped synthetic dataset, N_walk = 1024 per GPU, fp64 (configs[1]); with N GPUs the ensemble is
               N x 1024 walkers (weak scaling; N = 8 is configs[3], 8 192 walkers)
  3            Classic synthetic dataset, N_walk = 4096 per GPU (configs[2])
  4            Humped, 8 192 walkers in total, sharded over the N GPUs (configs[3]; strong scaling)
  5            all four GRB types, 1 024 walkers each, light curves of mixed lengths (8 ... 1 944 points) selected per
               walker, 4 096 walkers in total sharded over the N GPUs (configs[4]; strong scaling)

Every rank evaluates its contiguous block of the replicated proposal batch; with N > 1 ONE RCCL all-gather of the lnprob
slices follows so every rank sees the full ensemble (what the stretch move needs).  Proposals are resident in HBM before
the timed region (generated on device, replicated on every rank from a common seed).

Prints one JSON line on rank 0 (contract in the task description) with extra objects:
  roofline         what binds the lnprob kernel.  Mode A moves 60 algorithmic bytes per evaluation, so its HBM fraction is
                   ~1e-4 by construction (kept as roofline.hbm) and there is no MFMA work: the line prices the kernel against
                   the fp64 VECTOR peak with the flops the PMC counters saw (profiles/pmc_figures.json, `stale` if those
                   were collected on another build) and carries roofline.latency, the cycles one walker keeps a wavefront
                   resident: the floor of every launch of up to n_simd walkers.  Mode B (--curve) is priced against HBM.
  predicted        what one-GPU kernel times predict for N GPUs (weak and strong; profiles/scaling_inputs.json): the first
                   8-GPU run is to be read against it.  Also printed by --dry-run.
  configs          the other BASELINE workloads through the same timed loop, shorter: N = 1: configs 3, 4 (8 192 walkers
                   on one GPU), 5 and mode B ("curve": config 2 with the model light curve written to HBM); N > 1: configs
                   4 and 5 with their fixed totals sharded over the N GPUs (strong scaling).
  sustained        N = 1: >= 2 s of back-to-back config-2 passes (the headline's timed region is a few ms), with the
                   shader clock read before / during / after.
  config1          N = 1: BASELINE configs[0] (24 walkers x 50 steps of the stretch move): through the CPU port on this
                   box's host cores, and through the HIP sampler.
  kernel_ms        N = 1: the same kernel on harder inputs: walkers uniform over the prior box, and a burnt-in ensemble
                   (positions after 500 sampler steps), each with its Newton sweeps per tile.
  check            the reference's own golden walkers (tests/golden/golden_synth.npz) evaluated in this run.
  ensemble_sampler the same metric through the device-resident stretch move (N = 1: fused single-GPU sampler;
                   N > 1: walker-sharded half-steps with one all-gather each — dependent launches, nothing overlapped).
  dropin           N = 1: the same metric through the drop-in boundary as the reference's caller uses it: an emcee-shaped
                   loop (vectorize=True) over magprop_amd.synth.lnprob, numpy in / numpy out, two dependent calls of
                   N_walk / 2 proposals per step; with the time inside those host-buffer calls alone.
  cpu_baseline     oracle/lsoda_port.py in its reference_cost mode (scipy odeint + Python RHS that re-derives the constants
                   in every call + element-wise torque loop: within +5 % of the real reference per evaluation, identical
                   values) timed on this box's host cores over a bounded sample of the same walkers (rank 0, at every N,
                   before that rank touches the GPU; the other ranks wait for it in the rendezvous).
`value` (the contract's K timed steps) is the authoritative figure; `sustained` is its >= 2 s cross-check and
`ensemble_sampler` the same metric through the dependent half-steps of ONE ensemble (SURVEY.md 8(d) names both).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before anything loads a HIP runtime (dmabuf IPC for RCCL)

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TYPES = ("Humped", "Classic", "Sloped", "Stuttering")
TRUTH = {"Humped": [1.0, 5.0, -3.0, 2.0, -1.0, 0.0], "Classic": [1.0, 5.0, -3.0, 3.0, -1.0, 0.0],
         "Sloped": [1.0, 1.0, -3.0, 2.0, 1.0, 1.0], "Stuttering": [1.0, 5.0, -5.0, 2.0, -1.0, 2.0]}
CANON = {"Humped": [1.0, 5.0, 1.0e-3, 100.0, 0.1, 1.0], "Classic": [1.0, 5.0, 1.0e-3, 1000.0, 0.1, 1.0],
         "Sloped": [1.0, 1.0, 1.0e-3, 100.0, 10.0, 10.0], "Stuttering": [1.0, 5.0, 1.0e-5, 100.0, 0.1, 100.0]}
PRIOR_LOWER = np.array([1.0e-3, 0.69, -6.0, np.log10(50.0), -2.0, -1.0])
PRIOR_UPPER = np.array([10.0, 10.0, -2.0, np.log10(2000.0), 2.0, 3.0])
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak (spec)
BYTES_PER_EVAL_A = 48 + 8 + 4  # mode A: 6 fp64 parameters in, lnprob + status out
BYTES_LTOT = 10001 * 8         # mode B adds the model light curve
EVENT_EVERY = 4                # kernel duration is sampled with HIP events on every 4th launch of a timed region
# |lnprob - reference| contract of SURVEY.md 8(c); points where the reference's default-tolerance LSODA run itself is off
# by more than this are enumerated in the golden files (*_lsoda_noise_idx) and checked against the tight run instead
REF_ATOL, REF_RTOL = 1.0e-5, 2.0e-6
TIGHT_ATOL, TIGHT_RTOL = 1.0e-7, 1.0e-7


def usable_cores():
    """CPU threads this process may actually use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def csrc_hash():
    """16 hex digits over the sources the kernels are built from (include/magprop_amd.h + magprop_amd/csrc/*): what ties
    profiles/pmc_figures.json and profiles/scaling_inputs.json (counters and kernel times of ONE build) to the build that
    runs.  tools/collect_profiles.py and tools/scaling_inputs.py store it; a mismatch is reported as `stale`."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(ROOT, "include", "magprop_amd.h")] + sorted(
        f for f in glob.glob(os.path.join(ROOT, "magprop_amd", "csrc", "*")) if f.endswith((".h", ".hpp", ".hip", ".cpp")))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


def _interp_ms(table, n):
    """Kernel time of a launch of n walkers from a {walkers: ms} table (log-log between the measured sizes, flat outside)."""
    pts = sorted((int(k), float(v)) for k, v in table.items())
    if n <= pts[0][0]:
        return pts[0][1]
    if n >= pts[-1][0]:
        return pts[-1][1] * n / pts[-1][0]          # beyond the table the device is full: time grows with the walkers
    for (a, ta), (b, tb) in zip(pts, pts[1:]):
        if a <= n <= b:
            w = (np.log(n) - np.log(a)) / (np.log(b) - np.log(a))
            return float(np.exp((1 - w) * np.log(ta) + w * np.log(tb)))


def predicted_scaling(world):
    """What ONE-GPU measurements predict for `world` GPUs (no multi-GPU hardware is available to the builder: the driver's
    8-GPU run is the first; its line is to be read against this).  Inputs: profiles/scaling_inputs.json -- kernel time
    against launch size on one MI355X (tools/scaling_inputs.py) -- and the cost of keeping one RCCL all-gather per pass in
    flight, +8.5 % per pass (measured in a group of one with the real torch.distributed path, profiles/r04_rccl_overlap.md).
    A prediction, not evidence."""
    try:
        inp = json.load(open(os.path.join(ROOT, "profiles", "scaling_inputs.json")))
    except Exception:  # noqa: BLE001
        return None
    ov = float(inp.get("collective_overhead_frac", 0.085))
    near, c5 = inp["lnprob_near_truth_ms"], inp["config5_ms"]

    def leg(total, table, weak):
        per = total if weak else -(-total // world)
        n_all = per * world if weak else total
        # the one-GPU line it is compared with: weak -- the N = 1 run of the same per-GPU load (what the driver's
        # value(N) / value(1) measures); strong -- the same total on one GPU
        n_one = per if weak else n_all
        v1 = n_one / _interp_ms(table, n_one) * 1e3
        tN = _interp_ms(table, per) * (1.0 + (ov if world > 1 else 0.0))
        vN = n_all / tN * 1e3
        return {"walkers_per_gpu": per, "walkers_total": n_all, "kernel_ms_of_the_share": _interp_ms(table, per),
                "ms_per_pass": tN, "evals_per_sec": vN, "one_gpu_walkers": n_one, "one_gpu_evals_per_sec": v1,
                "speedup_vs_one_gpu": vN / v1}
    out = {"inputs": "profiles/scaling_inputs.json", "inputs_build": inp.get("build"), "inputs_stale": inp.get("build") != csrc_hash(),
           "collective_overhead_frac": ov, "n_gpus": world,
           "weak_config2_1024_per_gpu": leg(1024, near, True), "strong_config4_8192_total": leg(8192, near, False),
           "strong_config5_4096_total": leg(4096, c5, False),
           "claim": "north star '>= 6x at 8 GPUs': claimed on WEAK scaling -- value(8 GPUs x 1 024 walkers) / value(1 GPU x 1 024 "
                    "walkers), the ratio the driver forms from bench.py's per-N lines (BASELINE configs[3] is 8 x 1 024 = 8 192 "
                    "walkers); predicted 8 / (1 + collective overhead).  The same 8 192 walkers against ONE GPU holding all of "
                    "them (strong scaling, strong_config4_8192_total) is predicted BELOW 6x: one GPU runs 8 192 walkers two "
                    "wavefronts per SIMD at a higher rate (18 M/s) than 1 024 walkers one per SIMD (13 M/s)",
           "note": "a prediction from one-GPU kernel times; nothing multi-GPU has been measured by the builder"}
    if world == 1:
        p8 = predicted_scaling(8)
        out["at_8_gpus"] = {k: {"speedup_vs_one_gpu": p8[k]["speedup_vs_one_gpu"], "evals_per_sec": p8[k]["evals_per_sec"]}
                            for k in ("weak_config2_1024_per_gpu", "strong_config4_8192_total", "strong_config5_4096_total")}
    return out


# ---------------------------------------------------------------- CPU legs (before this process touches the GPU: they fork)
def cpu_baseline(grb, budget_s, seed):
    """Time the scipy/LSODA port over a bounded sample."""
    import multiprocessing as mp

    from oracle import lsoda_port as lp
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
    x, y, yerr = g[grb + "_x"], g[grb + "_y"], g[grb + "_yerr"]
    tarr = lp.grid("L")
    cores = usable_cores()
    rng = np.random.default_rng(seed)
    P = np.array(TRUTH[grb]) + 1.0e-4 * rng.standard_normal((16384, 6))
    ctx = mp.get_context("fork")
    done, t_used, vals = 0, 0.0, []
    with ctx.Pool(cores, initializer=lp._pool_init, initargs=(tarr, x, y, yerr)) as pool:
        pool.map(lp._pool_eval, list(P[:cores]))                 # warm the workers (imports, first call)
        chunk = cores * 2
        while t_used < budget_s and done + chunk <= len(P):
            t0 = time.perf_counter()
            vals += pool.map(lp._pool_eval, list(P[done:done + chunk]))
            t_used += time.perf_counter() - t0
            done += chunk
    return {"value": done / t_used, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"{done} of the step-0 style walkers ({grb}, truth+1e-4*randn) through oracle/lsoda_port.py "
                      f"(scipy {__import__('scipy').__version__} odeint/LSODA + Python RHS, multiprocessing.Pool({cores}))"
                      f" in {t_used:.1f} s; the port runs with the reference's cost structure (constants re-derived in every RHS "
                      f"call, element-wise torque loop) and was calibrated against the real reference in the development "
                      f"container: 62 vs 59 ms per evaluation on one core, identical lnprob (DESIGN.md section 5)",
            "ms_per_eval_per_core": 1e3 * t_used * cores / done, "check_lnprob0": float(vals[0])}


def config1_cpu(seed, n_walk=24, n_steps=50):
    """BASELINE configs[0] on the host: emcee's stretch move (oracle/stretch_oracle.py, the restatement of the absent
    emcee) over the LSODA port, proposals of a half-step mapped over a process pool as synth_mcmc.py:178-185 does."""
    import multiprocessing as mp

    from oracle import lsoda_port as lp
    from oracle import stretch_oracle as so
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
    x, y, yerr = g["Humped_x"], g["Humped_y"], g["Humped_yerr"]
    cores = usable_cores()
    rng = np.random.default_rng(seed + 24)
    p0 = np.array(TRUTH["Humped"]) + 1.0e-4 * rng.standard_normal((n_walk, 6))
    ctx = mp.get_context("fork")
    with ctx.Pool(min(cores, n_walk), initializer=lp._pool_init, initargs=(lp.grid("L"), x, y, yerr)) as pool:
        pool.map(lp._pool_eval, list(p0[:min(cores, n_walk)]))   # warm the workers
        t0 = time.perf_counter()
        _, _, acc = so.run_batched(p0, n_steps, seed, batch_fn=lambda P: pool.map(lp._pool_eval, list(P)))
        dt = time.perf_counter() - t0
    evals = n_walk * (n_steps + 1)
    return p0, {"seconds": dt, "evals": evals, "evals_per_sec": evals / dt, "cores": min(cores, n_walk),
                "acceptance_fraction": float(acc.mean() / n_steps),
                "what": f"{n_walk} walkers x {n_steps} steps, stretch move (oracle/stretch_oracle.py) over oracle/lsoda_port.py "
                        f"(reference cost structure), half-steps of {n_walk // 2} proposals over multiprocessing.Pool"}


# ---------------------------------------------------------------- self-launch (N > 1 without a launcher)
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n, argv):
    """Start one child per rank (this process has not touched the GPU and never will), wait, propagate failure.
    stdout is inherited, so rank 0's JSON line is this process's output."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc, alive, t_fail = 0, set(range(n)), None
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc, t_fail = code, time.monotonic()
                print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
        if t_fail is not None and alive and time.monotonic() - t_fail > 20.0:
            for r in alive:                                   # the peers of a dead rank would wait in a collective for ever
                procs[r].kill()
            t_fail = time.monotonic() + 1e9
        time.sleep(0.05)
    return rc if rc >= 0 else 1


# ---------------------------------------------------------------- helpers
def read_sclk_mhz(pci=None):
    """Current shader clock in MHz from sysfs (pp_dpm_sclk, the starred level) of the card at PCI address `pci`
    ("dddd:bb:dd.f"; a box of this pool sees the sysfs nodes of all eight cards of its host, so the card has to be
    matched).  None if the card cannot be identified or the node is unreadable."""
    if pci is None:
        return None
    for p in sorted(glob.glob("/sys/class/drm/card*/device")):
        try:
            if os.path.basename(os.path.realpath(p)).lower() != pci.lower():
                continue
            for line in open(os.path.join(p, "pp_dpm_sclk")).read().splitlines():
                if line.rstrip().endswith("*"):
                    return float(line.split(":")[1].strip().lower().split("mhz")[0])
        except (OSError, ValueError, IndexError):
            continue
    return None


def device_pci_address(dev_index):
    """"dddd:bb:dd.0" of a torch device, or None."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(dev_index)
        return f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:  # noqa: BLE001
        return None


def config5_datasets(g):
    """Light curves of 8 ... 1 944 points: the four seeded synthetic sets (50 points), the three long sets of
    tests/golden/golden_longlc.npz (112 / 410 / 1 944: real SGRB lengths), four short ones (8 / 63 / 64 /
    65 points) drawn around the four types' model curves and, when the fixture is present, the two light curves of
    tests/golden/golden_swift.npz whose time stamps are those of real Swift observations (densely clustered early)."""
    from magprop_amd import model_lum
    gl = np.load(os.path.join(ROOT, "tests", "golden", "golden_longlc.npz"))
    sets = [(g[t + "_x"], g[t + "_y"], g[t + "_yerr"]) for t in TYPES]
    sets += [tuple(gl[f"synth{m}_ds"]) for m in (112, 410, 1944)]
    rng = np.random.default_rng(55)
    tarr = np.logspace(0.0, 6.0, 10001)
    for t, m in zip(TYPES, (8, 63, 64, 65)):
        lc = model_lum(CANON[t])[1]
        x = np.sort(10.0 ** rng.uniform(0.0, 6.0, m))
        x[0], x[-1] = tarr[0], tarr[-1]
        y0 = np.interp(x, tarr, lc)
        sets.append((x, y0 + rng.normal(0, 0.2 * y0), 0.2 * y0))
    sw = os.path.join(ROOT, "tests", "golden", "golden_swift.npz")
    if os.path.exists(sw):
        gs = np.load(sw)
        sets += [tuple(gs[k + "_ds"]) for k in ("swift_060614", "swift_051016B") if k + "_ds" in gs]
    return sets


def config5_sampler_sets(g):
    """Four light curves of different lengths, one per GRB type, for the four ensembles of BASELINE config 5 through the
    sampler: the seeded 50-point Humped set, and 410 / 8 / 1 944 points drawn around the Classic / Sloped / Stuttering model
    curves (20 % errors), as tests/test_gpu_batched.py::test_config5_through_the_ensemble_sampler does."""
    from magprop_amd import model_lum
    tarr = np.logspace(0.0, 6.0, 10001)
    sets = [(g["Humped_x"], g["Humped_y"], g["Humped_yerr"])]
    for t, m, seed in (("Classic", 410, 1), ("Sloped", 8, 3), ("Stuttering", 1944, 2)):
        rng = np.random.default_rng(seed)
        x = np.sort(10.0 ** rng.uniform(0.0, 6.0, m))
        x[0], x[-1] = tarr[0], tarr[-1]
        y0 = np.interp(x, tarr, model_lum(CANON[t])[1])
        sets.append((x, y0 + rng.normal(0, 0.2 * y0), 0.2 * y0))
    return sets


def sampler_leg(c, n_walk, datasets, p0, steps, seed, **kw):
    """walkers x steps / s of the device-resident stretch move over `datasets` (one ensemble of n_walk walkers per entry),
    single GPU: 5 untimed steps, then `steps` timed ones, nothing copied back."""
    from magprop_amd import EnsembleSampler
    es = EnsembleSampler(n_walk, 6, datasets=datasets, seed=seed, device=c.dev_index, **kw)
    es.run_mcmc(p0, 5, store=False)
    t = time.perf_counter()
    es.run_mcmc(None, steps, store=False)
    t = time.perf_counter() - t
    n = es.ntotal
    from magprop_amd import _capi
    whole = kw.get("whole_step", True) and _capi.whole_step_fits(3 * (n // 2), es.handle.n_simd)
    out = {"walkers": n, "ensembles": es.nensembles, "n_obs": [int(len(d[0])) for d in datasets], "steps": steps,
           "walker_steps_per_sec": n * steps / t, "ms_per_step": 1e3 * t / steps,
           "acceptance_fraction": float(es.acceptance_fraction.mean()),
           "launches": "one per step (3/2 x walkers evaluations) + commit" if whole else "one fused launch per half-step"}
    es.close()
    return out


def dropin_leg(g, n_walk, steps, seed, device):
    """The drop-in boundary as the reference's own caller drives it (code/synthetic_datasets/synth_mcmc.py:170-185):
    `magprop_amd.synth.lnprob(pars, x, y, yerr, fbad)` with x, y, yerr as pandas Series, called by an emcee-shaped loop
    with vectorize=True -- ONE call with all n_walk rows, then per step TWO dependent calls of n_walk / 2 proposals, numpy
    in / numpy out (the stretch move of tests/test_gpu_emcee_contract.py's test double; emcee itself is absent from the
    image).  walker_steps_per_sec is the whole loop; host_entry_ms_per_call is the time inside the n_walk / 2-row calls
    alone (PCIe both ways, launch, synchronisation and the Python front end included)."""
    import pandas as pd

    import magprop_amd as mpa
    x, y, yerr = (pd.Series(g["Humped_" + k]) for k in ("x", "y", "yerr"))
    args = (x, y, yerr, None)
    rng = np.random.RandomState(seed % (2 ** 31))
    pos = np.array(TRUTH["Humped"]) + 1.0e-4 * rng.standard_normal((n_walk, 6))
    half, a_ = n_walk // 2, 2.0
    lp = np.asarray(mpa.synth.lnprob(pos, *args, device=device))
    t_calls, n_calls, n_acc = 0.0, 0, 0

    def step():
        nonlocal t_calls, n_calls, n_acc
        inds = rng.permutation(n_walk)
        for S, C in ((inds[:half], inds[half:]), (inds[half:], inds[:half])):
            zz = ((a_ - 1.0) * rng.rand(half) + 1.0) ** 2.0 / a_
            partner = C[rng.randint(half, size=half)]
            q = pos[partner] - (pos[partner] - pos[S]) * zz[:, None]
            t0 = time.perf_counter()
            new = mpa.synth.lnprob(q, *args, device=device)
            t_calls += time.perf_counter() - t0
            n_calls += 1
            with np.errstate(invalid="ignore"):
                acc = 5.0 * np.log(zz) + new - lp[S] > np.log(rng.rand(half))
            pos[S[acc]] = q[acc]
            lp[S[acc]] = new[acc]
            n_acc += int(acc.sum())
    for _ in range(5):
        step()
    t_calls, n_calls, n_acc = 0.0, 0, 0
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    return {"walkers": n_walk, "steps": steps, "walker_steps_per_sec": n_walk * steps / dt, "ms_per_step": 1e3 * dt / steps,
            "host_entry_ms_per_call": 1e3 * t_calls / n_calls, "rows_per_call": half, "calls": n_calls,
            "acceptance_fraction": n_acc / (n_walk * steps),
            "entry": "magprop_amd.synth.lnprob(pars, x, y, yerr, fbad), pandas Series arguments, numpy in / numpy out",
            "note": "emcee-shaped loop (vectorize=True): two dependent host-buffer calls of n_walk / 2 proposals per step; "
                    "the loop's own numpy arithmetic (proposals, accept / reject) is inside walker_steps_per_sec, outside "
                    "host_entry_ms_per_call"}


class Ctx:
    """What every leg of one rank shares: device, process group, golden data, command-line switches."""


def run_passes(c, config, scaling, steps, warmup, curve=False, nwalk=None, grb=None, spin_up=0):
    """The timed loop of the contract over one workload: W untimed + K timed passes bracketed by barrier + synchronize,
    max over ranks.  Returns the figures of the leg (rank-local kernel times; whole-job rate)."""
    import torch
    import torch.distributed as dist

    from magprop_amd import LogProb
    from magprop_amd.distributed import ShardedLnprob, shard_range
    a, dev, world, rank, g = c.a, c.dev, c.world, c.rank, c.g
    grb = grb or {2: "Humped", 3: "Classic", 4: "Humped", 5: "Humped"}[config]
    preset = {2: 1024, 3: 4096, 4: 8192, 5: 4096}[config]
    if nwalk is not None:
        n_global = nwalk * world if scaling == "weak" else nwalk
    elif scaling == "weak":
        n_global = preset * world
    else:
        n_global = 8192 if config == 2 else preset
    x, y, yerr = g[grb + "_x"], g[grb + "_y"], g[grb + "_yerr"]
    total = warmup + steps
    gen = torch.Generator(device=dev).manual_seed(a.seed)        # same seed on every rank: replicated proposals
    ds_global = None
    if config == 5:
        sets = config5_datasets(g)
        lp = LogProb(*sets[0], device=c.dev_index)
        for s_ in sets[1:]:
            lp.add_dataset(*s_)
        nw = n_global // 4
        truth = torch.tensor([TRUTH[t] for t in TYPES], dtype=torch.float64, device=dev).repeat_interleave(nw, dim=0)
        rng5 = np.random.default_rng(a.seed)
        ids = np.empty(n_global, dtype=np.int32)
        for k in range(4):   # half of every type's walkers on its own seeded set, the others over the further sets
            ids[k * nw:(k + 1) * nw] = np.where(np.arange(nw) < nw // 2, k, rng5.integers(4, len(sets), nw))
        ds_global = torch.from_numpy(ids).to(dev)
        n_obs_desc = sorted(len(s_[0]) for s_ in sets)
    else:
        lp = LogProb(x, y, yerr, device=c.dev_index)
        truth = torch.tensor(TRUTH[grb], dtype=torch.float64, device=dev)
        n_obs_desc = int(x.size)
    props = truth + a.spread * torch.randn(total, n_global, 6, dtype=torch.float64, device=dev, generator=gen)
    lo, hi, per = shard_range(n_global, rank, world)
    n_local = hi - lo
    ds_local = ds_global[lo:hi].contiguous() if ds_global is not None else None
    ltot = torch.empty(n_local, 10001, dtype=torch.float64, device=dev) if curve else None
    status = torch.zeros(max(n_local, 1), dtype=torch.int32, device=dev)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(total)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(total)]
    step_idx = [0]
    stream = torch.cuda.current_stream(dev)

    def launch(p, out, ids=None, st=None):
        lp.handle.lnprob_batch_dev(p.data_ptr(), p.shape[0], 6, out.data_ptr(),
                                   d_ds_id=ids.data_ptr() if ids is not None else 0,
                                   d_status=st.data_ptr() if st is not None else 0,
                                   d_ltot=ltot.data_ptr() if ltot is not None else 0, stream=stream.cuda_stream)

    def eval_local(p, out=None):
        i = step_idx[0]
        if out is None:
            out = torch.empty(p.shape[0], dtype=torch.float64, device=dev)
        timed = i % EVENT_EVERY == 0 or i == warmup       # HIP events around every 4th launch: each pair costs ~2 us of stream time
        if timed:
            ev0[i].record(stream)
        launch(p, out, ds_local, status)
        if timed:
            ev1[i].record(stream)
        return out

    sharded = ShardedLnprob(eval_local, via_host=(a.backend == "gloo"), writes_out=True, always_gather=a.always_gather)
    # every pass keeps its full lnprob vector: the check sum over all of them is taken after the timed region (the hot
    # path of an ensemble loop does not reduce its log-posteriors; a per-pass reduction would only delay the next launch)
    results = torch.empty(total, per * world, dtype=torch.float64, device=dev)

    def start(i):
        step_idx[0] = i
        return sharded.start(props[i], recv=results[i])        # shard -> kernel -> all-gather (RCCL) of lnprob enqueued

    consumer = torch.cuda.Stream(dev) if a.overlap else None   # the stream that would read the gathered vectors

    def loop(first, last):
        # The passes are independent batches (as the ensembles of BASELINE config 5 are): with --overlap 1 (default)
        # the all-gather of pass i runs on RCCL's stream while the kernel of pass i+1 runs on the compute stream, and
        # the stream that waits for the gathered vector is the consumer's, not the compute stream (a cross-queue wait
        # costs the waiting queue ~10 us even when the collective is long finished: profiles/r04_rccl_overlap.md).
        # --overlap 0 waits for every gather before the next kernel (what ONE ensemble's dependent half-steps see).
        full, pending = None, None
        for i in range(first, last):
            t = start(i)
            if pending is not None:
                full = sharded.finish(pending, stream=consumer)   # stream-level wait for the collective, no host sync
                pending = None
            if a.overlap:
                pending = t
            else:
                full = sharded.finish(t)
        if pending is not None:
            full = sharded.finish(pending, stream=consumer)
        return full

    if n_local > 0 and spin_up > 0:                             # bring the clocks up (untimed, before the warmup steps)
        spin_out = torch.empty(n_local, dtype=torch.float64, device=dev)
        for _ in range(spin_up):
            launch(props[0][lo:hi], spin_out, ds_local, status)
        torch.cuda.synchronize(dev)
    loop(0, warmup)                                            # same ops as the timed loop (lazy kernel loads happen here)
    c.fence()
    t0 = time.perf_counter()
    full = loop(warmup, total)
    t_host = time.perf_counter() - t0                           # host-side enqueue time (diagnostic)
    c.fence()
    dt = time.perf_counter() - t0
    checksum = float(results[warmup:total, :n_global].sum().item())   # all timed passes, after the clock has stopped
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    kern_ms = np.array([ev0[i].elapsed_time(ev1[i]) for i in range(warmup, total) if i % EVENT_EVERY == 0 or i == warmup])
    n_simd = lp.handle.n_simd
    if not curve and config != 5 and 2 * n_local <= n_simd:
        variant = "team of four wavefronts per walker (one step per lane each), " + ("one" if 4 * n_local <= n_simd else "two") + " per SIMD"
    else:
        from magprop_amd import _capi
        # (mode B: by rounds of resident workgroups, magprop_amd/csrc/mp_device.h kernel_spl_curves)
        spl4 = _capi.curve_steps_per_lane(n_local, n_simd) == 4 if curve else n_local <= n_simd
        variant = ("curve kernel, " if curve else "") + ("4 steps per lane, one wavefront per SIMD" if spl4
                                                          else "2 steps per lane, two wavefronts per SIMD")
    variant += "; order-5 exponential Adams-Moulton, steps over 1/2/4/8 grid intervals (adaptive)"
    if config == 5:
        workload = (f"BASELINE config 5: four GRB types x {n_global // 4} walkers at truth+{a.spread:g}*randn, {len(n_obs_desc)} light "
                    f"curves of {n_obs_desc} points selected per walker, {n_global} walkers in one launch per pass")
    else:
        workload = (f"BASELINE config {config}: {grb} synthetic dataset (N_obs=50), {n_local} walkers per GPU "
                    f"({n_global} walkers total), walkers at truth+{a.spread:g}*randn")
    workload += f", 10001-point grid, mode {'B (lnprob + model light curve written to HBM)' if curve else 'A (lnprob only)'}"
    kavg = float(kern_ms.mean()) * 1e-3
    # the slowest walker of a rank decides its pass, and the slowest rank decides the job: the spread over the ranks
    k_ranks = torch.tensor([kavg * 1e3 if n_local > 0 else float("nan")], dtype=torch.float64, device=dev)
    if world > 1:
        k_all = [torch.empty_like(k_ranks) for _ in range(world)]
        if a.backend == "gloo":
            k_cpu = [t.cpu() for t in k_all]
            dist.all_gather(k_cpu, k_ranks.cpu())
            k_all = k_cpu
        else:
            dist.all_gather(k_all, k_ranks)
        k_ranks = torch.cat([t.cpu() for t in k_all])
    k_ranks = k_ranks.cpu().numpy()
    bytes_eval = BYTES_PER_EVAL_A + (BYTES_LTOT if curve else 0)
    achieved = bytes_eval * n_local / kavg / 1e9
    fig = c.pmc.get("curve" if curve else ("config5" if config == 5 else "lnprob"), {}).get(str(n_local), {})
    traffic, flops = fig.get("traffic_bytes"), fig.get("fp64_flops")
    stale = c.pmc.get("build") != csrc_hash()       # the counters were collected on another build than the one that runs
    hbm = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": traffic, "algorithmic_bytes_per_eval": bytes_eval}
    common = {"kernel": "mp::lnprob_team_kernel" if variant.startswith("team") else "mp::lnprob_kernel",
              "kernel_ms_avg": 1e3 * kavg, "kernel_ms_min": float(kern_ms.min()),
              "kernel_ms_max_over_ranks": float(np.nanmax(k_ranks)), "kernel_ms_min_over_ranks": float(np.nanmin(k_ranks)),
              "algorithmic_bytes_per_eval": bytes_eval, "evals_per_launch": n_local}
    if hbm["frac"] < 0.01 and flops is not None:
        # Mode A moves 60 bytes per evaluation: the HBM fraction is ~1e-4 by construction and says nothing.  What binds is
        # the fp64 vector unit's issue (and, with one wavefront per SIMD, the latency of one walker's dependent chain).
        cyc = fig.get("wave_cycles_per_wave")
        roofline = {"bound": "fp64_valu", "achieved": flops / kavg / 1e12, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": flops / kavg / 1e12 / FP64_VALU_PEAK_TFLOPS, "traffic": traffic, "stale": stale, "hbm": hbm,
                    "latency": None if cyc is None else {
                        "wave_cycles_per_walker": cyc, "ms_at_2.4_GHz": cyc / 2.4e6,
                        "note": "cycles one wavefront is resident for one walker (PMC, mean): the floor of any launch of up to "
                                "n_simd walkers, whatever the rate of the vector unit"},
                    "note": "fp64 flops executed (PMC: 64 x (2 FMA + MUL + ADD) instructions, profiles/pmc_figures.json) / kernel "
                            "time measured here vs the fp64 vector peak; HBM (roofline.hbm) does not bind: 60 algorithmic bytes "
                            "per evaluation", **common}
    else:
        roofline = dict(hbm, stale=stale, note="mode B streams the model light curve to HBM; the fp64 recurrence in front of "
                                               "the stores still decides the time: see valu" if curve else
                                               "no PMC figures for this launch size: HBM figures only (it does not bind)", **common)
    res = {
        "value": n_global * steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup, "scaling": scaling,
        "workload": workload, "baseline_config": config, "n_walk_per_gpu": n_local, "n_walk_total": n_global,
        "n_obs": n_obs_desc, "kernel_variant": variant, "sweep_tol": lp.handle.sweep_tol, "policy": lp.handle.policy,
        "roofline": roofline,
        "valu": {"bound": "fp64 VALU issue", "unit": "TFLOP/s", "fp64_flops_per_launch_pmc": flops, "stale": stale,
                 "achieved": None if flops is None else flops / kavg / 1e12, "peak": FP64_VALU_PEAK_TFLOPS,
                 "frac": None if flops is None else flops / kavg / 1e12 / FP64_VALU_PEAK_TFLOPS,
                 "note": "flops executed = 64 x (2 FMA + MUL + ADD) fp64 instructions counted by rocprofv3 PMC for this "
                         "launch size (profiles/pmc_figures.json) / kernel time measured here"},
        "kernel_evals_per_sec_per_gpu": n_local / kavg, "host_enqueue_ms_per_step": 1e3 * t_host / steps,
        "check": {"lnprob0": float(full[0].item()), "n_not_ok": int((status[:n_local] != 0).sum().item()), "checksum": checksum},
    }
    c.last = {"lp": lp, "props": props, "launch": launch, "stream": stream, "truth": truth, "gen": gen, "x": x, "y": y,
              "yerr": yerr, "grb": grb, "n_global": n_global, "n_local": n_local, "lo": lo, "hi": hi, "status": status}
    return res


def brief(r):
    """A sub-leg as it appears under `configs`."""
    return {k: r[k] for k in ("value", "ms_per_step", "steps", "scaling", "workload", "n_walk_per_gpu", "n_walk_total",
                              "kernel_variant", "kernel_evals_per_sec_per_gpu")} | {
        "kernel_ms_avg": r["roofline"]["kernel_ms_avg"], "kernel_ms_max_over_ranks": r["roofline"]["kernel_ms_max_over_ranks"],
        "roofline": {k: r["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel",
                                                   "algorithmic_bytes_per_eval", "stale") if k in r["roofline"]} |
                    ({"hbm": r["roofline"]["hbm"]} if "hbm" in r["roofline"] else {}),
        "valu": {k: r["valu"][k] for k in ("achieved", "peak", "frac", "fp64_flops_per_launch_pmc")},
        "n_not_ok": r["check"]["n_not_ok"]}


def main():
    # The contract is ONE JSON line on stdout.  Native libraries write banners there too (RCCL prints its version block
    # on stdout when a communicator is created), so everything but the result line is sent to stderr at fd level.
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json workload (see the docstring)")
    ap.add_argument("--nwalk", type=int, default=None, help="walkers per GPU (overrides the preset; weak scaling)")
    ap.add_argument("--grb", default=None, choices=list(TRUTH))
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="strong: the preset's walkers are the TOTAL (8 192 for --config 2) and are sharded over the GPUs")
    ap.add_argument("--curve", action="store_true", help="mode B: also write the model light curve to HBM")
    ap.add_argument("--spread", type=float, default=1.0e-4, help="walkers at truth + spread*randn (synth_mcmc.py:175-176)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=20261003)
    ap.add_argument("--spin-up", type=int, default=300,
                    help="untimed launches before the W warmup steps: the CPU-baseline leg leaves the GPU idle for ~15 s and "
                         "the first few dozen launches after that run at idle clocks (a 20-step timed region read 8 %% low)")
    ap.add_argument("--no-mcmc", action="store_true", help="skip the ensemble-sampler leg")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the prior-wide / burnt-in kernel timings, the golden check, the other configs and the sustained leg")
    ap.add_argument("--mcmc-steps", type=int, default=100)
    ap.add_argument("--sustained-seconds", type=float, default=2.0)
    ap.add_argument("--overlap", type=int, default=1, choices=[0, 1],
                    help="1: the all-gather of one pass overlaps the next pass's kernel (independent batches); 0: serialised")
    ap.add_argument("--always-gather", action="store_true",
                    help="diagnostic: run the RCCL all-gather even at N=1 (group of one) to time the collective path")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo is for rehearsing N>1 on a one-GPU box (ranks share the card, "
                         "rows are gathered through host memory) and is never a reported configuration")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check: every rank joins the process group over gloo, all-reduces a one and rank 0 prints "
                         "{\"dry_run\": true, \"ranks_seen\": N}; no GPU work (runs on a box without a GPU)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus, sys.argv[1:]))            # before anything here has touched the GPU

    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    a.gpus = world                                              # under a launcher the launcher's world size rules

    if a.dry_run:
        import torch
        import torch.distributed as dist
        if os.environ.get("MAGPROP_BENCH_FAIL_RANK") == str(rank):
            sys.exit(7)                                         # test hook: a rank that dies before the rendezvous
        seen = 1
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.ones(1)
            dist.all_reduce(t)
            seen = int(t.item())
            dist.destroy_process_group()
        if rank == 0:
            line = {"dry_run": True, "n_gpus": world, "ranks_seen": seen}
            pred = predicted_scaling(world)
            if pred is not None:
                line["predicted"] = pred
            json_out.write(json.dumps(line) + "\n")
            json_out.flush()
        return

    scaling = a.scaling or ("strong" if a.config in (4, 5) else "weak")
    grb = a.grb or {2: "Humped", 3: "Classic", 4: "Humped", 5: "Humped"}[a.config]
    headline = a.config == 2 and not a.curve and a.nwalk is None and a.scaling is None   # the driver's invocation
    cpu = c1 = p0_c1 = None
    if rank == 0 and not a.no_cpu_baseline:
        # before this rank initialises the GPU (the leg forks); at N > 1 the other ranks wait in the rendezvous meanwhile
        cpu = cpu_baseline(grb, a.cpu_seconds, a.seed)
        if world == 1 and headline and not a.no_extra:
            p0_c1, c1 = config1_cpu(a.seed)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if world > 1 and a.backend == "nccl" and n_dev < world:
        sys.exit(f"bench.py --gpus {world}: only {n_dev} GPU(s) visible (RCCL needs one per rank; --backend gloo rehearses on fewer)")
    c = Ctx()
    c.a, c.rank, c.world = a, rank, world
    c.dev_index = local_rank % n_dev                            # == local_rank except in the gloo rehearsal
    torch.cuda.set_device(c.dev_index)
    c.dev = dev = torch.device("cuda", c.dev_index)
    ranks_seen = 1
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)       # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
        ones = torch.ones(1, dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(ones)                                   # the collective path really spans `world` ranks
        ranks_seen = int(ones.item())
    elif a.always_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
    c.fence = fence
    c.g = g = np.load(os.path.join(ROOT, "tests", "golden", "golden_synth.npz"))
    try:
        c.pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_figures.json")))   # from the committed rocprofv3 PMC summaries
    except Exception:  # noqa: BLE001
        c.pmc = {}

    from magprop_amd import EnsembleSampler
    from magprop_amd.distributed import DistributedEnsembleSampler, HipShardEngine

    # ---- the headline leg (the contract's timed region)
    main_leg = run_passes(c, a.config, scaling, a.steps, a.warmup, curve=a.curve, nwalk=a.nwalk, grb=a.grb, spin_up=a.spin_up)
    L = c.last
    lp, launch, stream, truth, gen = L["lp"], L["launch"], L["stream"], L["truth"], L["gen"]
    x, y, yerr, n_global = L["x"], L["y"], L["yerr"], L["n_global"]

    # ---- the same metric through the device-resident ensemble sampler: walkers x steps / s
    mcmc = None
    if not a.no_mcmc and a.config == 5 and world == 1:
        # four ensembles (one per GRB type, light curves of 50 / 410 / 8 / 1 944 points) advanced together
        nw5 = n_global // 4
        rng5 = np.random.default_rng(a.seed + 5)
        p05 = np.concatenate([np.array(TRUTH[t]) + 1.0e-4 * rng5.standard_normal((nw5, 6)) for t in TYPES])
        mcmc = sampler_leg(c, nw5, config5_sampler_sets(g), p05, a.mcmc_steps, a.seed)
        mcmc["note"] = "emcee-style stretch move, four ensembles on light curves of different lengths advanced together"
    elif not a.no_mcmc and a.config != 5 and n_global % 2 == 0:
        p0 = (truth + 1.0e-4 * torch.randn(n_global, 6, dtype=torch.float64, device=dev, generator=gen)).cpu().numpy()
        es = EnsembleSampler(n_global, 6, x, y, yerr, seed=a.seed, device=c.dev_index)
        half_step_launches = None
        if world == 1:
            es.run_mcmc(p0, 5, store=False)
            tm = time.perf_counter()
            es.run_mcmc(None, a.mcmc_steps, store=False)
            tm = time.perf_counter() - tm
            from magprop_amd import _capi
            whole = _capi.whole_step_fits(3 * (n_global // 2), es.handle.n_simd)
            note = ("emcee-style stretch move; a whole step per launch: the proposals of the first half and both candidate "
                    "proposals of every walker of the second half (3/2 x walkers evaluations, a third of them discarded), then "
                    "the decisions in emcee's order" if whole else
                    "emcee-style stretch move, 2 fused kernel launches per step (propose+lnprob+accept+store)")
            acc = float(es.acceptance_fraction.mean())
            if whole:   # the same chain with one launch per half-step (what larger ensembles get)
                es2 = EnsembleSampler(n_global, 6, x, y, yerr, seed=a.seed, device=c.dev_index, whole_step=False)
                es2.run_mcmc(p0, 5, store=False)
                t2 = time.perf_counter()
                es2.run_mcmc(None, a.mcmc_steps, store=False)
                t2 = time.perf_counter() - t2
                half_step_launches = n_global * a.mcmc_steps / t2
                es2.close()
        else:
            dsam = DistributedEnsembleSampler(HipShardEngine(es, dev), via_host=(a.backend == "gloo"))
            dsam.run_mcmc(p0, 5, store=False)
            fence()
            tm = time.perf_counter()
            dsam.run_mcmc(None, a.mcmc_steps, store=False)
            fence()
            tm = time.perf_counter() - tm
            tt = torch.tensor([tm], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tm = float(tt.item())
            if dsam.whole_step:
                note = (f"walker-sharded stretch move, a whole step per launch: every rank evaluates its {dsam.hi - dsam.lo} of the "
                        f"{es.step_blocks} blocks of a step (the first half's proposals and both candidate proposals of every walker "
                        f"of the second half), ONE all-gather of the outcome rows per step ({es.step_row_doubles * 8} B each), "
                        "commit on every rank; dependent launches, nothing overlapped")
            else:
                note = (f"walker-sharded stretch move: per half-step every rank runs the fused kernel over its {dsam.hi - dsam.lo} "
                        f"of the {es.n_slots} proposals, ONE all-gather of the outcome rows ({es.row_doubles * 8} B each), commit "
                        "on every rank; dependent launches, nothing overlapped")
            acc = float(np.mean(dsam.acceptance_fraction))
        mcmc = {"walkers": n_global, "steps": a.mcmc_steps, "walker_steps_per_sec": n_global * a.mcmc_steps / tm,
                "ms_per_step": 1e3 * tm / a.mcmc_steps, "acceptance_fraction": acc, "note": note}
        if world == 1 and half_step_launches:
            mcmc["walker_steps_per_sec_one_launch_per_half_step"] = half_step_launches
        es.close()

    # ---- N = 1 extras: harder inputs for the same kernel, the reference's golden walkers, sustained run, config 1
    extra = golden = sustained = None
    if world == 1 and not a.no_extra and a.config != 5 and not a.curve:
        def time_kernel(P_np, reps=48, lp_=None):
            """Same sampling as the timed loop: HIP events around every EVENT_EVERY-th of `reps` back-to-back launches."""
            lp_ = lp_ or lp
            P = torch.from_numpy(np.ascontiguousarray(P_np)).to(dev)
            out = torch.empty(P.shape[0], dtype=torch.float64, device=dev)
            st = torch.empty(P.shape[0], dtype=torch.int32, device=dev)

            def go():
                lp_.handle.lnprob_batch_dev(P.data_ptr(), P.shape[0], 6, out.data_ptr(), d_status=st.data_ptr(), stream=stream.cuda_stream)
            idx = [r for r in range(reps) if r % EVENT_EVERY == 0]
            e0 = {r: torch.cuda.Event(enable_timing=True) for r in idx}
            e1 = {r: torch.cuda.Event(enable_timing=True) for r in idx}
            for _ in range(3):
                go()
            for r in range(reps):
                if r in e0:
                    e0[r].record(stream)
                go()
                if r in e0:
                    e1[r].record(stream)
            torch.cuda.synchronize(dev)
            ms = float(np.mean([e0[r].elapsed_time(e1[r]) for r in idx]))
            lp_.handle.lnprob_batch(P_np)                                       # host entry: records tiles and sweeps
            return {"kernel_ms": ms, "evals_per_sec": P.shape[0] / ms * 1e3, "tiles_per_walker": lp_.handle.last_mean_tiles,
                    "sweeps_per_tile": lp_.handle.last_mean_sweeps, "not_ok": int((st != 0).sum().item())}
        rngx = np.random.default_rng(a.seed + 1)
        wide = PRIOR_LOWER + (PRIOR_UPPER - PRIOR_LOWER) * rngx.random((n_global, 6))
        es = EnsembleSampler(n_global, 6, x, y, yerr, seed=a.seed + 2, device=c.dev_index)
        burnt = es.run_mcmc(np.array(TRUTH[grb]) + 1.0e-4 * rngx.standard_normal((n_global, 6)), 500, store=False)
        es.close()
        near = L["props"][a.warmup].cpu().numpy()
        extra = {"near_truth": time_kernel(near), "prior_wide": time_kernel(wide), "burnt_in_500_steps": time_kernel(burnt),
                 "note": "HIP events around every 4th of 48 back-to-back launches, as in the timed loop (roofline.kernel_ms_avg)"}
        # the same walkers with every grid interval a step (cfg.max_stride = 1), also at the strict sweep tolerance: what the
        # stride adaptivity and the default tolerance buy (rounds 1-2 ran an order-4 formula over every interval)
        from magprop_amd import LogProb as _LP, _capi as _c
        for key, kw in (("near_truth_fixed_steps", {"max_stride": 1}),
                        ("near_truth_fixed_steps_strict_tol", {"max_stride": 1, "sweep_tol": _c.SWEEP_TOL_STRICT})):
            lpx = _LP(x, y, yerr, device=c.dev_index, **kw)
            extra[key] = time_kernel(near, lp_=lpx)
            lpx.handle.close()
        Pg, tight, ref = g[grb + "_pars"], g[grb + "_lnprob_tight"], g[grb + "_lnprob"]
        noise = g[grb + "_lsoda_noise_idx"] if grb + "_lsoda_noise_idx" in g else np.zeros(0, dtype=int)
        og = lp(Pg)
        fin = np.isfinite(tight)
        plain = fin.copy()
        plain[noise] = False                                    # enumerated LSODA-noise points: judged against the tight run only
        golden = {"walkers": int(len(Pg)), "status_agrees": bool(np.array_equal(np.isfinite(og), np.isfinite(ref))),
                  "max_rel_dev_vs_reference_tight_lsoda": float(np.max(np.abs(og[fin] - tight[fin]) / np.abs(tight[fin]))),
                  "max_rel_dev_vs_reference_default_lsoda": float(np.max(np.abs(og[plain] - ref[plain]) / np.abs(ref[plain]))),
                  "lsoda_noise_points": int(len(noise)),
                  "tolerance": f"{TIGHT_ATOL:g} + {TIGHT_RTOL:g}|ref| (tight LSODA, every point), {REF_ATOL:g} + {REF_RTOL:g}|ref| "
                               "(default LSODA, every point but the enumerated LSODA-noise ones)",
                  "pass": bool(np.all(np.abs(og[fin] - tight[fin]) <= TIGHT_ATOL + TIGHT_RTOL * np.abs(tight[fin])) and
                               np.all(np.abs(og[plain] - ref[plain]) <= REF_ATOL + REF_RTOL * np.abs(ref[plain])))}
        if headline and a.sustained_seconds > 0:
            props, n_local = L["props"], L["n_local"]
            per_pass = main_leg["roofline"]["kernel_ms_avg"] * 1e-3
            n_pass = int(a.sustained_seconds * 1.05 / per_pass) + 1
            outs = torch.empty(8, n_local, dtype=torch.float64, device=dev)
            pci = device_pci_address(c.dev_index)
            clk0 = read_sclk_mhz(pci)
            torch.cuda.synchronize(dev)
            import threading
            mid = []
            timer = threading.Timer(0.5 * a.sustained_seconds, lambda: mid.append(read_sclk_mhz(pci)))   # halfway through
            timer.start()
            t0 = time.perf_counter()
            for i in range(n_pass):
                launch(props[i % props.shape[0]], outs[i % 8], None, L["status"])
            t_enq = time.perf_counter() - t0                     # (the launch queue applies back-pressure: ~ the GPU time)
            torch.cuda.synchronize(dev)
            timer.join()
            clk1 = mid[0] if mid else None
            dts = time.perf_counter() - t0
            clk2 = read_sclk_mhz(pci)
            sustained = {"seconds": dts, "passes": n_pass, "evals_per_sec": n_pass * n_local / dts, "ms_per_pass": 1e3 * dts / n_pass,
                         "host_enqueue_seconds": t_enq, "sclk_mhz": {"before": clk0, "during": clk1, "after": clk2, "pci": pci},
                         "what": f"{n_pass} back-to-back passes of the headline workload ({n_local} walkers), no events, one "
                                 "synchronisation at the end"}
    dropin = None
    if world == 1 and headline and not a.no_mcmc:
        dropin = dropin_leg(g, n_global, max(20, min(a.mcmc_steps, 200)), a.seed, c.dev_index)
    config1 = None
    if world == 1 and headline and not a.no_extra:
        rng1 = np.random.default_rng(a.seed + 24)
        p0 = p0_c1 if p0_c1 is not None else np.array(TRUTH["Humped"]) + 1.0e-4 * rng1.standard_normal((24, 6))
        es = EnsembleSampler(24, 6, g["Humped_x"], g["Humped_y"], g["Humped_yerr"], seed=a.seed, device=c.dev_index)
        es.run_mcmc(p0, 2, store=False)                         # first launches of this batch size
        t0 = time.perf_counter()
        es.run_mcmc(p0, 50, store=True)
        dth = time.perf_counter() - t0
        config1 = {"hip_sampler": {"seconds": dth, "evals": 24 * 51, "evals_per_sec": 24 * 51 / dth,
                                   "acceptance_fraction": float(es.acceptance_fraction.mean()),
                                   "what": "magprop_amd.EnsembleSampler(24, 6).run_mcmc(p0, 50) with the chain copied back to the host"},
                   "cpu_port": c1, "workload": "BASELINE configs[0]: Humped, N_walk=24, N_step=50 (1 224 evaluations)"}
        es.close()

    # ---- the other BASELINE workloads through the same loop
    configs = None
    if headline and not a.no_extra:
        ss, sw, su = min(a.steps, 40), min(a.warmup, 5), min(a.spin_up, 60)

        def sub(config, scal, sampler=False, **kw):
            r = brief(run_passes(c, config, scal, ss, sw, spin_up=su, **kw))
            Lx = c.last
            if sampler and not a.no_mcmc:
                # SURVEY.md 8(d)'s metric for this workload too: the ensemble loop (walkers x steps / s), not only raw passes
                rngs = np.random.default_rng(a.seed + 10 * config)
                ms = max(10, min(a.mcmc_steps, 60))
                if config == 5:
                    nw5 = Lx["n_global"] // 4
                    p0s = np.concatenate([np.array(TRUTH[t]) + 1.0e-4 * rngs.standard_normal((nw5, 6)) for t in TYPES])
                    r["ensemble_sampler"] = sampler_leg(c, nw5, config5_sampler_sets(g), p0s, ms, a.seed)
                else:
                    p0s = np.array(TRUTH[Lx["grb"]]) + 1.0e-4 * rngs.standard_normal((Lx["n_global"], 6))
                    r["ensemble_sampler"] = sampler_leg(c, Lx["n_global"], [(Lx["x"], Lx["y"], Lx["yerr"])], p0s, ms, a.seed)
            Lx["lp"].handle.close()                             # the leg's handle (stream, grid, datasets) is not needed again
            c.last = None
            return r
        if world == 1:
            configs = {"3": sub(3, "weak", sampler=True), "4_one_gpu": sub(4, "strong", sampler=True),
                       "5": sub(5, "strong", sampler=True), "curve": sub(2, "weak", curve=True),
                       # mode B beyond n_simd walkers: four rounds of the 4-steps-per-lane curve kernel at 4 096 walkers (mp_device.h
                       # kernel_spl_curves), and at 8 192 the 2-steps-per-lane curve kernel (the one kernel on an ABI path that
                       # still spills: 76 B of scratch per lane, tools/resource_usage.py)
                       "curve_4096": sub(2, "weak", curve=True, nwalk=4096),
                       "curve_8192": sub(2, "weak", curve=True, nwalk=8192)}
        else:
            configs = {"4_strong": sub(4, "strong"), "5_strong": sub(5, "strong")}

    if rank == 0:
        r = main_leg
        if world == 1:
            par = "single GPU, no collective (one process, one launch per pass)"
        else:
            par = (f"walker-shard x{world} + {'RCCL' if a.backend == 'nccl' else 'gloo (rehearsal)'} all-gather(lnprob)" +
                   (", gather of pass i overlapped with kernel of pass i+1" if a.overlap else ""))
        if world == 1 and a.always_gather:
            par = "single GPU + RCCL all-gather(lnprob) in a group of one (diagnostic)"
        out = {
            "metric": "walker_lnprob_evals_per_sec", "value": r["value"], "unit": "evals/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": r["workload"], "baseline_config": a.config,
                       "n_walk_per_gpu": r["n_walk_per_gpu"], "n_walk_total": r["n_walk_total"], "n_grid": 10001,
                       "n_obs": r["n_obs"], "variant": "synth", "kernel_variant": r["kernel_variant"],
                       "sweep_tol": r["sweep_tol"], "policy": r["policy"], "parallelism": par},
            "authoritative": "value = the contract's K timed steps (raw, independent passes); sustained = its >= 2 s cross-check; "
                             "ensemble_sampler.walker_steps_per_sec = the same metric through one ensemble's dependent half-steps",
            "roofline": r["roofline"], "valu": r["valu"],
            "kernel_evals_per_sec_per_gpu": r["kernel_evals_per_sec_per_gpu"], "spin_up_launches": a.spin_up,
            "host_enqueue_ms_per_step": r["host_enqueue_ms_per_step"], "ranks_seen": ranks_seen,
            "check": r["check"],
        }
        if extra is not None:
            out["kernel_ms"] = extra
            out["check"]["golden"] = golden
        if sustained is not None:
            out["sustained"] = sustained
        if configs is not None:
            out["configs"] = configs
        if config1 is not None:
            out["config1"] = config1
        if mcmc is not None:
            out["ensemble_sampler"] = mcmc
        if dropin is not None:
            out["dropin"] = dropin
        pred = predicted_scaling(world)
        if pred is not None:
            out["predicted"] = pred
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["speedup_vs_cpu_baseline"] = r["value"] / cpu["value"]
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

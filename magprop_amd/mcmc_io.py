"""On-disk formats of the reference's MCMC driver and the autocorrelation estimate it prints.

Readers/writers for what `code/synthetic_datasets/{generate_data,synth_mcmc,plot_synth}.py` exchange:
  <GRB>.csv            columns x,y,yerr                                   (generate_data.py:70-71)
  <GRB>_chain.csv      header "Npars, Nwalk, Nstep", then one row per (step, walker): the Npars parameters and
                       lnprob, '%.6f' and ", "-separated                  (synth_mcmc.py:188-194)
  <GRB>_<k>.csv        one row per step, the walkers' values of parameter k (synth_mcmc.py:197-204)
  <GRB>_lnp.csv        one row per step, the walkers' lnprob              (synth_mcmc.py:207-213)
  <GRB>_info.json      Npars, Nwalk, Nstep, seed, acceptance_fraction, tau (synth_mcmc.py:161-166,223-226)
k-corrected real light curves use columns t, Lum50, Lum50err (data/README.md:116-119, magnetar/mcmc_eqns.py:17-19).
"""
import json

import numpy as np


def read_dataset(path):
    """x, y, yerr from a synthetic-dataset CSV, or t, Lum50, Lum50err from a k-corrected GRB CSV."""
    import pandas as pd
    d = pd.read_csv(path, float_precision="round_trip")
    if {"x", "y", "yerr"} <= set(d.columns):
        return d["x"].values.astype(float), d["y"].values.astype(float), d["yerr"].values.astype(float)
    if {"t", "Lum50", "Lum50err"} <= set(d.columns):
        return d["t"].values.astype(float), d["Lum50"].values.astype(float), d["Lum50err"].values.astype(float)
    raise ValueError(f"{path}: expected columns x,y,yerr or t,Lum50,Lum50err")


def write_dataset(path, x, y, yerr):
    import pandas as pd
    pd.DataFrame({"x": x, "y": y, "yerr": yerr}).to_csv(path, index=False)


def write_chain_files(basename, chain, lnprob):
    """chain (nsteps, nwalkers, npars), lnprob (nsteps, nwalkers) -> the reference's <basename>_chain.csv,
    <basename>_<k>.csv, <basename>_lnp.csv (synth_mcmc.py:188-213)."""
    chain, lnprob = np.asarray(chain, dtype=float), np.asarray(lnprob, dtype=float)
    nstep, nwalk, npars = chain.shape
    # (np.savetxt formats row by row in C: an 8 192-walker x 3 000-step chain is 1.7e8 numbers, minutes of f-strings)
    rows = np.concatenate([chain, lnprob[:, :, None]], axis=2).reshape(nstep * nwalk, npars + 1)
    with open(f"{basename}_chain.csv", "w") as f:
        f.write(f"{npars}, {nwalk}, {nstep}\n")
        np.savetxt(f, rows, fmt="%.6f", delimiter=", ")
    for k in range(npars):
        np.savetxt(f"{basename}_{k}.csv", chain[:, :, k], fmt="%.6f", delimiter=", ")
    np.savetxt(f"{basename}_lnp.csv", lnprob, fmt="%.6f", delimiter=", ")


def read_chain_file(path):
    """Inverse of write_chain_files' <basename>_chain.csv (what plot_synth.py:137-143 consumes)."""
    with open(path) as f:
        npars, nwalk, nstep = (int(v) for v in f.readline().split(","))
        rows = np.array([[float(v) for v in line.split(",")] for line in f if line.strip()])
    rows = rows.reshape(nstep, nwalk, npars + 1)
    return rows[:, :, :npars], rows[:, :, npars]


def write_info(path, npars, nwalk, nstep, seed, acceptance_fraction=None, tau=None):
    info = {"Npars": int(npars), "Nwalk": int(nwalk), "Nstep": int(nstep), "seed": int(seed)}
    if acceptance_fraction is not None:
        info["acceptance_fraction"] = float(np.mean(acceptance_fraction))
    if tau is not None:
        info["tau"] = [float(t) for t in np.atleast_1d(tau)]      # the reference dumps an ndarray here, which json rejects
    with open(path, "w") as f:
        json.dump(info, f)
    return info


def read_info(path):
    with open(path) as f:
        return json.load(f)


# ---- integrated autocorrelation time (what sampler.get_autocorr_time() returns; Sokal's automatic window)
def _autocorr_1d(x):
    n = 1 << int(np.ceil(np.log2(2 * len(x))))
    f = np.fft.fft(x - np.mean(x), n=n)
    acf = np.fft.ifft(f * np.conjugate(f))[: len(x)].real
    return acf / acf[0] if acf[0] != 0 else acf


def integrated_time(chain, c=5.0, tol=50, quiet=True):
    """chain (nsteps, nwalkers, ndim) -> tau[ndim]: mean autocorrelation function over walkers, window M with
    M >= c*tau(M).  With quiet=False raises if the chain is shorter than tol*tau (emcee's AutocorrError)."""
    x = np.asarray(chain, dtype=float)
    nstep, nwalk, ndim = x.shape
    tau = np.empty(ndim)
    for d in range(ndim):
        acf = np.zeros(nstep)
        for k in range(nwalk):
            acf += _autocorr_1d(x[:, k, d])
        acf /= nwalk
        taus = 2.0 * np.cumsum(acf) - 1.0
        m = np.arange(len(taus)) < c * taus
        window = int(np.argmin(m)) if np.any(~m) else len(taus) - 1
        tau[d] = taus[window]
    if not quiet and np.any(tol * tau > nstep):
        raise RuntimeError(f"The chain is shorter than {tol} times the integrated autocorrelation time; tau: {tau}")
    return tau

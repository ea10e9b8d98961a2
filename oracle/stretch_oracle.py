"""oracle/stretch_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Pure-Python/numpy restatement of the ensemble sampler's stretch move as driven by the reference
(`em.EnsembleSampler(...).run_mcmc(pos, Nstep)`, code/synthetic_datasets/synth_mcmc.py:180-185).  emcee itself
is a third-party dependency absent from the reference checkout and from this image (pinned emcee==3.0rc2 in
requirements.txt:4); its published algorithm (Goodman & Weare 2010; emcee's RedBlueMove/StretchMove) is:

  per step: split the walkers at random into two halves; for each half in turn, for every walker k of it
  draw a partner j from the other half and z = ((a-1)u+1)^2/a, propose y = x_j - (x_j - x_k) z, accept with
  probability min(1, z^(ndim-1) p(y)/p(x_k)).

The product's fused HIP kernel uses Philox4x32-10 streams keyed by (seed; step, half, walker); this file uses
the same generator so that, for a target both sides compute identically (the isotropic unit Gaussian test
target), the chains agree bit for bit.  Only tests/ may import this module.
"""
import numpy as np

M32 = 0xFFFFFFFF


def philox4x32_10(k0, k1, c0, c1, c2, c3):
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        n0 = ((p1 >> 32) ^ c1 ^ k0) & M32
        n1 = p1 & M32
        n2 = ((p0 >> 32) ^ c3 ^ k1) & M32
        n3 = p0 & M32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def u01(hi, lo):
    return float(((hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0)


def split(seed, step, ens, n):
    """Fisher-Yates shuffle of range(n) with counters (step, ens, i, 0x5117)."""
    p = list(range(n))
    for i in range(n - 1, 0, -1):
        r = philox4x32_10(seed & M32, seed >> 32, step, ens, i, 0x5117)
        j = ((r[0] << 32) | r[1]) % (i + 1)
        p[i], p[j] = p[j], p[i]
    return p


def gaussian_lnprob(p):
    lnp = 0.0
    for v in p:
        lnp = lnp - (0.5 * v) * v
    return lnp


def run(pos, n_steps, seed, a=2.0, lnprob_fn=gaussian_lnprob, n_ensembles=1):
    """pos: (n_ensembles*n_walkers, ndim).  Returns chain (n_steps, n_total, ndim), chain_lnp, n_accepted."""
    pos = np.array(pos, dtype=np.float64)
    n_total, ndim = pos.shape
    n = n_total // n_ensembles
    half_n = n // 2
    lnp = np.array([lnprob_fn(p) for p in pos])
    acc = np.zeros(n_total, dtype=np.int64)
    chain = np.empty((n_steps, n_total, ndim))
    chain_lnp = np.empty((n_steps, n_total))
    for step in range(n_steps):
        perms = [split(seed, step, e, n) for e in range(n_ensembles)]
        for half in range(2):
            for e in range(n_ensembles):
                base = e * n
                perm = perms[e]
                for slot in range(half_n):
                    k = base + perm[half * half_n + slot]
                    r = philox4x32_10(seed & M32, seed >> 32, step, half, k, 0)
                    r2 = philox4x32_10(seed & M32, seed >> 32, step, half, k, 1)
                    n_comp = n - half_n
                    jc = int(u01(r[0], r[1]) * n_comp)
                    j = base + perm[(1 - half) * half_n + min(jc, n_comp - 1)]
                    zr = (a - 1.0) * u01(r[2], r[3]) + 1.0
                    zz = zr * zr / a
                    prop = pos[j] - (pos[j] - pos[k]) * zz
                    new = lnprob_fn(prop)
                    lnpdiff = (ndim - 1.0) * np.log(zz) + new - lnp[k]
                    with np.errstate(divide="ignore"):
                        accept = lnpdiff > np.log(u01(r2[0], r2[1]))
                    if accept:
                        pos[k] = prop
                        lnp[k] = new
                        acc[k] += 1
                    chain[step, k] = pos[k]
                    chain_lnp[step, k] = lnp[k]
    return chain, chain_lnp, acc

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
target), the chains agree bit for bit.  Only tests/ and the CPU-baseline legs of bench.py (config 1 timed on the host cores) may
import this module.
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


def run_batched(pos, n_steps, seed, a=2.0, batch_fn=None, n_ensembles=1):
    """`run` with the log-posteriors of a half-step evaluated in ONE call, batch_fn(proposals[m, ndim]) -> m values
    (the proposals of a half only read the other half, so the chain is the one `run` produces): what emcee does with
    a pool (code/synthetic_datasets/synth_mcmc.py:178-185).  Returns chain, chain_lnp, n_accepted."""
    pos = np.array(pos, dtype=np.float64)
    n_total, ndim = pos.shape
    n = n_total // n_ensembles
    half_n = n // 2
    lnp = np.array(batch_fn(pos), dtype=np.float64)
    acc = np.zeros(n_total, dtype=np.int64)
    chain = np.empty((n_steps, n_total, ndim))
    chain_lnp = np.empty((n_steps, n_total))
    for step in range(n_steps):
        perms = [split(seed, step, e, n) for e in range(n_ensembles)]
        for half in range(2):
            rows = halfstep_rows(pos, lnp, perms, seed, step, half, 0, half_n * n_ensembles, n, a, lnprob_fn=None,
                                 batch_fn=batch_fn)
            apply_rows(pos, lnp, acc, perms, half, rows, n, chain[step], chain_lnp[step])
    return chain, chain_lnp, acc


# ---------------------------------------------------------------- the walker-sharded protocol (include/magprop_amd.h:
# mp_sampler_halfstep_shard / mp_sampler_halfstep_apply), restated on numpy arrays
def halfstep_rows(pos, lnp, perms, seed, step, half, lo, hi, n, a=2.0, lnprob_fn=gaussian_lnprob, batch_fn=None):
    """Outcome rows (proposal, lnprob, accepted, status) of slots [lo, hi) of the active half, all ensembles flattened;
    `perms[e]` = this step's split of ensemble e (n walkers each).  Reads pos / lnp, changes nothing.
    batch_fn(proposals) evaluates all the slots' proposals in one call instead of lnprob_fn one by one."""
    ndim = pos.shape[1]
    half_n = n // 2
    rows = np.zeros((hi - lo, ndim + 3))
    new_all = None
    if batch_fn is not None:
        props = np.empty((hi - lo, ndim))
        for gs in range(lo, hi):
            e, slot = divmod(gs, half_n)
            base, perm = e * n, perms[e]
            k = base + perm[half * half_n + slot]
            r = philox4x32_10(seed & M32, seed >> 32, step, half, k, 0)
            n_comp = n - half_n
            j = base + perm[(1 - half) * half_n + min(int(u01(r[0], r[1]) * n_comp), n_comp - 1)]
            zr = (a - 1.0) * u01(r[2], r[3]) + 1.0
            props[gs - lo] = pos[j] - (pos[j] - pos[k]) * (zr * zr / a)
        new_all = np.asarray(batch_fn(props), dtype=np.float64)
    for gs in range(lo, hi):
        e, slot = divmod(gs, half_n)
        base, perm = e * n, perms[e]
        k = base + perm[half * half_n + slot]
        r = philox4x32_10(seed & M32, seed >> 32, step, half, k, 0)
        r2 = philox4x32_10(seed & M32, seed >> 32, step, half, k, 1)
        n_comp = n - half_n
        jc = int(u01(r[0], r[1]) * n_comp)
        j = base + perm[(1 - half) * half_n + min(jc, n_comp - 1)]
        zr = (a - 1.0) * u01(r[2], r[3]) + 1.0
        zz = zr * zr / a
        prop = pos[j] - (pos[j] - pos[k]) * zz
        new = lnprob_fn(prop) if new_all is None else new_all[gs - lo]
        lnpdiff = (ndim - 1.0) * np.log(zz) + new - lnp[k]
        with np.errstate(divide="ignore"):
            accept = lnpdiff > np.log(u01(r2[0], r2[1]))
        rows[gs - lo, :ndim] = prop
        rows[gs - lo, ndim] = new
        rows[gs - lo, ndim + 1] = 1.0 if accept else 0.0
    return rows


def apply_rows(pos, lnp, acc, perms, half, rows, n, chain_row=None, lnp_row=None):
    """Commit the gathered outcome rows of one half-step (rows[slot]) to the state, in place."""
    ndim = pos.shape[1]
    half_n = n // 2
    for gs in range(half_n * len(perms)):
        e, slot = divmod(gs, half_n)
        k = e * n + perms[e][half * half_n + slot]
        if rows[gs, ndim + 1] != 0.0:
            pos[k] = rows[gs, :ndim]
            lnp[k] = rows[gs, ndim]
            acc[k] += 1
        if chain_row is not None:
            chain_row[k] = pos[k]
            lnp_row[k] = lnp[k]


# ---------------------------------------------------------------- the whole-step protocol (include/magprop_amd.h:
# mp_sampler_step_shard / mp_sampler_step_apply): block b of 3 * slots evaluates, type = b // slots,
#   0: the proposal of slot b % slots of the first half; 1 / 2: the proposal of that slot of the second half with its partner
#   of the first half where it stands / at the partner's own proposal.
# Row = proposal, lnprob, status, (ndim - 1) ln z, ln u, lnprob of the walker before the move, partner's slot.
def _draw(seed, step, half, k, n_comp, a):
    r = philox4x32_10(seed & M32, seed >> 32, step, half, k, 0)
    r2 = philox4x32_10(seed & M32, seed >> 32, step, half, k, 1)
    jc = min(int(u01(r[0], r[1]) * n_comp), n_comp - 1)
    zr = (a - 1.0) * u01(r[2], r[3]) + 1.0
    with np.errstate(divide="ignore"):
        logu = np.log(u01(r2[0], r2[1]))
    return jc, zr * zr / a, logu


def step_rows(pos, lnp, perms, seed, step, lo, hi, n, a=2.0, lnprob_fn=gaussian_lnprob):
    """Outcome rows of blocks [lo, hi) of a whole step.  Reads pos / lnp, changes nothing."""
    ndim = pos.shape[1]
    half_n = n // 2
    n_slots = half_n * len(perms)
    n_comp = n - half_n
    rows = np.zeros((hi - lo, ndim + 6))
    for b in range(lo, hi):
        typ, gs = divmod(b, n_slots)
        half = 0 if typ == 0 else 1
        e, slot = divmod(gs, half_n)
        base, perm = e * n, perms[e]
        k = base + perm[half * half_n + slot]
        jc, zz, logu = _draw(seed, step, half, k, n_comp, a)
        j = base + perm[(1 - half) * half_n + jc]
        xj = pos[j]
        if typ == 2:   # the partner's own proposal of the first half
            jcj, zzj, _ = _draw(seed, step, 0, j, n_comp, a)
            jj = base + perm[half_n + jcj]
            xj = pos[jj] - (pos[jj] - pos[j]) * zzj
        prop = xj - (xj - pos[k]) * zz
        rows[b - lo, :ndim] = prop
        rows[b - lo, ndim] = lnprob_fn(prop)
        rows[b - lo, ndim + 2] = (ndim - 1.0) * np.log(zz)
        rows[b - lo, ndim + 3] = logu
        rows[b - lo, ndim + 4] = lnp[k]
        rows[b - lo, ndim + 5] = jc
    return rows


def apply_step_rows(pos, lnp, acc, perms, rows, n, chain_row=None, lnp_row=None):
    """Commit the gathered rows of a whole step (rows[block]) in emcee's order: the first half decides, every walker of the
    second half takes the candidate that matches its partner's outcome and decides on it."""
    ndim = pos.shape[1]
    half_n = n // 2
    n_slots = half_n * len(perms)

    def accepted(u):
        return (u[ndim + 2] + u[ndim]) - u[ndim + 4] > u[ndim + 3]
    moved0 = [accepted(rows[gs]) for gs in range(n_slots)]
    for half in range(2):
        for gs in range(n_slots):
            e, slot = divmod(gs, half_n)
            k = e * n + perms[e][half * half_n + slot]
            u = rows[gs]
            if half == 1:
                u1 = rows[n_slots + gs]
                u = rows[2 * n_slots + gs] if moved0[e * half_n + int(u1[ndim + 5])] else u1
            if accepted(u):
                pos[k] = u[:ndim]
                lnp[k] = u[ndim]
                acc[k] += 1
            if chain_row is not None:
                chain_row[k] = pos[k]
                lnp_row[k] = lnp[k]


class NumpyShardEngine:
    """The engine protocol of magprop_amd.distributed.DistributedEnsembleSampler on CPU tensors (tests of the sharding,
    the gather layout and the commit order without a GPU)."""

    def __init__(self, n_walkers, ndim, seed, a=2.0, n_ensembles=1, lnprob_fn=gaussian_lnprob):
        import torch
        self.device = torch.device("cpu")
        self.n, self.ndim, self.seed, self.a, self.n_ens, self.fn = n_walkers, ndim, seed, a, n_ensembles, lnprob_fn
        self.ntotal = n_walkers * n_ensembles
        self.n_slots = (n_walkers // 2) * n_ensembles
        self.row_doubles = ndim + 3
        self.step = 0
        self.calls = 0

    def set_positions(self, pos):
        self.pos = np.array(pos, dtype=np.float64)
        self.lnp = np.array([self.fn(p) for p in self.pos])
        self.acc = np.zeros(self.ntotal, dtype=np.int64)

    def _perms(self):
        return [split(self.seed, self.step, e, self.n) for e in range(self.n_ens)]

    def halfstep_shard(self, half, lo, hi, rows):
        self.calls += hi - lo
        rows.numpy()[: hi - lo] = halfstep_rows(self.pos, self.lnp, self._perms(), self.seed, self.step, half, lo, hi,
                                               self.n, self.a, self.fn)

    def halfstep_apply(self, half, rows, chain_row=None, lnp_row=None):
        apply_rows(self.pos, self.lnp, self.acc, self._perms(), half, rows.numpy(), self.n,
                   None if chain_row is None else chain_row.numpy(), None if lnp_row is None else lnp_row.numpy())
        if half == 1:
            self.step += 1

    # whole-step protocol
    @property
    def step_blocks(self):
        return 3 * self.n_slots

    @property
    def step_row_doubles(self):
        return self.ndim + 6

    def whole_step_ok(self, world):
        return True

    def step_shard(self, lo, hi, rows):
        self.calls += hi - lo
        rows.numpy()[: hi - lo] = step_rows(self.pos, self.lnp, self._perms(), self.seed, self.step, lo, hi, self.n, self.a, self.fn)

    def step_apply(self, rows, chain_row=None, lnp_row=None):
        apply_step_rows(self.pos, self.lnp, self.acc, self._perms(), rows.numpy(), self.n,
                        None if chain_row is None else chain_row.numpy(), None if lnp_row is None else lnp_row.numpy())
        self.step += 1

    def state(self):
        return self.pos.copy(), self.lnp.copy(), self.acc.copy()

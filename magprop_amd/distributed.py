"""Walker sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL).

The path shards naturally: each walker's lnprob depends only on its own parameters and on read-only
shared data (grid, datasets, prior) that every rank holds.  The only exchange is an all-gather of the
per-rank lnprob slices so that every rank sees the full ensemble for the stretch-move accept/reject
(the reference's counterpart is emcee's ``pool.map`` over walkers, synth_mcmc.py:178-185).
Payload is n*8 bytes (32 KB at 4096 proposals): latency-bound on xGMI, one collective per call.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous block of walkers for `rank`: equal-size blocks of ceil(n/world), the tail may be short/empty."""
    per = -(-n // world)
    lo = min(rank * per, n)
    return lo, min(lo + per, n), per


class ShardedLnprob:
    """Evaluate lnprob for a full (n, ndim) proposal batch with the work split over the process group.

    eval_local(pars_local) -> lnprob tensor of the local slice, on the same device (with `writes_out=True` it is called
    as eval_local(pars_local, out=slice_of_the_send_buffer) and writes there: no copy).  Every rank passes the same full
    `pars` (replicated proposals: no scatter needed) and gets the full lnprob vector.

    `sharded(pars)` is the blocking form.  `t = sharded.start(pars)` ... `full = sharded.finish(t)` is the pipelined
    form: start() launches this rank's block on the current stream and enqueues the all-gather on RCCL's own stream
    without making the compute stream wait for it, so the next independent batch's kernel overlaps the collective
    (two batches in flight, double-buffered; the tensor returned by finish() stays valid until the second start() after
    it).  Dependent batches (the two half-steps of one stretch move) have nothing to overlap and use the blocking form.
    """

    def __init__(self, eval_local, group=None, via_host=False, writes_out=False, always_gather=False):
        self.eval_local = eval_local
        self.group = group
        self.via_host = via_host      # gather through host memory (gloo rehearsal of the multi-GPU path)
        self.writes_out = writes_out
        self.always_gather = always_gather   # run the collective even in a group of one (exercises RCCL on a 1-GPU box)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._slots = [None, None]    # per slot: (key, send buffer [per], receive buffer [per * world])
        self._k = 0

    def _buffers(self, n, per, device):
        k = self._k
        self._k ^= 1
        key = (n, per, str(device))
        slot = self._slots[k]
        if slot is None or slot[0] != key:
            local = torch.full((per,), float("-inf"), dtype=torch.float64, device=device)   # a short tail block stays -inf
            slot = (key, local, torch.empty(per * self.world, dtype=torch.float64, device=device))
            self._slots[k] = slot
        return slot[1], slot[2]

    def start(self, pars, recv=None):
        """recv: optional caller-owned float64 tensor that receives the full lnprob vector (at least
        ceil(n/world)*world elements; n when the group has one rank) instead of the internal double buffer."""
        n = pars.shape[0]
        if self.world == 1 and not self.always_gather:
            if recv is not None and self.writes_out:
                self.eval_local(pars, out=recv[:n])
                return None, recv, n
            return None, self.eval_local(pars), n
        lo, hi, per = shard_range(n, self.rank, self.world)
        local, buf = self._buffers(n, per, pars.device)
        if recv is not None:
            buf = recv[: per * self.world]
        if hi > lo:
            if self.writes_out:
                self.eval_local(pars[lo:hi], out=local[: hi - lo])
            else:
                local[: hi - lo] = self.eval_local(pars[lo:hi])
        if self.via_host:
            hbuf = torch.empty(per * self.world, dtype=torch.float64)
            dist.all_gather_into_tensor(hbuf, local.cpu(), group=self.group)
            buf.copy_(hbuf)
            return None, buf, n
        work = dist.all_gather_into_tensor(buf, local, group=self.group, async_op=True)
        return work, buf, n

    @staticmethod
    def finish(ticket):
        work, buf, n = ticket
        if work is not None:
            work.wait()               # the current stream waits for the collective; the host does not
        return buf[:n]

    def __call__(self, pars):
        return self.finish(self.start(pars))


class DistributedStretchSampler:
    """emcee's stretch move with the walkers' log-posterior evaluations sharded over the process group.

    Every rank holds the full ensemble state and draws the same random numbers (replicated generator, same seed), so
    the proposals of a half-step are identical everywhere; rank r evaluates its contiguous block of them on its GPU and
    ONE all-gather of the lnprob slices (RCCL over xGMI) lets every rank take the same accept/reject decisions — the
    pattern BASELINE.json's north star describes for 8 GPUs.  The move itself is a handful of small tensor operations on
    the state's device; the single-GPU, fully fused version is `magprop_amd.EnsembleSampler`.

    eval_local(pars_local) -> lnprob of the local block (same device), e.g. `LogProb.lnprob_device`.
    """

    def __init__(self, eval_local, nwalkers, ndim, seed=0, a=2.0, group=None, device="cpu", via_host=False):
        if nwalkers % 2 or nwalkers < 2:
            raise ValueError("nwalkers must be even")
        self.nwalkers, self.ndim, self.a = int(nwalkers), int(ndim), float(a)
        self.device = torch.device(device)
        self.lnprob_fn = ShardedLnprob(eval_local, group=group, via_host=via_host)
        self.gen = torch.Generator(device="cpu").manual_seed(int(seed))     # replicated on every rank
        self.pos = None
        self.lnp = None
        self.naccepted = torch.zeros(self.nwalkers, dtype=torch.int64, device=self.device)
        self.iteration = 0

    def _rand(self, n):
        return torch.rand(n, dtype=torch.float64, generator=self.gen).to(self.device)

    def run_mcmc(self, pos, nsteps, store=True):
        """Returns (chain[nsteps, nwalkers, ndim], lnprob[nsteps, nwalkers]) on the state's device (or None, None)."""
        if pos is not None:
            self.pos = torch.as_tensor(pos, dtype=torch.float64).to(self.device).contiguous().clone()
            if self.pos.shape != (self.nwalkers, self.ndim):
                raise ValueError(f"pos must have shape {(self.nwalkers, self.ndim)}")
            self.lnp = self.lnprob_fn(self.pos).clone()
        if self.pos is None:
            raise RuntimeError("no state: pass the initial positions first")
        n, half = self.nwalkers, self.nwalkers // 2
        chain = torch.empty(nsteps, n, self.ndim, dtype=torch.float64, device=self.device) if store else None
        clnp = torch.empty(nsteps, n, dtype=torch.float64, device=self.device) if store else None
        for step in range(nsteps):
            perm = torch.randperm(n, generator=self.gen).to(self.device)           # random red/blue split
            for h in range(2):
                act = perm[h * half:(h + 1) * half]
                comp = perm[(1 - h) * half:(2 - h) * half]
                zz = ((self.a - 1.0) * self._rand(half) + 1.0) ** 2 / self.a          # g(z) ~ 1/sqrt(z) on [1/a, a]
                partner = comp[torch.randint(half, (half,), generator=self.gen).to(self.device)]
                xk, xj = self.pos[act], self.pos[partner]
                prop = (xj - (xj - xk) * zz[:, None]).contiguous()
                new = self.lnprob_fn(prop)                                         # sharded kernel + all-gather
                lnpdiff = (self.ndim - 1.0) * torch.log(zz) + new - self.lnp[act]
                accept = lnpdiff > torch.log(self._rand(half))                     # False for NaN / -inf proposals
                self.pos[act] = torch.where(accept[:, None], prop, xk)
                self.lnp[act] = torch.where(accept, new, self.lnp[act])
                self.naccepted[act] += accept.to(torch.int64)
            if store:
                chain[step] = self.pos
                clnp[step] = self.lnp
            self.iteration += 1
        return chain, clnp

    @property
    def acceptance_fraction(self):
        return self.naccepted.to(torch.float64) / max(self.iteration, 1)

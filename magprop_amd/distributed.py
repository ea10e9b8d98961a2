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

    eval_local(pars_local) -> lnprob tensor of the local slice, on the same device.  Every rank passes
    the same full `pars` (replicated proposals: no scatter needed) and gets the full lnprob vector.
    """

    def __init__(self, eval_local, group=None, via_host=False):
        self.eval_local = eval_local
        self.group = group
        self.via_host = via_host      # gather through host memory (gloo rehearsal of the multi-GPU path)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._buf = None

    def __call__(self, pars):
        n = pars.shape[0]
        lo, hi, per = shard_range(n, self.rank, self.world)
        if self.world == 1:
            return self.eval_local(pars)
        local = torch.full((per,), float("-inf"), dtype=torch.float64, device=pars.device)
        if hi > lo:
            local[: hi - lo] = self.eval_local(pars[lo:hi])
        if self._buf is None or self._buf.numel() != per * self.world or self._buf.device != pars.device:
            self._buf = torch.empty(per * self.world, dtype=torch.float64, device=pars.device)
        if self.via_host:
            hbuf = torch.empty(per * self.world, dtype=torch.float64)
            dist.all_gather_into_tensor(hbuf, local.cpu(), group=self.group)
            self._buf.copy_(hbuf)
        else:
            dist.all_gather_into_tensor(self._buf, local, group=self.group)
        return self._buf[:n]

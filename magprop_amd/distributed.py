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

    Slot reuse is ordered explicitly: a slot remembers the collective that last read its send buffer (and wrote its receive
    buffer), and the start() that takes the slot again -- two passes later -- lets its stream wait for that collective
    before the kernel rewrites the buffer, UNLESS it has completed already (the normal case: one host-side poll, no
    cross-queue wait).  Without this nothing would order kernel i+2 behind all-gather i once finish() waits on a consumer
    stream instead of the compute stream: a rank running ahead of a slow peer would send pass i+2 values labelled pass i.
    """

    def __init__(self, eval_local, group=None, via_host=False, writes_out=False, always_gather=False):
        self.eval_local = eval_local
        self.group = group
        self.via_host = via_host      # gather through host memory (gloo rehearsal of the multi-GPU path)
        self.writes_out = writes_out
        self.always_gather = always_gather   # run the collective even in a group of one (exercises RCCL on a 1-GPU box)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._slots = [None, None]    # per slot: [key, send buffer [per], receive buffer [per * world], last collective on it]
        self._k = 0

    def _buffers(self, n, per, device):
        k = self._k
        self._k ^= 1
        key = (n, per, str(device))
        slot = self._slots[k]
        if slot is not None and slot[3] is not None:
            # the collective that used this slot two passes ago may still be reading the send buffer (a slow peer): the
            # stream about to rewrite it waits for that collective -- only if it is really still in flight
            if not slot[3].is_completed():
                slot[3].wait()
            slot[3] = None
        if slot is None or slot[0] != key:
            local = torch.full((per,), float("-inf"), dtype=torch.float64, device=device)   # a short tail block stays -inf
            slot = [key, local, torch.empty(per * self.world, dtype=torch.float64, device=device), None]
            self._slots[k] = slot
        return slot[1], slot[2], slot

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
        local, buf, slot = self._buffers(n, per, pars.device)
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
        slot[3] = work
        return work, buf, n

    @staticmethod
    def finish(ticket, stream=None):
        """The full lnprob vector of a started batch.  The stream that will consume it (`stream`, default: the current
        one) waits for the collective; the host does not.  Independent batches should name a consumer stream other than
        the one their kernels run on: a cross-queue wait costs the waiting queue ~10 us on MI355X whether or not the
        collective has already finished (measured: profiles/r04_rccl_overlap.md), which would delay the next batch's
        kernel for nothing."""
        work, buf, n = ticket
        if work is not None:
            if stream is None:
                work.wait()
            else:
                with torch.cuda.stream(stream):
                    work.wait()
        return buf[:n]

    def __call__(self, pars):
        return self.finish(self.start(pars))


class HipShardEngine:
    """Adapter: a `magprop_amd.EnsembleSampler` (the C ABI's mp_sampler_halfstep_* entry points) behind the tensor-level
    engine protocol `DistributedEnsembleSampler` drives.  Everything is enqueued on torch's current stream of the
    sampler's device, so torch.distributed orders its collectives with the kernels."""

    def __init__(self, sampler, device):
        self.s = sampler
        self.device = torch.device(device)
        self.ntotal, self.ndim = sampler.ntotal, sampler.ndim
        self.n_slots, self.row_doubles = sampler.n_slots, sampler.row_doubles

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def set_positions(self, pos):
        self.s.set_positions(pos)          # every rank evaluates the whole initial ensemble (once per run)

    def halfstep_shard(self, half, lo, hi, rows):
        assert rows.is_contiguous() and rows.dtype == torch.float64 and rows.shape[0] >= hi - lo
        self.s.halfstep_shard(half, lo, hi, rows.data_ptr() if hi > lo else 0, stream=self._stream())

    def halfstep_apply(self, half, rows, chain_row=None, lnp_row=None):
        assert rows.is_contiguous() and rows.shape[0] >= self.n_slots
        self.s.halfstep_apply(half, rows.data_ptr(), chain_row.data_ptr() if chain_row is not None else 0,
                              lnp_row.data_ptr() if lnp_row is not None else 0, stream=self._stream())

    # whole-step protocol
    @property
    def step_blocks(self):
        return self.s.step_blocks

    @property
    def step_row_doubles(self):
        return self.s.step_row_doubles

    def whole_step_ok(self, world):
        """A rank's share of the 3/2 x walkers evaluations of a whole step fits its device two wavefronts per SIMD."""
        return -(-self.s.step_blocks // world) <= 2 * self.s.handle.n_simd

    def step_shard(self, lo, hi, rows):
        assert rows.is_contiguous() and rows.dtype == torch.float64 and rows.shape[0] >= hi - lo
        self.s.step_shard(lo, hi, rows.data_ptr() if hi > lo else 0, stream=self._stream())

    def step_apply(self, rows, chain_row=None, lnp_row=None):
        assert rows.is_contiguous() and rows.shape[0] >= self.step_blocks
        self.s.step_apply(rows.data_ptr(), chain_row.data_ptr() if chain_row is not None else 0,
                          lnp_row.data_ptr() if lnp_row is not None else 0, stream=self._stream())

    def state(self):
        """(pos, lnprob, n_accepted) as numpy arrays (synchronises)."""
        return self.s.get_last_sample()

    def drain_bad(self):
        """Move the device window of failed proposals into the library's host log (synchronises the device) and append
        the new rows to the sampler's fbad file, if it has one."""
        self.s.get_bad(first_row=0, max_rows=0)
        self.s._flush_fbad()


class DistributedEnsembleSampler:
    """emcee's stretch move on an ensemble whose walkers are sharded over the process group, one fused kernel and ONE
    all-gather per half-step (the pattern of code/synthetic_datasets/synth_mcmc.py:178-185 with the pool replaced by GPUs).

    Every rank holds the full ensemble state in HBM and the same seed.  Per half-step, rank r runs the fused stretch
    kernel over ITS contiguous block of the active half's slots — the random numbers are counter-based and keyed by the
    global walker index, so each rank draws exactly what a single-GPU launch would have drawn for those walkers, with no
    communication — which leaves one outcome row per slot (proposal, its lnprob, accepted, status).  The rows of all
    ranks are all-gathered (RCCL over xGMI; 72 B per proposal) and every rank commits all of them with a small kernel.
    Nothing synchronises with the host inside the loop and no random number crosses a wire.  The chain equals the
    single-GPU `magprop_amd.EnsembleSampler` chain (bit for bit while the per-rank block selects the same kernel variant).

    engine: HipShardEngine (product) or any object with the same methods (tests drive the protocol on CPU with the numpy
    restatement of the move).
    """

    def __init__(self, engine, group=None, via_host=False, always_gather=False, whole_step=None):
        """whole_step: None = a whole step per launch and ONE all-gather per step when the engine offers it and a rank's
        share of the 3/2 x walkers evaluations fits its device (engine.whole_step_ok); False = one launch and one gather
        per half-step (larger ensembles get that anyway).  Same chain either way."""
        self.engine = engine
        self.group = group
        self.via_host = via_host       # gather through host memory: gloo rehearsal with GPU engines
        self.always_gather = always_gather   # run the collective even in a group of one (exercises RCCL on a 1-GPU box)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if whole_step is None:
            ok = bool(hasattr(engine, "step_shard") and engine.whole_step_ok(self.world))
            # The protocol fixes the row width and the number of rows of the all-gather, so every rank must choose the same
            # one: the local verdict depends on the local device (its SIMD count), and ranks on different devices or
            # partitions could disagree.  ONE all-reduce(MIN) at construction settles it.
            if self.world > 1:
                v = torch.tensor([1 if ok else 0], dtype=torch.int32,
                                 device=engine.device if dist.get_backend(group) == "nccl" else "cpu")
                dist.all_reduce(v, op=dist.ReduceOp.MIN, group=group)
                ok = bool(int(v.item()))
            self.whole_step = ok
        else:
            self.whole_step = bool(whole_step)
        n_units = engine.step_blocks if self.whole_step else engine.n_slots        # what is sharded: blocks of a step / slots of a half
        self.lo, self.hi, self.per = shard_range(n_units, self.rank, self.world)
        R = engine.step_row_doubles if self.whole_step else engine.row_doubles
        dev = engine.device
        self.send = torch.zeros(self.per, R, dtype=torch.float64, device=dev)                 # this rank's outcome rows
        alone = self.world == 1 and not always_gather
        self.rows = self.send if alone else torch.zeros(self.per * self.world, R, dtype=torch.float64, device=dev)
        self.iteration = 0
        self.drain_every = 256         # steps between drains of the device window of failed proposals (fbad)

    def _drain_bad(self):
        """The sharded entry points never drain the 65 536-row device window of failed proposals themselves (they only
        enqueue work); a long run started prior-wide would overflow it and lose fbad rows.  Draining synchronises the
        device, so it happens every `drain_every` steps and at the end of run_mcmc, not per step."""
        drain = getattr(self.engine, "drain_bad", None)
        if drain is not None:
            drain()

    def _gather(self):
        if self.rows is self.send:
            return
        if self.via_host:
            host = torch.empty(self.rows.shape, dtype=torch.float64)
            dist.all_gather_into_tensor(host.view(-1), self.send.cpu().view(-1), group=self.group)
            self.rows.copy_(host)
        else:
            dist.all_gather_into_tensor(self.rows.view(-1), self.send.view(-1), group=self.group)

    def run_mcmc(self, pos, nsteps, store=True):
        """pos: (ntotal, ndim) array or None to continue.  Returns (chain[nsteps, ntotal, ndim], lnprob[nsteps, ntotal])
        tensors on the engine's device (None, None with store=False)."""
        e = self.engine
        if pos is not None:
            e.set_positions(pos)
        chain = lnp = None
        if store and nsteps > 0:
            chain = torch.empty(nsteps, e.ntotal, e.ndim, dtype=torch.float64, device=e.device)
            lnp = torch.empty(nsteps, e.ntotal, dtype=torch.float64, device=e.device)
        for step in range(nsteps):
            if self.whole_step:
                e.step_shard(self.lo, self.hi, self.send)
                self._gather()     # row index == block index
                e.step_apply(self.rows, chain[step] if store else None, lnp[step] if store else None)
                self.iteration += 1
                if self.iteration % self.drain_every == 0:
                    self._drain_bad()
                continue
            for half in (0, 1):
                e.halfstep_shard(half, self.lo, self.hi, self.send)
                self._gather()     # rank r's block lands at rows [r*per, (r+1)*per): row index == slot index (shard_range)
                e.halfstep_apply(half, self.rows, chain[step] if store else None, lnp[step] if store else None)
            self.iteration += 1
            if self.iteration % self.drain_every == 0:
                self._drain_bad()
        if nsteps > 0:
            self._drain_bad()
        return chain, lnp

    @property
    def acceptance_fraction(self):
        return self.engine.state()[2] / max(self.iteration, 1)

"""Data parallelism for the trainer: one process per GPU, gradients averaged with ONE flat-bucket all-reduce
per step over RCCL/xGMI (``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" on CPU for tests).

The reference has no distributed code.  Meshes of a batch are independent through the
forward pass and the Chamfer loss, so the batch is sharded across ranks and the only exchange is the
gradient sum: 3.29 M fp32 = 13.1 MB for the image-free model.  A single flat bucket (instead of ~130
per-tensor collectives) keeps the exchange latency-bound work to one launch; ring all-reduce over 7 xGMI
links moves 2*(N-1)/N * 13.1 MB per GPU ~ 0.15 ms against a >= 40 ms compute step.
"""
import os

import torch
import torch.distributed as dist
import torch.utils.data


def init_from_env(backend=None):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class FlatGradBucket:
    """One contiguous buffer for the whole gradient: collectives reduce it in place and the optimiser reads the averaged
    values there (every ``.grad`` is a view into the buffer when ``optimizer.step()`` runs).

    Step protocol: ``zero()`` -> ``loss.backward()`` -> ``all_reduce_mean()`` -> ``optimizer.step()``.
    ``zero()`` clears the ``.grad`` fields rather than memsetting the buffer, so autograd *assigns* each gradient
    instead of launching one ``grad += g`` kernel per parameter (131 launches per step on the 20 x 300 model);
    ``all_reduce_mean()`` then brings them into the buffer with one multi-tensor copy and re-homes ``.grad``.  On the GPU the
    GCN stacks' gradients — 99.9 % of the bytes — never take that detour: the library writes them into these views itself
    (``ops.GradSink``: first use of a step overwrites, the second use of the shared ``mesh_deform_2`` accumulates).

    Two chunks: ``early`` names parameters whose gradients are final before the backward pass ends — in
    ``Deformation`` the modules only stages 2 and 3 use (``mesh_deform_2``; ``img_encoder_local``).  They sit at the front
    of the buffer and carry a post-accumulate hook each: when the last of them has received its gradient, ``reduce_early()``
    starts their all-reduce asynchronously, so it overlaps with the rest of the backward pass (stage 1 and the global image
    encoder); ``all_reduce_mean()`` reduces the rest and waits for both.

    Early parameters that never receive a gradient (``Image_Encoder.forward`` leaves its loop before the last layers of the
    default 6 x 3 pyramid: 20 tensors of ``img_encoder_local``) would keep the countdown from ever reaching zero, so the
    countdown only counts the early parameters that HAD a gradient in the previous step (at first all are assumed to; a
    model with unused early parameters starts nothing early in its first step, learns the count, and overlaps from then on).  Whether the early chunk started during the backward pass or not,
    the collective SEQUENCE is always [early chunk][rest]: a rank whose countdown did not fire issues the same two calls at
    the end, so ranks cannot fall out of step over it.

    ``force_collectives=True`` issues the collectives on an initialised process group of ONE rank too (they are no-ops
    numerically): the RCCL path — ``div_`` → async ``all_reduce`` on the communicator's stream → ``wait`` → optimiser — then
    runs on a single GPU exactly as it does on eight (tests/test_gpu_trainer.py)."""

    def __init__(self, params, early=(), sinks=True, force_collectives=False):
        self.force_collectives = bool(force_collectives)
        params = [p for p in params if p.requires_grad]
        early_ids = {id(p) for p in early}
        self.params = [p for p in params if id(p) in early_ids] + [p for p in params if id(p) not in early_ids]
        self.n_early = sum(1 for p in params if id(p) in early_ids)
        n = sum(p.numel() for p in self.params)
        # one element behind the gradients travels with the [rest] reduce: the "an early gradient arrived late" flag of
        # all_reduce_mean(), so that every rank learns of a rank's late gradient without a collective of its own
        self._store = torch.zeros(n + 1, dtype=self.params[0].dtype, device=self.params[0].device)
        self.flat = self._store[:n]
        self._flag = self._store[n:]
        self._flag_host = torch.zeros(1, dtype=self.flat.dtype).pin_memory() if self.flat.is_cuda else None
        self._flag_event = None
        self.views = []
        off = 0
        for k, p in enumerate(self.params):
            if k == self.n_early:
                self.early_numel = off
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        if self.n_early == len(self.params):
            self.early_numel = off
        self._early_done = False
        self._early_work = None
        self._early_live = self.n_early   # early parameters that received a gradient in the previous step (at first: all assumed)
        self._fired = 0
        self._late = False
        self._pending = self.n_early if self.n_early else -1
        self.early_started_in_backward = 0   # steps whose early all-reduce was launched from inside the backward pass
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_early_grad) for p in self.params[:self.n_early]]
        # Gradients written where they live: the library's GCN backward takes these views as its output pointers (first use
        # of a step overwrites, the second use of the shared mesh_deform_2 accumulates in the kernel), so for those
        # parameters there is no fresh tensor for autograd to assign, no add kernel for the second use and nothing to copy
        # in gather().  Only on the GPU (the sinks are consulted by ops.GCNStackFn).
        self._sinks = []
        self._sink_by_id = {}
        self._open = False
        if sinks and self.flat.is_cuda:
            from . import ops as _ops
            for k, (p, v) in enumerate(zip(self.params, self.views)):
                self._sinks.append(_ops.register_grad_sink(p, v, self._on_early_grad_sink if k < self.n_early else None))
                self._sink_by_id[id(p)] = self._sinks[-1]
        self._rehome()

    def _on_early_grad(self, _param):
        if _param is not None and self._sink_by_id:
            # torch fires the post-accumulate hook even when the incoming gradient is undefined — which is what the library
            # returns for a parameter whose gradient it wrote through the sink (counted once, by the sink's on_final)
            k = self._sink_by_id.get(id(_param))
            if k is not None and k.written:
                return
        self._fired += 1
        if self._early_done:
            self._late = True             # arrived after its chunk was gathered (and possibly reduced): see all_reduce_mean()
        self._pending -= 1
        if self._pending == 0:
            self.early_started_in_backward += 1
            self.reduce_early()

    def close(self):
        """Detach from the parameters: remove the gradient hooks and the library's gradient sinks (a bucket that is replaced
        must not keep writing into — or reducing — its old buffer)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._sinks:
            from . import ops as _ops
            for p, k in zip(self.params, self._sinks):
                _ops.unregister_grad_sink(p, k)
        self._sinks = []

    def _on_early_grad_sink(self):
        self._on_early_grad(None)

    def _rehome(self, lo=0, hi=None):
        for p, v in zip(self.params[lo:hi], self.views[lo:hi]):
            p.grad = v

    def zero(self):
        if self._open and any(k.written for k in self._sinks):
            # the library wrote gradients into the buffer but nobody re-homed the .grad fields: the optimiser would have
            # skipped those parameters
            raise RuntimeError("a3vt: FlatGradBucket.all_reduce_mean() (or gather()) must run between backward() and "
                               "optimizer.step(); the previous step skipped it")
        self._check_peer_flag()
        self._open = True
        for p in self.params:
            p.grad = None
        self._early_done = False
        self.wait_early()                       # a reduce nobody waited for must not still be writing `flat`
        self._fired = 0
        self._late = False
        self._pending = self._early_live if self._early_live else -1   # unknown / none live: the countdown cannot fire
        for k in self._sinks:
            k.reset()

    def _gather(self, lo, hi):
        """Bring the gradients of params[lo:hi] into their slice of the flat buffer: nothing to do for those the library wrote
        in place (ops.GradSink) or that already live there, one multi-tensor copy for the ones autograd assigned as fresh
        tensors, zeros for parameters the graph did not use."""
        fresh_v, fresh_g, unused = [], [], []
        for k, (p, v) in enumerate(zip(self.params[lo:hi], self.views[lo:hi])):
            g = p.grad
            if g is None:
                if not (self._sinks and self._sinks[lo + k].written):
                    unused.append(v)                    # parameter unused in this graph
            elif g.data_ptr() != v.data_ptr():
                fresh_v.append(v)
                fresh_g.append(g.view_as(v) if g.shape != v.shape else g)
        if fresh_v:
            torch._foreach_copy_(fresh_v, fresh_g)
        if unused:
            torch._foreach_zero_(unused)
        self._rehome(lo, hi)

    def gather(self, wait=True):
        """Collect every gradient into the flat buffer (the early chunk only if ``reduce_early`` has not already).
        ``wait=False`` (all_reduce_mean): leave an early-chunk reduce in flight — the caller waits for it after it has issued
        the [rest] reduce, so the two overlap."""
        if not self._early_done:
            self._gather(0, self.n_early)
        self._gather(self.n_early, len(self.params))
        if wait:
            self.wait_early()                   # callers that read `flat` without all_reduce_mean()
        self._open = False
        self._early_live = self._fired     # what the next step's countdown waits for

    def _active(self):
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force_collectives)

    def reduce_early(self):
        """Gather the early chunk and start its all-reduce (asynchronously).  Safe to call when there is no early chunk or no
        process group (then it only gathers).  Call once per step, when those gradients are final."""
        if self._early_done or self.n_early == 0:
            return
        chunk = self.flat[:self.early_numel]
        self._gather(0, self.n_early)
        self._early_done = True
        if self._active():
            chunk.div_(dist.get_world_size())
            self._early_work = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True)

    def all_reduce_mean(self):
        """Average the whole gradient over the ranks: the early chunk's reduce may already be in flight; the rest is reduced
        here; both are complete on return (on the current stream for RCCL)."""
        started = self._early_done
        # `_late`: more early parameters received a gradient than in the previous step, so the countdown fired before the last
        # of them: that gradient is not in the chunk whose reduce is in flight.  Never silently, and never on one rank alone:
        # the collective sequence [early][rest] is completed first (a rank where this happens — a per-rank, data-dependent graph
        # change — must not leave its peers blocked in [rest]), and a flag element behind the gradients travels with [rest], so
        # EVERY rank learns that some rank's early chunk was averaged without a gradient and raises: at once where the flag
        # can be read without stalling the device (CPU tensors / gloo; the late rank itself), at the start of its next step
        # otherwise (the flag is copied to pinned memory behind the reduce; the check waits for that copy alone).
        late = self._late
        self._check_peer_flag()
        self.gather(wait=False)
        if late:
            self._early_live = self.n_early     # (gather() set it to what fired this step)
        peer_late = False
        if self._active():
            world = dist.get_world_size()
            if self.n_early and not started:
                # the countdown did not fire during the backward pass (first step, or an early parameter without a gradient):
                # same two collectives, in the same order, as on a rank where it did
                chunk = self.flat[:self.early_numel]
                chunk.div_(world)
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM)
            self._flag.fill_(float(world) if late else 0.0)
            rest = self._store[self.early_numel:]           # the other gradients + the flag (>= 1 after the mean: some rank was late)
            rest.div_(world)
            dist.all_reduce(rest, op=dist.ReduceOp.SUM)
            if self._early_work is not None:
                self._early_work.wait()
                self._early_work = None
            if self._flag_host is None:
                peer_late = bool(self._flag.item() >= 0.5)
            else:
                self._flag_host.copy_(self._flag, non_blocking=True)
                self._flag_event = torch.cuda.Event()
                self._flag_event.record()
        else:
            self.wait_early()
        if late or peer_late:
            raise RuntimeError(self._late_message(late))
        return None

    def _late_message(self, here):
        return ("a3vt: FlatGradBucket: an early parameter received its gradient after the early chunk had been gathered"
                + (" on this rank" if here else " on another rank") + " — the set of parameters that receive gradients "
                "changed since the previous step, so the early chunk was averaged WITHOUT that gradient.  Both collectives of "
                "the step have completed on every rank and the flag travelled with them: every rank raises this error (the "
                "late rank and CPU groups in the same step, GPU peers at the start of their next one).  In a multi-process job "
                "do NOT repeat the step on one rank — that would pair its collectives with the peers' next step: stop the job "
                "on all ranks and resume from the last checkpoint.  Single process: discard this step's gradients and repeat "
                "it (the countdown now expects every early parameter).")

    def _check_peer_flag(self):
        """The late flag of the previous step's [rest] reduce (GPU groups: copied to pinned memory behind the reduce)."""
        if self._flag_event is not None:
            self._flag_event.synchronize()      # that copy alone: the stream is long past it
            self._flag_event = None
            if float(self._flag_host[0]) >= 0.5:
                raise RuntimeError(self._late_message(False))

    def wait_early(self):
        """Wait for an early-chunk reduce that is still in flight (callers that read ``flat`` without ``all_reduce_mean``)."""
        if self._early_work is not None:
            self._early_work.wait()
            self._early_work = None


def broadcast_parameters(module, src=0):
    """Make every rank start from rank `src`'s weights (one flat broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    ps = [p.data for p in module.parameters()]
    flat = torch.cat([p.reshape(-1) for p in ps])
    dist.broadcast(flat, src)
    off = 0
    for p in ps:
        p.copy_(flat[off:off + p.numel()].view_as(p))
        off += p.numel()
    from . import ops
    ops.invalidate_bf16_copies()                # writes through .data move no version counter


def broadcast_buffers(module, src=0):
    """Make every rank hold rank `src`'s module buffers (the running statistics of the image encoders' BatchNorm layers,
    reference vision/model.py:15-23).  The reference has no SyncBN and neither do we: in training every rank normalises with
    its own batch statistics and updates its own running averages; before a sharded validation or a checkpoint the ranks
    adopt rank 0's, so that every shard is scored with the statistics that are saved."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    bufs = [b for b in module.buffers() if b.numel() > 0]
    floats = [b for b in bufs if b.is_floating_point()]
    if floats:
        flat = torch.cat([b.reshape(-1).float() for b in floats])
        dist.broadcast(flat, src)
        off = 0
        for b in floats:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
    for b in bufs:
        if not b.is_floating_point():                   # num_batches_tracked (int64)
            dist.broadcast(b, src)


class RankPlan:
    """What one rank does in one epoch: ``indices`` (its shard of the dataset, int64) and ``seed`` (for the python /
    numpy / torch generators of this rank: grasp choices, point shuffles, the Philox surface-sample stream)."""

    def __init__(self, indices, seed):
        self.indices, self.seed = indices, seed

    def __len__(self):
        return len(self.indices)


def rank_plan(rank, world, seed, n_items, epoch=0, shuffle=True, pad=True):
    """Per-rank data plan of the data-parallel trainer (SURVEY §8d "per-rank seed = seed + rank", §8e).

    All ranks derive the SAME permutation of ``range(n_items)`` from ``(seed, epoch)`` (no communication) and take every
    ``world``-th entry starting at ``rank`` — shards are disjoint and together cover the dataset.  ``pad=True`` (training)
    first extends the permutation by wrapping around to a multiple of ``world`` so that every rank runs the same number of
    steps (a rank with one batch fewer would leave the others waiting in the gradient all-reduce); ``pad=False``
    (validation) leaves the shards uneven and exactly disjoint — the scores are summed across ranks afterwards.
    The rank's RNG seed is ``seed + rank`` in both cases (same for every epoch: the generators just keep running)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    if shuffle:
        import numpy as np
        order = np.random.default_rng([int(seed), int(epoch)]).permutation(n_items).astype("int64")
    else:
        order = torch.arange(n_items, dtype=torch.int64).numpy()
    if pad and n_items % world and n_items > 0:
        import numpy as np
        extra = world - n_items % world
        order = np.concatenate([order, order[:extra] if extra <= n_items else np.resize(order, extra)])
    return RankPlan(order[rank::world].copy(), int(seed) + int(rank))


class ShardSampler(torch.utils.data.Sampler):
    """``DataLoader`` sampler over this rank's :func:`rank_plan` shard; call ``set_epoch`` before each epoch."""

    def __init__(self, n_items, rank, world, seed, shuffle=True, pad=True):
        self.n_items, self.rank, self.world, self.seed, self.shuffle, self.pad = n_items, rank, world, seed, shuffle, pad
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def plan(self):
        return rank_plan(self.rank, self.world, self.seed, self.n_items, self.epoch, self.shuffle, self.pad)

    def __iter__(self):
        return iter(self.plan().indices.tolist())

    def __len__(self):
        return len(self.plan())


def seed_rank(seed, rank):
    """Seed python / numpy / torch of this process with ``seed + rank`` (after the weights were broadcast): ranks then
    draw different grasp subsets, point shuffles and surface samples instead of N copies of the same batch."""
    import random

    import numpy as np
    s = int(seed) + int(rank)
    random.seed(s)
    np.random.seed(s % (2 ** 32))
    torch.manual_seed(s)
    return s


def all_reduce_sum_(*tensors):
    """In-place sum over ranks of a few small tensors (validation loss / example counts); no-op for one process."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensors
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tensors


def shard_range(global_batch, rank, world):
    """Contiguous [lo, hi) slice of a global batch for this rank (equal shards; remainder to the first ranks)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)

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
    """One contiguous buffer for the whole gradient: a single collective reduces it and the optimiser reads the
    averaged values in place (every ``.grad`` is a view into the buffer when ``optimizer.step()`` runs).

    Step protocol: ``zero()`` -> ``loss.backward()`` -> ``all_reduce_mean()`` -> ``optimizer.step()``.
    ``zero()`` clears the ``.grad`` fields rather than memsetting the buffer, so autograd *assigns* each gradient
    instead of launching one ``grad += g`` kernel per parameter (131 launches per step on the 20 x 300 model);
    ``all_reduce_mean()`` then gathers them into the buffer with one multi-tensor copy and re-homes ``.grad``."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=self.params[0].dtype, device=self.params[0].device)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self._rehome()

    def _rehome(self):
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero(self):
        for p in self.params:
            p.grad = None

    def gather(self):
        """Collect the freshly assigned gradients into the flat buffer (no-op for gradients already living there)."""
        pieces, fresh = [], False
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:                               # parameter unused in this graph
                g = torch.zeros_like(v)
            if g.data_ptr() != v.data_ptr():
                fresh = True
            pieces.append(g)
        if fresh:                                       # pieces that already live in the buffer must not alias the output
            flat_pieces = [g.reshape(-1).clone() if g.data_ptr() == v.data_ptr() else g.reshape(-1)
                           for g, v in zip(pieces, self.views)]
            torch.cat(flat_pieces, out=self.flat)
        self._rehome()

    def all_reduce_mean(self, async_op=False):
        self.gather()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return None
        self.flat.div_(dist.get_world_size())
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)


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


def shard_range(global_batch, rank, world):
    """Contiguous [lo, hi) slice of a global batch for this rank (equal shards; remainder to the first ranks)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
